from gat_amd.data import convert_to_graph, create_knn_graph  # noqa: F401
