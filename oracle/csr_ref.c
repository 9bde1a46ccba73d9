/* Plain-C restatement of the graph-layout step of the GAT layer.  TEST INFRASTRUCTURE
 * ONLY: loaded by tests/, __graft_entry__.smoke() and bench.py's checker, never by the
 * product library.
 *
 * PARITY UNPINNED: the step restated here happens inside torch_geometric 2.5.3's
 * GATConv.forward (remove_self_loops + add_self_loops), which is not under
 * /root/reference; the anchor is the call site /root/reference/gat/model.py:40,43,46
 * and the edge_index orientation produced at /root/reference/gat/converter.py:12-15
 * (row 0 = source, row 1 = destination, int64).
 *
 *   normalised list  = [ edges with src != dst, original order ] ++ [ (n,n) : n = 0..N-1 ]
 *   dst-CSR          = stable counting sort of that list by destination
 *   transposed CSR   = stable counting sort of the dst-CSR edges by source
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* number of edges after normalisation; -1 on an out-of-range node id */
int64_t csr_ref_count(const int64_t *edge_index, int64_t num_edges, int64_t num_nodes,
                      int add_self_loops)
{
    const int64_t *src = edge_index, *dst = edge_index + num_edges;
    int64_t kept = 0;
    for (int64_t e = 0; e < num_edges; ++e) {
        if (src[e] < 0 || src[e] >= num_nodes || dst[e] < 0 || dst[e] >= num_nodes)
            return -1;
        if (!add_self_loops || src[e] != dst[e])
            ++kept;
    }
    return add_self_loops ? kept + num_nodes : kept;
}

/* norm_src / norm_dst: the normalised edge list (E' entries each), may be NULL.
 * rowptr, rowptr_t: N+1 entries; col, perm, col_t, pos_t: E' entries.  Returns E'. */
int64_t csr_ref_build(const int64_t *edge_index, int64_t num_edges, int64_t num_nodes,
                      int add_self_loops, int64_t *norm_src, int64_t *norm_dst,
                      int32_t *rowptr, int32_t *col, int32_t *perm,
                      int32_t *rowptr_t, int32_t *col_t, int32_t *pos_t)
{
    const int64_t ep = csr_ref_count(edge_index, num_edges, num_nodes, add_self_loops);
    if (ep < 0)
        return -1;
    const int64_t *src = edge_index, *dst = edge_index + num_edges;
    int64_t *ns = norm_src ? norm_src : (int64_t *)malloc(sizeof(int64_t) * (size_t)(ep ? ep : 1));
    int64_t *nd = norm_dst ? norm_dst : (int64_t *)malloc(sizeof(int64_t) * (size_t)(ep ? ep : 1));
    int64_t k = 0;
    for (int64_t e = 0; e < num_edges; ++e)
        if (!add_self_loops || src[e] != dst[e]) {
            ns[k] = src[e];
            nd[k] = dst[e];
            ++k;
        }
    if (add_self_loops)
        for (int64_t n = 0; n < num_nodes; ++n, ++k)
            ns[k] = nd[k] = n;

    /* destination CSR */
    int64_t *cursor = (int64_t *)calloc((size_t)num_nodes + 1, sizeof(int64_t));
    for (int64_t e = 0; e < ep; ++e)
        cursor[nd[e] + 1]++;
    for (int64_t n = 0; n < num_nodes; ++n)
        cursor[n + 1] += cursor[n];
    for (int64_t n = 0; n <= num_nodes; ++n)
        rowptr[n] = (int32_t)cursor[n];
    for (int64_t e = 0; e < ep; ++e) {
        const int64_t p = cursor[nd[e]]++;
        col[p] = (int32_t)ns[e];
        perm[p] = (int32_t)e;
    }

    /* transposed CSR over the dst-CSR edge order */
    memset(cursor, 0, sizeof(int64_t) * ((size_t)num_nodes + 1));
    for (int64_t p = 0; p < ep; ++p)
        cursor[col[p] + 1]++;
    for (int64_t n = 0; n < num_nodes; ++n)
        cursor[n + 1] += cursor[n];
    for (int64_t n = 0; n <= num_nodes; ++n)
        rowptr_t[n] = (int32_t)cursor[n];
    for (int64_t i = 0; i < num_nodes; ++i)
        for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
            const int64_t t = cursor[col[p]]++;
            col_t[t] = (int32_t)i;
            pos_t[t] = p;
        }
    free(cursor);
    if (!norm_src)
        free(ns);
    if (!norm_dst)
        free(nd);
    return ep;
}
