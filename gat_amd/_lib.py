"""ctypes binding of libgat_mi355.so (declarations: include/gat_mi355.h).

The library is the only compute path of this package: if it is missing or a call
fails, a RuntimeError is raised -- there is no CPU or eager fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GATMI_LIB") or os.path.join(_PKG, "libgat_mi355.so")  # override: kernel A/B builds

MASK_NONE, MASK_BYTES, MASK_HASH, MASK_HASH_DEV = 0, 1, 2, 3

_i32p = C.POINTER(C.c_int32)


class Plan(C.Structure):
    _fields_ = [("n_items", C.c_int32), ("n_split_rows", C.c_int32), ("n_slots", C.c_int32),
                ("_pad", C.c_int32), ("item_row", C.c_void_p), ("item_beg", C.c_void_p),
                ("item_slot", C.c_void_p), ("split_row", C.c_void_p), ("split_slot", C.c_void_p)]


class Graph(C.Structure):
    _fields_ = [("num_src", C.c_int32), ("num_dst", C.c_int32), ("num_edges", C.c_int32),
                ("seg_len", C.c_int32), ("dst_offset", C.c_int32), ("_pad", C.c_int32),
                ("rowptr", C.c_void_p), ("col", C.c_void_p),
                ("perm", C.c_void_p), ("rowptr_t", C.c_void_p), ("col_t", C.c_void_p),
                ("pos_t", C.c_void_p), ("d2s", C.c_void_p), ("plan", Plan), ("plan_t", Plan)]


class GraphSizes(C.Structure):
    _fields_ = [("max_edges", C.c_int64), ("max_items", C.c_int64),
                ("max_split_rows", C.c_int64), ("workspace_bytes", C.c_int64)]


class Shape(C.Structure):
    _fields_ = [("num_nodes", C.c_int32), ("in_features", C.c_int32), ("heads", C.c_int32),
                ("channels", C.c_int32)]


class Attn(C.Structure):
    _fields_ = [("shape", Shape), ("negative_slope", C.c_float), ("dropout_p", C.c_float),
                ("mask_mode", C.c_int32), ("feat_bf16", C.c_int32), ("seed", C.c_uint64),
                ("mask", C.c_void_p), ("hp", C.c_void_p), ("a_dst", C.c_void_p),
                ("att_src", C.c_void_p), ("att_dst", C.c_void_p), ("bias", C.c_void_p)]


class AttnPartial(C.Structure):
    _fields_ = [("m", C.c_void_p), ("s", C.c_void_p), ("acc", C.c_void_p)]


MAX_PARTS = 16     # GATMI_MAX_PARTS


class InAgg(C.Structure):
    _fields_ = [("shape", Shape), ("negative_slope", C.c_float), ("dropout_p", C.c_float),
                ("mask_mode", C.c_int32), ("stream", C.c_int32), ("seed", C.c_uint64),
                ("mask", C.c_void_p), ("xs", C.c_void_p), ("a_dst", C.c_void_p),
                ("weight", C.c_void_p), ("att_src", C.c_void_p), ("att_dst", C.c_void_p),
                ("bias", C.c_void_p), ("xs16", C.c_void_p)]


class Glue(C.Structure):
    _fields_ = [("rows", C.c_int64), ("cols", C.c_int32), ("training", C.c_int32),
                ("eps", C.c_float), ("momentum", C.c_float), ("dropout_p", C.c_float),
                ("seed_on_device", C.c_int32), ("seed", C.c_uint64), ("gamma", C.c_void_p),
                ("beta", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p)]


# name -> (restype, argtypes); every symbol include/gat_mi355.h declares
SYMBOLS = {
    "gatmi_last_error": (C.c_char_p, []),
    "gatmi_version": (C.c_int, []),
    "gatmi_prof_enable": (C.c_int, [C.c_int]),
    "gatmi_prof_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_float), C.c_int]),
    "gatmi_graph_sizes_query": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                          C.POINTER(GraphSizes)]),
    "gatmi_graph_build": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                    C.POINTER(Graph), C.c_void_p, C.c_size_t, C.c_void_p]),
    "gatmi_project_workspace_bytes": (C.c_size_t, [C.POINTER(Shape)]),
    "gatmi_project_fwd": (C.c_int, [C.POINTER(Shape), C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gatmi_project_bwd": (C.c_int, [C.POINTER(Shape), C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_size_t, C.c_void_p]),
    "gatmi_attn_fwd_workspace_bytes": (C.c_size_t, [C.POINTER(Graph), C.POINTER(Shape)]),
    "gatmi_attn_fwd": (C.c_int, [C.POINTER(Graph), C.POINTER(Attn), C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_size_t, C.c_void_p]),
    "gatmi_attn_fwd_static": (C.c_int, [C.POINTER(Graph), C.POINTER(Attn), C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "gatmi_dropout_mask": (C.c_int, [C.POINTER(Graph), C.c_int, C.c_float, C.c_uint64, C.c_void_p,
                                     C.c_void_p]),
    "gatmi_attn_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(Graph), C.POINTER(Shape)]),
    "gatmi_attn_bf16_supported": (C.c_int, [C.POINTER(Shape)]),
    "gatmi_project_bf16_supported": (C.c_int, [C.POINTER(Shape)]),
    "gatmi_project_bf16_workspace_bytes": (C.c_size_t, [C.POINTER(Shape)]),
    "gatmi_project_fwd_bf16": (C.c_int, [C.POINTER(Shape), C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gatmi_project_bwd_bf16": (C.c_int, [C.POINTER(Shape), C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_size_t, C.c_void_p]),
    "gatmi_attn_bwd_workspace_bytes_bf16": (C.c_size_t, [C.POINTER(Graph), C.POINTER(Shape)]),
    "gatmi_attn_bwd": (C.c_int, [C.POINTER(Graph), C.POINTER(Attn), C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gatmi_attn_fwd_partial": (C.c_int, [C.POINTER(Graph), C.POINTER(Attn), C.POINTER(AttnPartial),
                                         C.c_void_p, C.c_size_t, C.c_void_p]),
    "gatmi_attn_fwd_combine": (C.c_int, [C.POINTER(Shape), C.c_int, C.POINTER(AttnPartial), C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    "gatmi_attn_bwd_node_workspace_bytes": (C.c_size_t, [C.POINTER(Shape), C.c_int, C.c_int]),
    "gatmi_attn_bwd_chunk_workspace_bytes": (C.c_size_t, [C.POINTER(Graph), C.POINTER(Shape)]),
    "gatmi_attn_bwd_begin": (C.c_int, [C.POINTER(Attn), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_size_t, C.c_void_p]),
    "gatmi_attn_bwd_chunk": (C.c_int, [C.POINTER(Graph), C.POINTER(Attn), C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_size_t, C.c_void_p]),
    "gatmi_attn_bwd_end": (C.c_int, [C.POINTER(Attn), C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gatmi_add_dst_term": (C.c_int, [C.POINTER(Shape), C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "gatmi_glue_workspace_bytes": (C.c_size_t, [C.c_int]),
    "gatmi_glue_fwd": (C.c_int, [C.POINTER(Glue), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_size_t, C.c_void_p]),
    "gatmi_glue_bwd": (C.c_int, [C.POINTER(Glue), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                 C.c_void_p]),
    "gatmi_glue_stats": (C.c_int, [C.POINTER(Glue), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_size_t, C.c_void_p]),
    "gatmi_glue_apply": (C.c_int, [C.POINTER(Glue), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "gatmi_glue_bwd_sums": (C.c_int, [C.POINTER(Glue), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gatmi_glue_bwd_apply": (C.c_int, [C.POINTER(Glue), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gatmi_glue_mask": (C.c_int, [C.c_int64, C.c_int32, C.c_float, C.c_uint64, C.c_void_p,
                                  C.c_void_p]),
    "gatmi_tail_workspace_bytes": (C.c_size_t, []),
    "gatmi_tail_fwd": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_size_t, C.c_void_p]),
    "gatmi_tail_bwd": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    "gatmi_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "gatmi_knn_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "gatmi_knn_graph": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_size_t, C.c_void_p]),
    "gatmi_knn_graph_ties": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_size_t, C.c_void_p]),
    "gatmi_inagg_supported": (C.c_int, [C.POINTER(Shape)]),
    "gatmi_inagg_z_floats": (C.c_size_t, [C.POINTER(Shape)]),
    "gatmi_inagg_xs_floats": (C.c_size_t, [C.POINTER(Shape)]),
    "gatmi_inagg_xs16_bytes": (C.c_size_t, [C.POINTER(Shape)]),
    "gatmi_inagg_table16": (C.c_int, [C.POINTER(Shape), C.c_void_p, C.c_void_p, C.c_void_p]),
    "gatmi_inagg_pack_workspace_bytes": (C.c_size_t, [C.POINTER(Shape)]),
    "gatmi_inagg_pack": (C.c_int, [C.POINTER(Shape), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gatmi_inagg_pack16": (C.c_int, [C.POINTER(Shape), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gatmi_inagg_fwd_workspace_bytes": (C.c_size_t, [C.POINTER(Graph), C.POINTER(Shape)]),
    "gatmi_inagg_fwd": (C.c_int, [C.POINTER(Graph), C.POINTER(InAgg), C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gatmi_inagg_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(Graph), C.POINTER(Shape)]),
    "gatmi_inagg_bwd": (C.c_int, [C.POINTER(Graph), C.POINTER(InAgg), C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_size_t, C.c_void_p]),
}

_lib = None


def build_library(verbose: bool = False) -> str:
    """Compile libgat_mi355.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    subprocess.run(["make", "-C", os.path.join(_PKG, "csrc"), "-j4"] + ([] if verbose else ["-s"]),
                   check=True)
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C gat_amd/csrc`).  gat_amd has no fallback path.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib().gatmi_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {code}): {msg}")


def set_option(name: str, value: int) -> None:
    """Kernel-selection switch (gatmi_set_option): value 0/1, -1 hands it back to the environment."""
    check(lib().gatmi_set_option(name.encode(), int(value)), "gatmi_set_option")


def prof_enable(on: bool) -> None:
    lib().gatmi_prof_enable(int(on))


def prof_read(max_records: int = 65536):
    """-> list of (kernel tag, milliseconds) recorded since the last read."""
    tags = C.create_string_buffer(max_records * 32)
    ms = (C.c_float * max_records)()
    n = lib().gatmi_prof_read(tags, ms, max_records)
    raw = tags.raw
    return [(raw[i * 32:(i + 1) * 32].split(b"\0", 1)[0].decode(), float(ms[i])) for i in range(n)]
