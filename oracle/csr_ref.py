"""ctypes face of oracle/csr_ref.c (plain-C restatement of the graph-layout step).
TEST INFRASTRUCTURE ONLY -- see the header of gat_oracle.py."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

from .gat_oracle import CSRLayout

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcsr_ref.so")
_lib = None


def build() -> str:
    subprocess.run(["make", "-s", "-C", _HERE, "libcsr_ref.so"], check=True)
    return _SO


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        lib = ctypes.CDLL(_SO)
        p64 = ctypes.POINTER(ctypes.c_int64)
        p32 = ctypes.POINTER(ctypes.c_int32)
        lib.csr_ref_count.restype = ctypes.c_int64
        lib.csr_ref_count.argtypes = [p64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
        lib.csr_ref_build.restype = ctypes.c_int64
        lib.csr_ref_build.argtypes = [p64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                      p64, p64, p32, p32, p32, p32, p32, p32]
        _lib = lib
    return _lib


def build_csr_c(edge_index: np.ndarray, num_nodes: int, add_self_loops: bool = True):
    """edge_index: int64 [2, E] (C contiguous).  Returns (CSRLayout, norm_edge_index)."""
    lib = _load()
    ei = np.ascontiguousarray(edge_index, dtype=np.int64)
    e = ei.shape[1]
    p64 = ctypes.POINTER(ctypes.c_int64)
    p32 = ctypes.POINTER(ctypes.c_int32)
    ep = lib.csr_ref_count(ei.ctypes.data_as(p64), e, num_nodes, int(add_self_loops))
    if ep < 0:
        raise ValueError("edge_index holds a node id outside [0, num_nodes)")
    norm = np.empty((2, ep), dtype=np.int64)
    i32 = lambda n: np.empty(n, dtype=np.int32)
    rowptr, col, perm = i32(num_nodes + 1), i32(ep), i32(ep)
    rowptr_t, col_t, pos_t = i32(num_nodes + 1), i32(ep), i32(ep)
    got = lib.csr_ref_build(ei.ctypes.data_as(p64), e, num_nodes, int(add_self_loops),
                            norm[0].ctypes.data_as(p64), norm[1].ctypes.data_as(p64),
                            rowptr.ctypes.data_as(p32), col.ctypes.data_as(p32),
                            perm.ctypes.data_as(p32), rowptr_t.ctypes.data_as(p32),
                            col_t.ctypes.data_as(p32), pos_t.ctypes.data_as(p32))
    assert got == ep
    return CSRLayout(num_nodes, int(ep), rowptr, col, perm, rowptr_t, col_t, pos_t), norm
