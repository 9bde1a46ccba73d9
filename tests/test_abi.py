"""CPU tests of the boundary: libgat_mi355.so loads without a GPU and exports every
symbol include/gat_mi355.h declares; the ctypes mirrors have the C layout.  No compute
entry point is called here."""
import ctypes
import os
import re

import pytest

from gat_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gat_mi355.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gatmi_[a-z0-9_]+)\s*\(", text)))


def test_header_lists_entry_points():
    names = declared_symbols()
    for must in ("gatmi_graph_build", "gatmi_project_fwd", "gatmi_project_bwd", "gatmi_attn_fwd",
                 "gatmi_attn_bwd", "gatmi_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(handle, name), f"{name} declared in the header but not exported"


def test_binding_covers_header():
    assert sorted(_lib.SYMBOLS) == declared_symbols()


def test_struct_layouts_match_c():
    assert ctypes.sizeof(_lib.Shape) == 16
    assert ctypes.sizeof(_lib.Plan) == 16 + 5 * 8
    assert ctypes.sizeof(_lib.Graph) == 24 + 7 * 8 + 2 * ctypes.sizeof(_lib.Plan)
    assert ctypes.sizeof(_lib.GraphSizes) == 32
    assert ctypes.sizeof(_lib.Attn) == 16 + 16 + 8 + 6 * 8
    assert _lib.Attn.hp.offset == 48


def test_version_and_error_string_callable_without_gpu():
    lib = _lib.lib()
    assert lib.gatmi_version() >= 100
    assert isinstance(lib.gatmi_last_error(), bytes)


def test_argument_errors_are_reported_not_crashed():
    lib = _lib.lib()
    sizes = _lib.GraphSizes()
    rc = lib.gatmi_graph_sizes_query(10, 5, 5, 1, 100, ctypes.byref(sizes))  # seg_len not % 64
    assert rc == 1 and b"seg_len" in lib.gatmi_last_error()
    with pytest.raises(RuntimeError, match="seg_len"):
        _lib.check(rc, "gatmi_graph_sizes_query")


def test_kernel_switches_documented_and_known_to_the_library():
    """Every A/B switch DESIGN.md names is one gatmi_set_option accepts (no GPU needed: the
    call only stores a value), and an unknown name is an argument error."""
    import os
    import re
    design = open(os.path.join(os.path.dirname(__file__), "..", "DESIGN.md")).read()
    names = sorted(set(re.findall(r"`(GATMI_[A-Z0-9_]+)=", design)) - {"GATMI_LIB", "GATMI_EXACT_EXP", "GATMI_RING"})
    assert len(names) >= 12, names
    for name in names:
        _lib.set_option(name, 0)
        _lib.set_option(name, -1)
    with pytest.raises(RuntimeError):
        _lib.set_option("GATMI_NO_SUCH_SWITCH", 1)


def test_layer_refuses_cpu_tensors():
    import torch
    from gat_amd import GATConv
    conv = GATConv(4, 3, heads=2)
    assert sorted(conv.state_dict()) == ["att_dst", "att_src", "bias", "lin.weight"]
    with pytest.raises(RuntimeError, match="GPU only"):
        conv(torch.randn(5, 4), torch.zeros(2, 3, dtype=torch.long))
