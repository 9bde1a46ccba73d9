"""gatmi_attn_fwd_static -- the forward without a workspace argument, its scratch in a device
array of the library (include/gat_mi355.h) -- against gatmi_attn_fwd, the entry point it is a
variant of, through the C ABI.  Same kernels, so the bar is bit-exact.  Two calls in a row, the
second on a smaller input: the array keeps the first call's contents and nothing may depend
on them.  Collected last (file name) so that a failure here hides no other test."""
import ctypes as C

import pytest
import torch

import cases

pytestmark = pytest.mark.gpu


def hub_graph(n, hub_deg, e_rand, seed):
    """random edges plus three destinations of in-degree hub_deg (rows the layout cuts into
    segments: the only users of the forward's scratch)."""
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, e_rand), generator=g)
    hubs = [torch.stack([torch.randint(0, n, (hub_deg,), generator=g),
                         torch.full((hub_deg,), d, dtype=torch.long)]) for d in (0, n // 2, n - 1)]
    return torch.cat([ei] + hubs, dim=1)


def both_forwards(cuda, n, hub_deg, e_rand, f, h, c, p, seg, seed):
    from gat_amd import _lib, ops
    from gat_amd.graph import build_graph
    lib = _lib.lib()
    ei = hub_graph(n, hub_deg, e_rand, seed)
    layout = build_graph(ei.to(cuda), n, seg_len=seg)
    assert layout.desc.plan.n_slots >= 3 * (hub_deg // seg), "the case must use the scratch"
    x, w, a_s, a_d, b, _ = cases.layer_inputs(n, f, h, c, seed)
    hp, a_dst = ops.project_fwd(x.to(cuda), w.to(cuda), a_d.view(-1).to(cuda), h, c)
    cfg = ops.AttnConfig(h, c, 0.2, p, _lib.MASK_HASH if p > 0 else _lib.MASK_NONE, seed=77 + seed)
    params = (a_s.view(-1).to(cuda), a_d.view(-1).to(cuda), b.to(cuda))   # alive while `a` points at them
    a = cfg.c_args(n, f, hp, a_dst, *params)
    st = torch.cuda.current_stream().cuda_stream
    need = lib.gatmi_attn_fwd_workspace_bytes(layout.ref, C.byref(a.shape))
    ws = torch.empty(need, dtype=torch.uint8, device=cuda)
    out_ref = torch.full((n, h * c), float("nan"), device=cuda)
    lse_ref = torch.full((n, h), float("nan"), device=cuda)
    _lib.check(lib.gatmi_attn_fwd(layout.ref, C.byref(a), out_ref.data_ptr(), lse_ref.data_ptr(),
                                  ws.data_ptr(), ws.numel(), st), "gatmi_attn_fwd")
    out = torch.full((n, h * c), float("nan"), device=cuda)
    lse = torch.full((n, h), float("nan"), device=cuda)
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats(cuda)["allocation.all.allocated"]
    _lib.check(lib.gatmi_attn_fwd_static(layout.ref, C.byref(a), out.data_ptr(), lse.data_ptr(), st),
               "gatmi_attn_fwd_static")
    assert torch.cuda.memory_stats(cuda)["allocation.all.allocated"] == before
    torch.cuda.synchronize()
    del params
    return out, lse, out_ref, lse_ref, need


def test_static_scratch_forward_equals_workspace_forward_twice(cuda):
    # first call: 4x64 layer, hubs of 3000 in-edges cut at 64 -> ~140 segments in the scratch
    out, lse, out_ref, lse_ref, need1 = both_forwards(cuda, 5000, 3000, 40000, 48, 4, 64, 0.3, 64, 0)
    assert torch.isfinite(out).all() and torch.equal(out, out_ref) and torch.equal(lse, lse_ref)
    # second call, smaller input and another layer shape: the scratch still holds call 1's partials
    out, lse, out_ref, lse_ref, need2 = both_forwards(cuda, 700, 400, 3000, 25, 8, 30, 0.0, 128, 1)
    assert need2 < need1
    assert torch.isfinite(out).all() and torch.equal(out, out_ref) and torch.equal(lse, lse_ref)


def test_static_scratch_rejects_what_does_not_fit(cuda):
    from gat_amd import _lib, ops
    from gat_amd.graph import build_graph
    lib = _lib.lib()
    n, h, c = 12000, 4, 256              # Dp = 1024: 4 KiB of partials per segment
    g = torch.Generator().manual_seed(3)
    ei = torch.stack([torch.randint(0, n, (200000,), generator=g),
                      torch.randint(0, 16, (200000,), generator=g)])   # 16 hubs of ~12500 in-edges
    layout = build_graph(ei.to(cuda), n, seg_len=64)
    hp = torch.zeros(n, h * c, device=cuda)
    a_dst = torch.zeros(n, h, device=cuda)
    att = torch.zeros(h * c, device=cuda)
    cfg = ops.AttnConfig(h, c)
    a = cfg.c_args(n, 8, hp, a_dst, att, att, None)
    need = lib.gatmi_attn_fwd_workspace_bytes(layout.ref, C.byref(a.shape))
    assert need > (8 << 20)
    out = torch.zeros(n, h * c, device=cuda)
    rc = lib.gatmi_attn_fwd_static(layout.ref, C.byref(a), out.data_ptr(), None,
                                   torch.cuda.current_stream().cuda_stream)
    assert rc == 4 and b"static array holds" in lib.gatmi_last_error()
