"""World-1 RCCL rehearsal on one MI355X: the "nccl" backend (RCCL on ROCm) cannot hold two ranks on one
card, but a single-rank communicator still runs every collective FORM the partitioned path issues
(all_gather_into_tensor, reduce_scatter_tensor -- which gloo does not have, so the gloo rehearsals never
reach that branch of gat_amd/dist.py --, all_reduce, barrier) with RCCL's own argument checks.  One training
step of PartitionedGAT through RCCL against the single-process oracle network, then each exchange mode of
PartitionedGATConv against GATConv.  Run as a child process (tests/test_gpu_rccl_world1.py) or by hand:
python tests/rehearse_rccl_world1.py.  Prints one line per check and exits non-zero on a failure."""
import os, sys
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29677")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    ok = True
    try:
        import cases
        from oracle import gat_oracle as O
        from gat_amd import GATConv
        from gat_amd.dist import PartitionedGAT, PartitionedGATConv
        assert dist.get_backend() == "nccl"
        # the whole network, one training step
        ei, n = cases.knn_graph(n=3000, f=25, seed=2)
        g = torch.Generator().manual_seed(3)
        x = torch.rand(n, 25, generator=g)
        y = torch.randint(0, 2, (n,), generator=g)
        torch.manual_seed(7)
        ref = O.OracleGAT(25, 2, hidden_dim=30, dropout=0.0)
        model = PartitionedGAT(25, 2, hidden_dim=30, dropout=0.0).to(dev)
        model.load_state_dict(ref.state_dict())
        ref.train()
        ref_logp = ref(x, ei)
        ref_loss = torch.nn.functional.nll_loss(ref_logp, y)
        ref_loss.backward()
        model.train()
        eid = ei.to(dev)
        part = model.partition_for(eid, n)
        assert model.fused          # ELU/BN/dropout on glue.hip, log_softmax + nll on tail.hip
        logp, share = model.forward_loss(x[part.lo:part.hi].to(dev), eid, n, y[part.lo:part.hi].to(dev))
        share.backward()
        model.sync_grads()
        err = float((logp.detach().cpu() - ref_logp.detach()).abs().max())
        lerr = abs(float(model.global_loss(share)) - float(ref_loss.detach()))
        rp = dict(ref.named_parameters())
        gerr = max(float((p.grad.cpu() - rp[k].grad).abs().max() / (rp[k].grad.abs().max() + 1e-12))
                   for k, p in model.named_parameters()
                   if not k.endswith("conv1.bias") and not k.endswith("conv2.bias"))
        good = err < 1e-4 and lerr < 1e-5 and gerr < 1e-3
        ok &= good
        print(f"{'ok' if good else 'FAILED'} PartitionedGAT over RCCL: logp err {err:.2e} loss err {lerr:.2e} "
              f"worst relative grad err {gerr:.2e}", flush=True)
        # the same network in bf16 mode with dropout: one training step runs and stays finite
        torch.manual_seed(7)
        m16 = PartitionedGAT(25, 2, hidden_dim=32, dropout=0.3, feature_dtype=torch.bfloat16).to(dev).train()
        lp16, sh16 = m16.forward_loss(x[part.lo:part.hi].to(dev), eid, n, y[part.lo:part.hi].to(dev))
        sh16.backward()
        m16.sync_grads()
        good = bool(torch.isfinite(lp16).all()) and all(bool(torch.isfinite(p.grad).all()) for p in m16.parameters())
        good &= 0.3 < float(m16.global_loss(sh16)) < 2.0
        ok &= good
        print(f"{'ok' if good else 'FAILED'} PartitionedGAT bf16 mode + dropout over RCCL: loss {float(sh16):.4f}", flush=True)
        # every exchange mode of one layer against the single-GPU layer
        bf = torch.bfloat16
        # the last four: exchange "h" in pieces (asynchronous all_gather_into_tensor / reduce_scatter_tensor
        # per piece with the attention of the arrived pieces in between), and on exact tables (per-owner
        # broadcast / reduce)
        for mode, f, h, c, xg, dt, chunks, padding in (
                ("h", 64, 4, 16, True, None, 1, "auto"), ("x", 12, 4, 16, True, None, 1, "auto"),
                ("xs", 12, 4, 16, False, None, 1, "auto"), ("as", 12, 4, 16, False, None, 1, "auto"),
                ("h", 256, 4, 64, True, None, 1, "auto"), ("xs", 100, 4, 64, False, bf, 1, "auto"),
                ("as", 100, 4, 64, False, bf, 1, "auto"), ("as", 100, 8, 32, False, bf, 1, "auto"),
                ("h", 64, 4, 16, True, None, 3, "pad"), ("h", 256, 4, 64, True, None, 4, "pad"),
                ("h", 256, 8, 32, True, None, 2, "exact"), ("x", 12, 4, 16, True, None, 1, "exact")):
            torch.manual_seed(11)
            # bf16 rows: both sides on the input-side path (GATConv alone would pick by a cost model)
            one = GATConv(f, c, heads=h, dropout=0.0, feature_dtype=dt,
                          input_aggregation=True if dt is not None else None).to(dev)
            many = PartitionedGATConv(f, c, heads=h, dropout=0.0, exchange=mode, feature_dtype=dt,
                                      chunks=chunks, padding=padding).to(dev)
            many.load_state_dict(one.state_dict())
            xa = torch.randn(n, f, device=dev, requires_grad=xg)
            xb = xa.detach().clone().requires_grad_(xg)
            go = torch.randn(n, h * c, device=dev)
            oa = one(xa, eid)
            oa.backward(go)
            if dt is not None:        # a first call fills the table caches; the compared one is the second
                many(xb, eid, n)
            ob = many(xb, eid, n)
            ob.backward(go)
            e_out = float((oa.detach() - ob.detach()).abs().max() / oa.detach().abs().max())
            e_w = float((one.lin.weight.grad - many.lin.weight.grad).abs().max() / one.lin.weight.grad.abs().max())
            e_x = float((xa.grad - xb.grad).abs().max() / xa.grad.abs().max()) if xg else 0.0
            good = e_out < 1e-5 and e_w < 1e-4 and e_x < 1e-4
            ok &= good
            print(f"{'ok' if good else 'FAILED'} exchange={mode} {f}->{h}x{c}{' bf16 rows' if dt is not None else ''}"
                  f"{f' in {chunks} pieces' if chunks > 1 else ''}{' exact tables' if padding == 'exact' else ''}: "
                  f"out {e_out:.1e} d_W {e_w:.1e} d_x {e_x:.1e}", flush=True)
        # a bf16 table in pieces against the same table in one piece
        torch.manual_seed(13)
        one = PartitionedGATConv(256, 64, heads=4, dropout=0.0, exchange="h", feature_dtype=bf, chunks=1).to(dev)
        many = PartitionedGATConv(256, 64, heads=4, dropout=0.0, exchange="h", feature_dtype=bf, chunks=3).to(dev)
        many.load_state_dict(one.state_dict())
        xa = torch.randn(n, 256, device=dev, requires_grad=True)
        xb = xa.detach().clone().requires_grad_(True)
        go = torch.randn(n, 256, device=dev)
        oa = one(xa, eid, n); oa.backward(go)
        ob = many(xb, eid, n); ob.backward(go)
        e_out = float((oa.detach() - ob.detach()).abs().max() / oa.detach().abs().max())
        e_x = float((xa.grad - xb.grad).norm() / xa.grad.norm())
        good = e_out < 1e-5 and e_x < 2e-3
        ok &= good
        print(f"{'ok' if good else 'FAILED'} exchange=h 256->4x64 bf16 table in 3 pieces vs 1: out {e_out:.1e} "
              f"d_x (relative L2) {e_x:.1e}", flush=True)
        dist.barrier()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
