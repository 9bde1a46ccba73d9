"""Two REAL ranks on one MI355X: both processes use cuda:0 and the HIP kernels, the collectives go
through gloo (RCCL cannot hold two ranks on one card).  What world-1 RCCL (rehearse_rccl_world1.py)
and the one-process "one GPU plays every rank" tests cannot show is shown here: the partition, the
piece tables and the asynchronous exchange with a peer that really is another process -- every exchange
mode of PartitionedGATConv against the single-GPU GATConv on the whole graph, and one training step of
PartitionedGAT against the oracle network.  Started by tests/test_a2_gpu_gloo_world2.py as
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tests/rehearse_gloo2_gpu.py
Rank 0 prints one line per check (the verdict is the MIN over the ranks); exit code 1 on a failure."""
import os, sys
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")
    all_ok = True

    def verdict(good, text):
        """every rank's verdict -> one line on rank 0"""
        nonlocal all_ok
        flag = torch.tensor([1 if good else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        good = bool(flag.item())
        all_ok &= good
        if rank == 0:
            print(f"{'ok' if good else 'FAILED'} {text}", flush=True)

    try:
        import cases
        from oracle import gat_oracle as O
        from gat_amd import GATConv, synth
        from gat_amd.dist import PartitionedGAT, PartitionedGATConv, make_partition
        # ---- one layer, every exchange mode, against the single-GPU layer on the whole graph
        n = 6000
        ei = synth.powerlaw_graph(n, 8 * n, 900, seed=4)
        eid = ei.to(dev)
        bf = torch.bfloat16
        for mode, f, h, c, xg, dt, chunks, padding in (
                ("h", 64, 4, 16, True, None, 1, "auto"), ("h", 256, 4, 64, True, None, 1, "pad"),
                ("h", 64, 4, 16, True, None, 3, "pad"), ("h", 256, 4, 64, True, None, 4, "pad"),
                ("h", 256, 8, 32, True, None, 2, "exact"), ("h", 100, 3, 20, True, None, 4, "exact"),
                ("x", 12, 4, 16, True, None, 1, "auto"), ("xs", 12, 4, 16, False, None, 1, "auto"),
                ("as", 100, 4, 64, False, None, 1, "auto"), ("as", 25, 8, 30, False, None, 1, "exact"),
                ("as", 100, 4, 64, False, bf, 1, "auto"), ("xs", 100, 4, 64, False, bf, 1, "auto")):
            torch.manual_seed(11)
            one = GATConv(f, c, heads=h, dropout=0.0, feature_dtype=dt,
                          input_aggregation=True if dt is not None else None).to(dev)
            many = PartitionedGATConv(f, c, heads=h, dropout=0.0, exchange=mode, feature_dtype=dt,
                                      chunks=chunks, padding=padding).to(dev)
            many.load_state_dict(one.state_dict())
            g = torch.Generator().manual_seed(5)
            xa = torch.randn(n, f, generator=g).to(dev).requires_grad_(xg)
            go = torch.randn(n, h * c, generator=g).to(dev)
            part = make_partition(eid, n, world, rank, "edges", padding)
            xb = xa.detach()[part.lo:part.hi].clone().requires_grad_(xg)
            oa = one(xa, eid)
            oa.backward(go)
            if dt is not None:        # a first call fills the table caches; the compared one is the second
                many(xb, eid, n)
            ob = many(xb, eid, n)
            ob.backward(go[part.lo:part.hi].contiguous())
            mine = slice(part.lo, part.hi)
            e_out = float((oa.detach()[mine] - ob.detach()).abs().max() / oa.detach().abs().max())
            e_w = float((one.lin.weight.grad - many.lin.weight.grad).abs().max() / one.lin.weight.grad.abs().max())
            e_a = float((one.att_src.grad - many.att_src.grad).abs().max() / one.att_src.grad.abs().max())
            e_x = float((xa.grad[mine] - xb.grad).abs().max() / xa.grad.abs().max()) if xg else 0.0
            good = ob.shape[0] == part.n_local and e_out < 1e-5 and e_w < 1e-4 and e_a < 1e-3 and e_x < 1e-4
            verdict(good, f"exchange={mode} {f}->{h}x{c}{' bf16 rows' if dt is not None else ''}"
                          f"{f' in {chunks} pieces' if chunks > 1 else ''}{' exact tables' if part.uneven else ''}"
                          f" (rank {rank}: rows {part.lo}..{part.hi}): out {e_out:.1e} d_W {e_w:.1e} "
                          f"d_att_src {e_a:.1e} d_x {e_x:.1e}")
        # ---- the network: one training step of PartitionedGAT against the oracle network
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
        logp, share = model.forward_loss(x[part.lo:part.hi].to(dev), eid, n, y[part.lo:part.hi].to(dev))
        share.backward()
        model.sync_grads()
        err = float((logp.detach().cpu() - ref_logp.detach()[part.lo:part.hi]).abs().max())
        lerr = abs(float(model.global_loss(share)) - float(ref_loss.detach()))
        rp = dict(ref.named_parameters())
        gerr = max(float((p.grad.cpu() - rp[k].grad).abs().max() / (rp[k].grad.abs().max() + 1e-12))
                   for k, p in model.named_parameters()
                   if not k.endswith("conv1.bias") and not k.endswith("conv2.bias"))
        verdict(model.fused and err < 1e-4 and lerr < 1e-5 and gerr < 1e-3,
                f"PartitionedGAT on two ranks (fused glue, synced BatchNorm): logp err {err:.2e} loss err {lerr:.2e} "
                f"worst relative grad err {gerr:.2e}")
        torch.cuda.synchronize()
        dist.barrier()
    finally:
        dist.destroy_process_group()
    return 0 if all_ok else 1


if __name__ == "__main__":
    sys.exit(main())
