"""2-rank gloo rehearsal of PartitionedGAT on ONE GPU (both ranks on cuda:0, HipBackend): log-probabilities,
loss and parameter gradients of one training step against the single-process oracle network.  Not a
pytest (two GPU processes; the file name keeps it out of collection); run as
python tests/rehearse_gloo2_model.py  on a GPU box.  Lives under tests/ because it checks against the oracle."""
import os, sys
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cases
        from oracle import gat_oracle as O
        from gat_amd.dist import PartitionedGAT
        dev = torch.device("cuda:0")
        ei, n = cases.knn_graph(n=3000, f=25, seed=2)
        g = torch.Generator().manual_seed(3)
        x = torch.rand(n, 25, generator=g)
        y = torch.randint(0, 2, (n,), generator=g)
        torch.manual_seed(7)
        ref = O.OracleGAT(25, 2, hidden_dim=30, dropout=0.0)
        model = PartitionedGAT(25, 2, hidden_dim=30, dropout=0.0).to(dev)
        model.load_state_dict(ref.state_dict())
        ref.train(); ref_logp = ref(x, ei); ref_loss = torch.nn.functional.nll_loss(ref_logp, y); ref_loss.backward()
        model.train()
        eid = ei.to(dev)
        part = model.partition_for(eid, n)
        # gloo moves CPU tensors: collectives on CUDA tensors go through gloo's device copy path
        logp = model(x[part.lo:part.hi].to(dev), eid, n)
        share = model.loss_share(logp, y[part.lo:part.hi].to(dev), n)
        share.backward(); model.sync_grads()
        err = float((logp.detach().cpu() - ref_logp.detach()[part.lo:part.hi]).abs().max())
        lerr = abs(float(model.global_loss(share)) - float(ref_loss.detach()))
        rp = dict(ref.named_parameters())
        gerr = max(float((p.grad.cpu() - rp[k].grad).abs().max() / (rp[k].grad.abs().max() + 1e-12))
                   for k, p in model.named_parameters() if not k.endswith("conv1.bias") and not k.endswith("conv2.bias"))
        q.put((rank, f"logp err {err:.2e}  loss err {lerr:.2e}  worst relative grad err {gerr:.2e}"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAILED " + "".join(traceback.format_exception(type(e), e, e.__traceback__))))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, 29641, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    for rank, msg in res:
        print(f"rank {rank}: {msg}", flush=True)
    sys.exit(1 if any("FAILED" in m for _, m in res) else 0)
