"""BASELINE.json config 4 on the real kernels: a sweep's trials run concurrently, one process per
rank, no collective on the data path (gat_amd/trials.py; /root/reference/grid_search.py:70-118 is
the sequential original).  A one-GPU box has one card, so the two ranks share cuda:0 -- what is
exercised is everything but the device ordinal: the dynamic hand-out of configurations, two
trainings running at once on the HIP path, per-rank checkpoint files, the gather of the results.

The worker processes are spawned BEFORE this process touches the GPU (a process that has
initialised the GPU must not be the one that starts them: the file name makes this the first GPU
test collected, and the test skips itself when the GPU is already initialised)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rank_main(rank, world, port, workdir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      LOCAL_RANK=str(rank), WORLD_SIZE=str(world), GAT_SYNTH_ROWS="2000", GAT_QUIET="1")
    os.chdir(workdir)
    try:
        import torch.distributed as dist
        from gat_amd import _lib, trials
        torch.manual_seed(100 + rank)
        configs = [trials.Config(torch.optim.AdamW, lr, 4.8e-4, epochs, 5, hidden, 0.425)
                   for lr, epochs, hidden in ((0.0425, 6, 30), (0.04, 3, 28), (0.035, 8, 32),
                                              (0.0325, 4, 34), (0.0375, 5, 30))]
        ran = []

        def experiment(cfg):
            ran.append(cfg.lr)
            return trials.run_experiment(cfg, num_runs=1)

        _lib.prof_enable(True)
        res = trials.run_trials(configs, experiment, schedule="dynamic")
        tags = {t for t, _ in _lib.prof_read()}
        left = [f for f in os.listdir(".") if f.startswith("best_model")]
        q.put((rank, ran, [(r.config.lr, r.composite_score, r.avg_val_accuracy, r.avg_val_loss,
                            len(r.metrics[0].epoch_values)) for r in res], sorted(tags), left,
               torch.cuda.current_device()))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAILED " + "".join(traceback.format_exception(type(e), e, e.__traceback__))))


def test_sweep_of_five_trials_on_two_concurrent_ranks(tmp_path):
    import torch.multiprocessing as mp
    if torch.cuda.device_count() == 0:
        pytest.skip("no GPU")
    if torch.cuda.is_initialized():
        pytest.skip("this process has already initialised the GPU: run this file first / alone")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, 29671, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=280) for _ in procs)
    for p in procs:
        p.join(60)
    for item in got:
        assert not (isinstance(item[1], str) and item[1].startswith("FAILED")), item[1]
    lrs = [0.0425, 0.04, 0.035, 0.0325, 0.0375]
    assert sorted(got[0][1] + got[1][1]) == sorted(lrs)        # every configuration trained exactly once
    assert got[0][1] and got[1][1]                              # and both ranks trained some
    assert got[0][2] == got[1][2]                               # every rank holds the full, ordered result list
    for (lr, score, acc, loss, epochs), want_lr, max_epochs in zip(got[0][2], lrs, (6, 3, 8, 4, 5)):
        assert lr == want_lr and 1 <= epochs <= max_epochs
        # a 3-epoch trial may still sit on the minority class: the sweep is what is under test, not the
        # quality of every configuration
        assert 0.0 < score <= 1.0 and 0.0 <= acc <= 1.0 and 0.0 < loss < 10.0
    assert max(r[2] for r in got[0][2]) > 0.7
    for _, _, _, tags, left, _ in got:
        # the HIP path trained them: conv1 (the fused tiles on this 5-neighbour graph) / conv2-3
        assert ("inagg_fwd_fused" in tags or "inagg_fwd" in tags) and "attn_bwd_edge" in tags
        assert left == []                                       # per-rank checkpoints are cleaned up
