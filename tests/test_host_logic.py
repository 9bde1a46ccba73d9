"""CPU tests of the host side that mirrors the reference's interfaces: metrics arithmetic
against scikit-learn (what model.py:61-63,76-79 call), the Data container, and trial
sharding one-per-rank under gloo (config 4) with a stub experiment."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gat_amd.data import Data, Metrics, convert_to_graph, synthetic_traces
from gat_amd.model import confusion_counts, weighted_prf


@pytest.mark.parametrize("k,drop_last", [(2, False), (2, True), (4, True)])
def test_weighted_metrics_equal_sklearn(k, drop_last):
    from sklearn.metrics import confusion_matrix, f1_score, precision_score, recall_score
    rng = np.random.default_rng(k)
    y = rng.integers(0, k, 700)
    p = rng.integers(0, k - 1 if drop_last else k, 700)   # a class that is never predicted
    cm = confusion_counts(torch.tensor(y), torch.tensor(p), k)
    assert np.array_equal(cm.numpy(), confusion_matrix(y, p, labels=list(range(k))))
    got = weighted_prf(cm)
    want = (precision_score(y, p, average="weighted", zero_division=0),
            recall_score(y, p, average="weighted", zero_division=0),
            f1_score(y, p, average="weighted", zero_division=0))
    np.testing.assert_allclose(got, want, rtol=1e-12)


def test_convert_to_graph_matches_reference_layout():
    """converter.py:7-15: int64 [2, 5N], row 0 = query node repeated 5x, ascending."""
    X, y = synthetic_traces(200, 25, seed=1)
    d = convert_to_graph(X, y)
    assert isinstance(d, Data) and d.num_features == 25 and d.num_nodes == 200
    assert d.edge_index.dtype == torch.int64 and tuple(d.edge_index.shape) == (2, 1000)
    assert torch.equal(d.edge_index[0], torch.arange(200).repeat_interleave(5))
    assert d.x.dtype == torch.float32 and d.y.dtype == torch.int64


def test_metrics_container_fields():
    m = Metrics()
    assert list(m.__dict__) == ["epoch_values", "train_loss", "val_loss", "train_accuracy",
                                "val_accuracy", "train_precision", "train_recall", "train_f1",
                                "val_precision", "val_recall", "val_f1"]


def test_shim_import_paths():
    from gat.converter import convert_to_graph as c2g
    from gat.data_models import Metrics as M
    from gat.model import GAT
    assert c2g is convert_to_graph and M is Metrics and GAT.__name__ == "GAT"


def _dynamic_worker(rank, world, port, q):
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    from gat_amd import trials
    configs = [trials.Config(torch.optim.AdamW, 0.01 * i, 5e-4, 3, 2, 16, 0.4) for i in range(7)]
    ran = []

    def fake(cfg):
        ran.append(cfg.lr)
        time.sleep(0.6 if cfg.lr == 0.0 else 0.05)      # the first trial is a long one
        return trials.ConfigResults(cfg, [Metrics()], cfg.lr, 1.0, 1.0, cfg.lr, cfg.lr, cfg.lr, cfg.lr)

    for _ in range(2):                                   # two searches in one process group
        del ran[:]
        res = trials.run_trials(configs, fake, schedule="dynamic")
    q.put((rank, list(ran), [r.composite_score for r in res], [r.config.lr for r in res]))
    dist.destroy_process_group()


def test_trials_dynamic_queue_hands_out_each_configuration_once():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dynamic_worker, args=(r, 2, 29623, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    want = [0.01 * i for i in range(7)]
    assert sorted(got[0][1] + got[1][1]) == want           # every configuration ran exactly once
    long_owner = got[0] if 0.0 in got[0][1] else got[1]
    assert len(long_owner[1]) < 4                           # the rank stuck on the long trial took fewer
    for _, _, scores, lrs in got:
        assert scores == want and lrs == want


def _trial_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    from gat_amd import trials
    configs = [trials.Config(torch.optim.AdamW, 0.01 * i, 5e-4, 3, 2, 16, 0.4) for i in range(5)]
    ran = []

    def fake(cfg):
        ran.append(cfg.lr)
        return trials.ConfigResults(cfg, [Metrics()], cfg.lr, 1.0, 1.0, cfg.lr, cfg.lr, cfg.lr, cfg.lr)

    res = trials.run_trials(configs, fake)
    q.put((rank, ran, [r.composite_score for r in res], [r.config.lr for r in res]))
    dist.destroy_process_group()


def test_trials_shard_one_per_rank_and_merge_in_order():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_trial_worker, args=(r, 2, 29621, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    want = [0.01 * i for i in range(5)]
    assert got[0][1] == want[0::2] and got[1][1] == want[1::2]     # disjoint round-robin shards
    for _, _, scores, lrs in got:                                   # every rank sees all, in order
        assert scores == want and lrs == want
