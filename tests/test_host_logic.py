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


# ------------------------------------------------------------------ search drivers (host logic)
def test_monte_carlo_draws_follow_the_reference_ranges():
    import monte_carlo_search as mc
    a, b = mc.sample_configs(50, seed=3), mc.sample_configs(50, seed=3)
    assert [c.__dict__ for c in a] == [c.__dict__ for c in b]            # every rank draws the same list
    assert a[0].__dict__ != mc.sample_configs(1, seed=4)[0].__dict__
    for c in a:                                                          # monte_carlo_search.py:154-162
        assert 0.03 <= c.lr <= 0.05 and 4.5e-4 <= c.weight_decay <= 6e-4 and 0.4 <= c.dropout <= 0.45
        assert c.epochs in (25, 35) and c.patience in (3, 7) and c.hidden_dim in (24, 40)
        assert c.optimizer is torch.optim.AdamW
    assert len({c.lr for c in a}) == 50                                  # continuous, not a grid


def test_evolutionary_search_breeds_as_the_reference_does():
    import random
    import evolutionary_computing_search as ev
    rng = random.Random(0)
    pop = ev.create_initial_population(10, ev.RANGES, rng)
    assert len(pop) == 10 and all(c.lr in ev.RANGES["lr"] and c.dropout in ev.RANGES["dropout"] for c in pop)
    a, b = pop[0], pop[1]
    for child in ev.crossover_configs(a, b, rng):
        for name in ev.FIELDS:
            assert getattr(child, name) in (getattr(a, name), getattr(b, name))
    m = ev.mutate_config(ev.Config(**a.__dict__), ev.RANGES, rng, rate=1.0)   # every field redrawn, in range
    assert m.lr in ev.RANGES["lr"] and 0.00046 <= m.weight_decay <= 0.00052 and m.optimizer is a.optimizer
    scored = [(i / 10.0, c) for i, c in enumerate(pop)]
    nxt = ev.next_population(scored, ev.RANGES, rng)
    assert len(nxt) == 15                                                # 5 survivors, 2 children each (:174-183)
    assert [nxt[i] for i in (0, 3, 6, 9, 12)] == [pop[i] for i in (9, 8, 7, 6, 5)]   # best first

    seen = []

    def evaluate(population):           # score: prefers wide layers and lr 0.04
        seen.append(len(population))
        return [c.hidden_dim / 36.0 - abs(c.lr - 0.04) for c in population]

    best, score = ev.evolutionary_search(ev.RANGES, evaluate, seed=1)
    best2, score2 = ev.evolutionary_search(ev.RANGES, evaluate, seed=1)
    assert seen[:6] == [10, 15, 15, 15, 15, 15]                          # one training per population
    assert best.__dict__ == best2.__dict__ and score == score2
    assert best.hidden_dim == 36 and score > 0.99


def _evo_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import evolutionary_computing_search as ev
    from gat_amd import trials
    ran = []

    def fake(cfg):
        ran.append(cfg.lr)
        s = cfg.hidden_dim / 36.0 - abs(cfg.lr - 0.04)
        return trials.ConfigResults(cfg, [Metrics()], s, 1.0, 1.0, s, s, s, s)

    def evaluate(population):
        return [r.composite_score for r in trials.run_trials(population, fake, schedule="dynamic")]

    best, score = ev.evolutionary_search(ev.RANGES, evaluate, seed=2, generations=2)
    q.put((rank, len(ran), best.__dict__["hidden_dim"], best.lr, score))
    dist.destroy_process_group()


def test_evolutionary_generations_shard_over_ranks_and_agree():
    """A generation's evaluations go through the shared trial queue (one per rank at a time); both
    ranks breed the same next generation from the gathered scores and end with the same best."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_evo_worker, args=(r, 2, 29625, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert got[0][1] + got[1][1] == 10 + 15 + 15          # every configuration evaluated once, somewhere
    assert got[0][1] > 0 and got[1][1] > 0
    assert got[0][2:] == got[1][2:]


def test_operand_checks_refuse_short_tensors_before_any_launch():
    """ops._expect: the kernels index by the sizes the descriptors state, so a tensor shorter than
    stated must be an exception on the host, never a launch (a d_h table cut to the first piece's
    rows reached gatmi_project_bwd in a bench tool and faulted the card)."""
    from gat_amd import ops
    t = torch.zeros(5, 8)
    with pytest.raises(ValueError, match="expected shape"):
        ops._expect(t, (6, 8), torch.float32, "d_hp")
    with pytest.raises(TypeError, match="expected torch.float32"):
        ops._expect(t.double(), (5, 8), torch.float32, "d_hp")
    with pytest.raises(ValueError, match="contiguous"):
        ops._expect(torch.zeros(8, 5).t(), (5, 8), torch.float32, "d_hp")
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops._expect(t, (5, 8), torch.float32, "d_hp")
    ops._expect(None, (1,), torch.float32, "optional")          # absent optional operands pass
    with pytest.raises(ValueError, match="d_hp: expected shape"):
        ops.project_bwd(torch.zeros(10, 4).cuda() if torch.cuda.is_available() else _FakeCuda(10, 4),
                        _FakeCuda(6, 4), _FakeCuda(5, 8), 2, 3, True)


class _FakeCuda(torch.Tensor):
    """A CPU tensor that says it is on the GPU: reaches the shape checks without a device."""

    @staticmethod
    def __new__(cls, *shape):
        return torch.Tensor._make_subclass(cls, torch.zeros(*shape))

    @property
    def is_cuda(self):
        return True
