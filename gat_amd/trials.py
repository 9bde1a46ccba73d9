"""Hyper-parameter trials, one per GPU (BASELINE.json config 4, SURVEY.md 8e way 1).

The reference runs `run_experiment(config)` for one configuration after another in one
process (/root/reference/grid_search.py:70-118, random_search.py:113-146 and three more
verbatim clones).  Trials share nothing, so under `torch.distributed.run` rank r simply
takes configurations r, r + P, r + 2P ... on its own GPU (or, schedule="dynamic", whatever a
shared counter hands it next): no collective touches the data path; the per-trial results are
gathered once at the end over gloo.  Without a launcher
(WORLD_SIZE unset) everything runs in this process, in the reference's order.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Any, Callable, List, Sequence

import numpy as np

from .data import Metrics, convert_to_graph, synthetic_traces

TEST_SIZE = 0.25       # /root/reference/grid_search.py:14
RANDOM_STATE = 42      # /root/reference/grid_search.py:15


@dataclass
class Config:          # /root/reference/grid_search.py:20-28
    optimizer: Any
    lr: float
    weight_decay: float
    epochs: int
    patience: int
    hidden_dim: int
    dropout: float


@dataclass
class ConfigResults:   # /root/reference/grid_search.py:30-40
    config: Config
    metrics: List[Metrics]
    avg_val_accuracy: float
    avg_val_loss: float
    avg_train_loss: float
    avg_val_precision: float
    avg_val_recall: float
    avg_val_f1: float
    composite_score: float


_XY = None
_GRAPHS = None


def shuffle_settings():
    """(shuffle_seed, num_chunks) for the feature engineering.  One process: the reference's own
    behaviour -- an unseeded shuffle, one chunk per host CPU (preprocesser.py:40-47) -- unless
    GAT_SHUFFLE_SEED / GAT_ETL_CHUNKS say otherwise.  Several ranks: every rank prepares the table
    itself, so both are FIXED (seed 0, 8 chunks by default) -- with an unseeded shuffle each rank
    would train and score its trials on a differently ordered table and a different split, and the
    scores gathered from the ranks would not be comparable (the reference evaluates every
    configuration of a search on one split)."""
    seed, chunks = os.environ.get("GAT_SHUFFLE_SEED"), os.environ.get("GAT_ETL_CHUNKS")
    if world()[1] > 1:
        return int(seed) if seed is not None else 0, int(chunks) if chunks is not None else 8
    return (int(seed) if seed is not None else None), (int(chunks) if chunks is not None else None)


def load_xy():
    """(X, y) once per process, as grid_search.py:42-49 caches them.  In this order: the reference's
    own input, data/traces.csv (or the file GAT_TRACES_CSV names), through the reference's feature
    engineering (gat_amd/etl.py); prepared arrays GAT_TRACES_NPZ points at; else -- the reference
    does not ship its CSV -- a synthetic table of the same shape, announced as such."""
    global _XY
    if _XY is None:
        from . import etl
        csv = etl.traces_path()
        path = os.environ.get("GAT_TRACES_NPZ")
        if os.path.exists(csv):
            seed, chunks = shuffle_settings()
            df = etl.preprocess_df(etl.read_traces(csv, announce=world()[0] == 0),
                                   num_chunks=chunks, shuffle_seed=seed)
            _XY = (etl.preprocess_X(df), etl.preprocess_y(df))
        elif path:
            import pandas as pd
            z = np.load(path)
            _XY = (pd.DataFrame(z["X"]), pd.Series(z["y"]))
        else:
            rows = int(os.environ.get("GAT_SYNTH_ROWS", "20000"))
            if world()[0] == 0:     # a run on the stand-in must not be mistaken for one on real traces
                print(f"[gat_amd] data/traces.csv is not part of the reference repository and GAT_TRACES_NPZ "
                      f"is not set: training on the SYNTHETIC stand-in table ({rows} rows x 25 features)")
            _XY = synthetic_traces(rows)
    return _XY


def split_data():
    """Stratified 75/25 split and one kNN graph per side, as grid_search.py:42-54.  The
    reference rebuilds both graphs in every run although the split is deterministic; they
    are built once per process here."""
    global _GRAPHS
    from sklearn.model_selection import train_test_split
    X, y = load_xy()
    if _GRAPHS is None:
        X_tr, X_te, y_tr, y_te = train_test_split(X, y, test_size=TEST_SIZE, stratify=y,
                                                  random_state=RANDOM_STATE)
        _bind_device()
        _GRAPHS = (convert_to_graph(X_tr, y_tr), convert_to_graph(X_te, y_te), y_tr, y_te)
    return _GRAPHS


def initialize_model(train_data, y_train, config):
    from .model import GAT
    model = GAT(optimizer=config.optimizer, num_features=train_data.num_features,
                num_classes=len(np.unique(y_train)), weight_decay=config.weight_decay,
                dropout=config.dropout, hidden_dim=config.hidden_dim, epochs=config.epochs,
                lr=config.lr, patience=config.patience,
                verbose=os.environ.get("GAT_QUIET") is None)
    model.checkpoint_path = f"best_model_rank{os.environ.get('RANK', '0')}_{os.getpid()}.pth"
    return model


def run_experiment(config, num_runs: int = 1) -> ConfigResults:
    """One configuration: train `num_runs` times, average the LAST epoch's validation
    numbers, composite = mean of accuracy / precision / recall / F1 (grid_search.py:70-109)."""
    runs = []
    for _ in range(num_runs):
        train_data, test_data, y_train, _ = split_data()
        model = initialize_model(train_data, y_train, config)
        runs.append(model.train_model(train_data, test_data))
        if os.path.exists(model.checkpoint_path):
            os.remove(model.checkpoint_path)
    last = lambda name: float(np.mean([getattr(m, name)[-1] for m in runs]))
    acc, prec, rec, f1 = last("val_accuracy"), last("val_precision"), last("val_recall"), last("val_f1")
    return ConfigResults(config, runs, acc, last("val_loss"), last("train_loss"), prec, rec, f1,
                         0.25 * acc + 0.25 * prec + 0.25 * rec + 0.25 * f1)


def world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def _bind_device() -> None:
    """Make cuda:LOCAL_RANK this process's current device, so that everything that takes the
    current device (the kNN graph build of split_data, workspaces) lands on the rank's own GPU
    instead of every rank opening a context on GPU 0."""
    import torch
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))


_CALLS = 0


def run_trials(configs: Sequence[Config], experiment: Callable[[Config], ConfigResults] = run_experiment,
               schedule: str = "static") -> List[ConfigResults]:
    """All configurations over the ranks; every rank returns the full list in configuration
    order.  schedule "static": rank r takes r, r + P, ... (no communication until the end);
    "dynamic": a shared counter in the process group's store hands out the next configuration
    to whichever GPU is free -- trials differ in epochs, width and early stopping, so their
    run times do.  Either way no collective touches the data path."""
    global _CALLS
    if schedule not in ("static", "dynamic"):
        raise ValueError(f"unknown schedule {schedule!r}")
    rank, size = world()
    _bind_device()
    if size == 1:
        return [experiment(c) for c in configs]
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("gloo")       # results only: a few floats per trial
    _CALLS += 1
    if schedule == "static":
        mine = [(i, experiment(c)) for i, c in enumerate(configs) if i % size == rank]
    else:
        store = dist.distributed_c10d._get_default_store()
        key = f"gat_amd/trials/{_CALLS}/next"
        mine = []
        while True:
            i = int(store.add(key, 1)) - 1    # atomic fetch-and-add on rank 0's store
            if i >= len(configs):
                break
            mine.append((i, experiment(configs[i])))
    slim = [(i, _portable(r)) for i, r in mine]
    gathered = [None] * size
    dist.all_gather_object(gathered, slim)
    merged = sorted((item for part in gathered for item in part), key=lambda t: t[0])
    return [_restore(r, configs[i]) for i, r in merged]


def _portable(r: ConfigResults):
    d = dict(r.__dict__)
    d["config"] = None                          # optimizer classes pickle, but keep it small
    return d


def _restore(d, config) -> ConfigResults:
    d = dict(d)
    d["config"] = config
    return ConfigResults(**d)


def best_of(results: Sequence[ConfigResults]):
    best = max(results, key=lambda r: r.composite_score, default=None)
    return (best.config, best.composite_score) if best and best.composite_score > 0 else (None, 0)


def write_results(results: Sequence[ConfigResults], path: str) -> None:
    """The text file the reference's searches leave behind (grid_search.py:166-175)."""
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w") as f:
        for r in results:
            f.write(f"Config: {r.config}, Avg Val Accuracy: {r.avg_val_accuracy}, "
                    f"Avg Val Loss: {r.avg_val_loss}, Avg Train Loss: {r.avg_train_loss}, "
                    f"Avg Val Precision: {r.avg_val_precision}, Avg Val Recall: {r.avg_val_recall}, "
                    f"Avg Val F1: {r.avg_val_f1}, Composite Score: {r.composite_score}\n")
