"""Feature engineering (gat_amd/etl.py) against the reference: the column
encoders against golden vectors produced by the reference's OWN /root/reference/gat/encoder.py
(tests/golden/etl/make_etl_golden.py), the rolling port-diversity index against the reference's
pandas algorithm restated in oracle/etl_oracle.py.  CPU only."""
import os

import numpy as np
import pandas as pd
import pytest

from gat_amd import etl
from oracle import etl_oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "etl", "etl_encoders.npz")


def test_encoders_equal_the_reference_encoders_bit_for_bit():
    z = np.load(GOLDEN)
    raw = etl.synthetic_raw_traces(3000, seed=7)
    for col in ("ip_source", "ip_destination"):
        got = etl.ip_encoder(raw.copy(), col)[[f"{col}_part{i}" for i in range(1, 9)]].values
        assert np.array_equal(got, z[col]), col
    for col in ("source_pod_label", "destination_pod_label", "source_namespace_label",
                "destination_namespace_label"):
        assert np.array_equal(etl.string_encoder(raw.copy(), col)[f"{col}_normalized"].values, z[col]), col
    for col in ("source_port_label", "destination_port_label"):
        got = etl.number_normalizer(raw.copy(), col)[f"{col}_normalized"].values
        assert np.array_equal(got, z[col], equal_nan=True), col
    for col in ("ack_flag", "psh_flag"):
        assert np.array_equal(etl.boolean_string_to_int(raw.copy(), col)[col].values, z[col])
    assert np.isnan(z["destination_port_label"]).any() and (z["source_pod_label"] >= 0).all()


def test_ip_encoder_forms_and_errors():
    got = etl.ip_parts(["10.0.0.1", "fe80::1:2", "::1", "1:2:3:4:5:6:7:8"])
    assert np.allclose(got[0], [10 / 255, 0, 0, 1 / 255, 0, 0, 0, 0])
    assert np.allclose(got[1], [0xfe80 / 65535, 0, 1 / 65535, 2 / 65535, 0, 0, 0, 0])
    assert np.allclose(got[2], [0, 0, 1 / 65535, 0, 0, 0, 0, 0])
    assert np.allclose(got[3], np.arange(1, 9) / 65535)
    with pytest.raises(ValueError, match="Invalid IP"):
        etl.ip_parts(["localhost"])


@pytest.mark.parametrize("rows,chunks,seed", [(2000, 1, 0), (2000, 8, 1), (1999, 7, 2), (300, 16, 3)])
def test_port_diversity_equals_the_pandas_rolling_algorithm(rows, chunks, seed):
    """Same chunks cut by position, same windows (t - 10 s, t], duplicate timestamps, missing ports,
    sources straddling a cut, len % chunks rows dropped -- row for row."""
    raw = etl.synthetic_raw_traces(rows, seed=seed)
    raw.loc[raw.index[::50], "timestamp"] = raw["timestamp"].iloc[0]        # duplicated timestamps
    want = etl_oracle.port_scan_label(raw, chunks)
    got = etl.construct_port_scan_label(raw, num_chunks=chunks, shuffle_seed=0)
    assert len(got) == len(want) == (rows // chunks) * chunks
    key = ["ip_source", "timestamp", "ip_destination", "source_port_label"]
    a = got.sort_values(key, kind="stable").reset_index(drop=True)
    b = want.sort_values(key, kind="stable").reset_index(drop=True)
    assert (a["timestamp"].values == b["timestamp"].values).all()
    np.testing.assert_array_equal(a["diversity_index"].values, b["diversity_index"].values)
    assert 0 < (a["diversity_index"] < 1).mean() < 1                       # both regimes occur


def test_pipeline_gives_the_25_training_columns():
    raw = etl.synthetic_raw_traces(1500, seed=4)
    df = etl.preprocess_df(raw, num_chunks=4, shuffle_seed=1)
    X, y = etl.preprocess_X(df), etl.preprocess_y(df)
    assert list(X.columns) == etl.FEATURES and len(etl.FEATURES) == 25
    assert X.shape == (1500, 25) and not X.isna().any().any()
    assert float(X.values.min()) >= 0.0 and float(X.values.max()) <= 1.0
    assert set(np.unique(y)) == {0, 1}
    # scanners (anomalous rows) see many distinct ports in their window
    assert X["diversity_index"][y == 1].mean() > X["diversity_index"][y == 0].mean()


def test_drop_in_preprocesser_reads_a_csv(tmp_path, monkeypatch):
    """gat.preprocesser (the reference's import path) runs the real pipeline when the traces file
    exists, and the synthetic stand-in otherwise."""
    import importlib
    raw = etl.synthetic_raw_traces(800, seed=9)
    (tmp_path / "data").mkdir()
    raw.to_csv(tmp_path / "data" / "traces.csv", index=False)
    monkeypatch.chdir(tmp_path)
    from gat_amd import trials
    monkeypatch.setattr(trials, "_XY", None)
    import gat.preprocesser as pp
    importlib.reload(pp)
    df = pp.preprocess_df()
    X, y = pp.preprocess_X(df), pp.preprocess_y(df)
    assert list(X.columns) == etl.FEATURES and len(X) == (800 // os.cpu_count()) * os.cpu_count()
    assert y.dtype.kind == "i"


_RANK_TABLE = r"""
import hashlib, os, sys
sys.path.insert(0, {root!r})
from gat_amd import trials
X, y = trials.load_xy()
h = hashlib.sha256(X.values.tobytes() + y.values.tobytes()).hexdigest()
print("TABLE", os.environ["RANK"], len(X), h)
"""


def test_every_rank_of_a_search_prepares_the_same_table(tmp_path):
    """with data/traces.csv present every rank runs the feature engineering itself; the
    reference's unseeded shuffle (preprocesser.py:47) would give each rank its own row order and so
    its own train/test split.  With WORLD_SIZE > 1 the shuffle seed and the chunk count are fixed:
    two ranks (separate processes) hold byte-identical X and y; one process alone keeps the
    reference's unseeded shuffle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    raw = etl.synthetic_raw_traces(640, seed=11)
    csv = tmp_path / "traces.csv"
    raw.to_csv(csv, index=False)
    lines = []
    for rank in (0, 1):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", GAT_TRACES_CSV=str(csv))
        out = subprocess.run([sys.executable, "-c", _RANK_TABLE.format(root=root)], env=env,
                             capture_output=True, text=True, check=True, cwd=tmp_path).stdout
        lines.append([ln for ln in out.splitlines() if ln.startswith("TABLE")][0].split())
    assert lines[0][2] == lines[1][2] == "640"
    assert lines[0][3] == lines[1][3], "ranks prepared different tables"
    from gat_amd import trials
    assert trials.shuffle_settings() == (None, None)       # one process: as the reference
