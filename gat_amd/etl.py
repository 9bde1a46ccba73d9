"""Feature engineering of the reference: raw Kubernetes network traces ->
the 25 numeric columns the model trains on.  Same functions, same column names and the same
results as /root/reference/gat/preprocesser.py:15-90 and /root/reference/gat/encoder.py:7-58,
written as array operations instead of per-row `apply` calls and an `mp.Pool` of per-group
`rolling().apply()` (the reference's slowest step: a Python call per row).  Host-side: this is
one-off ETL in front of the hot path, not part of it.

Columns of the raw table (the reference's data/traces.csv, which it does not ship): timestamp,
ip_source, ip_destination, source_pod_label, destination_pod_label, source_namespace_label,
destination_namespace_label, source_port_label, destination_port_label, ack_flag, psh_flag,
is_anomaly.
"""
from __future__ import annotations

import hashlib
import multiprocessing as mp
import os
from typing import Optional

import numpy as np
import pandas as pd

FEATURES = ["ack_flag", "psh_flag", "diversity_index"] \
    + [f"ip_source_part{i}" for i in range(1, 9)] + [f"ip_destination_part{i}" for i in range(1, 9)] \
    + ["source_pod_label_normalized", "destination_pod_label_normalized",
       "source_namespace_label_normalized", "destination_namespace_label_normalized",
       "source_port_label_normalized", "destination_port_label_normalized"]   # preprocesser.py:76-85


# ------------------------------------------------------------------ encoders (encoder.py)
def ip_parts(values) -> np.ndarray:
    """encoder.py:7-20 -> float64 [n, 8]: an IPv6 address is split at ':' (empty groups count as 0,
    hex, / 65535), an IPv4 address at '.' (/ 255), padded with zeros to eight parts."""
    out = np.zeros((len(values), 8), dtype=np.float64)
    cache = {}
    for i, ip in enumerate(values):
        row = cache.get(ip)
        if row is None:
            if ":" in ip:
                parts, top = [int(x, 16) if x else 0 for x in ip.split(":")], 65535
            elif "." in ip:
                parts, top = [int(x) for x in ip.split(".")], 255
            else:
                raise ValueError("Invalid IP address format")
            if len(parts) > 8:      # the reference builds a ragged row here and pandas raises
                raise ValueError(f"more than eight groups in {ip!r}")
            row = cache[ip] = np.array(parts + [0] * (8 - len(parts)), dtype=np.float64) / top
        out[i] = row
    return out


def ip_encoder(df: pd.DataFrame, column_name: str) -> pd.DataFrame:
    parts = pd.DataFrame(ip_parts(df[column_name].tolist()), index=df.index,
                         columns=[f"{column_name}_part{i}" for i in range(1, 9)])
    return df.drop(column_name, axis=1).join(parts)


def _minmax(v: np.ndarray) -> np.ndarray:
    """(v - min) / (max - min) in float64, as the torch.float64 arithmetic of encoder.py:34-37"""
    lo, hi = v.min(), v.max()
    with np.errstate(invalid="ignore", divide="ignore"):
        return (v - lo) / (hi - lo)


def string_encoder(df: pd.DataFrame, column_name: str) -> pd.DataFrame:
    """encoder.py:27-41: SHA-256 of the string as a 256-bit integer (missing -> 0), converted to
    float64, min-max normalised over the column."""
    col = df[column_name]
    uniq = {s: float(int(hashlib.sha256(s.encode()).hexdigest(), 16)) for s in col.dropna().unique()}
    hashed = np.array([uniq[s] if isinstance(s, str) else 0.0 for s in col.tolist()], dtype=np.float64)
    df = df.drop(column_name, axis=1)
    df[f"{column_name}_normalized"] = _minmax(hashed) if len(hashed) else hashed
    return df


def number_normalizer(df: pd.DataFrame, column_name: str) -> pd.DataFrame:
    """encoder.py:43-54: int() of every present value, min-max over the present ones; missing
    values stay missing (preprocess_X turns them into 0 afterwards)."""
    col = df[column_name]
    present = col.notna().values
    ints = np.array([int(v) for v in col[present].tolist()], dtype=np.float64)
    out = np.full(len(col), np.nan)
    if len(ints):
        out[present] = _minmax(ints)
    df = df.drop(column_name, axis=1)
    df[f"{column_name}_normalized"] = out
    return df


def boolean_string_to_int(df: pd.DataFrame, column: str) -> pd.DataFrame:
    df[column] = df[column].astype(int)          # encoder.py:56-58
    return df


# ------------------------------------------------------------------ rolling port diversity
def rolling_diversity(times_ns: np.ndarray, ports: np.ndarray, window_ns: int = 10_000_000_000) -> np.ndarray:
    """For every row i of ONE source address (rows in time order): distinct destination ports /
    rows among the rows j <= i with t_j > t_i - window -- `rolling("10s").apply(nunique / len)` of
    preprocesser.py:15-23.  Two pointers and a multiset: O(rows) instead of a Python call per row."""
    n = len(times_ns)
    out = np.zeros(n, dtype=np.float64)
    counts = {}
    lo = 0
    valid = 0                                   # rows of the window whose port is present
    for i in range(n):
        p = ports[i]
        if p == p:                              # a missing port (NaN) counts as a row, not as a value
            counts[p] = counts.get(p, 0) + 1
            valid += 1
        while times_ns[lo] <= times_ns[i] - window_ns:
            q = ports[lo]
            if q == q:
                c = counts[q] - 1
                valid -= 1
                if c:
                    counts[q] = c
                else:
                    del counts[q]
            lo += 1
        # pandas evaluates a window only when it holds at least one present value (min_periods = 1)
        out[i] = len(counts) / (i - lo + 1) if valid else np.nan
    return out


def construct_port_scan_label(df: pd.DataFrame, num_chunks: Optional[int] = None,
                              shuffle_seed: Optional[int] = None) -> pd.DataFrame:
    """preprocesser.py:37-48.  Rows are sorted by (ip_source, timestamp) and cut BY POSITION into
    `num_chunks` equal pieces (default: the host's CPU count, as the reference's pool is sized) --
    a source address that straddles a cut starts a fresh window in the next piece, and the
    len(df) % num_chunks rows left over at the end are dropped, exactly as split_dataframe does.
    Ends with the reference's shuffle (`shuffle_seed` None: unseeded, like `df.sample(frac=1)`)."""
    df = df.copy()
    df["timestamp"] = pd.to_datetime(df["timestamp"], utc=True)
    df.sort_values(by=["ip_source", "timestamp"], inplace=True)
    num_chunks = num_chunks or mp.cpu_count()
    chunk = len(df) // num_chunks
    pieces = []
    for k in range(num_chunks):
        part = df.iloc[k * chunk:(k + 1) * chunk]
        # groupby sorts its keys; inside a key the rows keep their (time) order
        order = np.argsort(part["ip_source"].values, kind="stable")
        part = part.iloc[order].copy()
        t = part["timestamp"].values.astype("datetime64[ns]").astype(np.int64)
        ports = pd.to_numeric(part["destination_port_label"], errors="coerce").values.astype(np.float64)
        src = part["ip_source"].values
        div = np.zeros(len(part))
        start = 0
        for end in list(np.flatnonzero(src[1:] != src[:-1]) + 1) + [len(part)]:
            if end > start:
                div[start:end] = rolling_diversity(t[start:end], ports[start:end])
            start = end
        part["diversity_index"] = div
        pieces.append(part)
    out = pd.concat(pieces, ignore_index=True) if pieces else df.assign(diversity_index=[])
    out["diversity_index"] = out["diversity_index"].fillna(0)
    return out.sample(frac=1, random_state=shuffle_seed).reset_index(drop=True)


# ------------------------------------------------------------------ the raw table
def traces_path() -> str:
    """Where the raw traces are looked for: GAT_TRACES_CSV, else the reference's fixed location
    (load_data.py:6)."""
    return os.environ.get("GAT_TRACES_CSV", os.path.join("data", "traces.csv"))


def read_traces(source=None, announce: bool = True) -> pd.DataFrame:
    """The raw table as a frame (the reference's I/O step, load_data.py:8): `source` is a path, or
    None for traces_path()."""
    source = traces_path() if source is None else os.fspath(source)
    if announce:
        print(f"[gat_amd] reading raw traces from {source}")
    frame = pd.read_csv(source)
    if announce:
        print(f"[gat_amd] {len(frame)} trace rows read")
    return frame


# ------------------------------------------------------------------ the three entry points
def preprocess_df(raw: pd.DataFrame, num_chunks: Optional[int] = None,
                  shuffle_seed: Optional[int] = None) -> pd.DataFrame:
    df = construct_port_scan_label(raw, num_chunks, shuffle_seed)
    df["is_anomaly"] = df["is_anomaly"].replace({"True": 1, "False": 0}).astype(int)   # preprocesser.py:53
    return df


def preprocess_X(df: pd.DataFrame) -> pd.DataFrame:
    """preprocesser.py:56-87 without its reporting side effects (data_insights.txt, PNG plots)."""
    X = df.drop(columns=["is_anomaly"])
    for column in ("ip_source", "ip_destination"):
        X = ip_encoder(X, column)
    for column in ("source_pod_label", "destination_pod_label", "source_namespace_label",
                   "destination_namespace_label"):
        X = string_encoder(X, column)
    for column in ("source_port_label", "destination_port_label"):
        X = number_normalizer(X, column)
    for column in ("ack_flag", "psh_flag"):
        X = boolean_string_to_int(X, column)
    X = X.apply(pd.to_numeric, errors="coerce").fillna(0)
    return X[FEATURES]


def preprocess_y(df: pd.DataFrame) -> pd.Series:
    return df["is_anomaly"]


def synthetic_raw_traces(num_rows: int = 5000, seed: int = 0) -> pd.DataFrame:
    """A raw table of the reference's schema (its CSV is not in its repository): a few hundred
    pods talking over IPv4 and IPv6, some sources scanning many ports in a burst (labelled
    anomalous), missing pod labels and ports here and there."""
    rng = np.random.default_rng(seed)
    n_src = max(num_rows // 40, 4)
    v4 = [f"10.{rng.integers(0, 4)}.{rng.integers(0, 256)}.{rng.integers(1, 255)}" for _ in range(n_src)]
    v6 = [f"fd00::{rng.integers(1, 0xffff):x}:{rng.integers(1, 0xffff):x}" for _ in range(max(n_src // 5, 1))]
    addrs = np.array(v4 + v6)
    src = rng.integers(0, len(addrs), num_rows)
    dst = rng.integers(0, len(addrs), num_rows)
    scanner = rng.random(len(addrs)) < 0.08
    anomaly = scanner[src] & (rng.random(num_rows) < 0.9)
    t0 = pd.Timestamp("2024-03-01T00:00:00Z").value
    times = t0 + np.sort(rng.integers(0, 600 * 10**9, num_rows))          # ten minutes of traffic
    dport = np.where(anomaly, rng.integers(1, 65535, num_rows), rng.choice([53, 80, 443, 8080, 5432], num_rows))
    pods = np.array([f"pod-{i}" for i in range(60)], dtype=object)
    spod, dpod = pods[src % 60].copy(), pods[dst % 60].copy()
    spod[rng.random(num_rows) < 0.03] = None
    dport_s = dport.astype(object)
    dport_s[rng.random(num_rows) < 0.01] = None
    return pd.DataFrame({
        "timestamp": pd.to_datetime(times, utc=True).strftime("%Y-%m-%dT%H:%M:%S.%fZ"),
        "ip_source": addrs[src], "ip_destination": addrs[dst],
        "source_pod_label": spod, "destination_pod_label": dpod,
        "source_namespace_label": np.array(["default", "kube-system", "prod", "dev"], dtype=object)[src % 4],
        "destination_namespace_label": np.array(["default", "kube-system", "prod", "dev"], dtype=object)[dst % 4],
        "source_port_label": rng.integers(1024, 65535, num_rows),
        "destination_port_label": dport_s,
        "ack_flag": rng.random(num_rows) < 0.7, "psh_flag": rng.random(num_rows) < 0.3,
        "is_anomaly": anomaly,
    })
