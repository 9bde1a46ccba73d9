"""The reference's feature engineering under its own import path
(/root/reference/gat/preprocesser.py:50-90): preprocess_df() loads data/traces.csv (or the file
GAT_TRACES_CSV names) and runs gat_amd/etl.py -- the reference's pipeline, vectorised.  The
reference does not ship that CSV; when it is absent the synthetic stand-in table of
gat_amd.trials.load_xy() is used instead and says so."""
import os

from gat_amd import etl
from gat_amd.trials import load_xy, shuffle_settings

_PREPARED = "__prepared__"      # marks the stand-in table, whose columns are final already


def preprocess_df():
    path = etl.traces_path()
    if os.path.exists(path):
        seed, chunks = shuffle_settings()
        return etl.preprocess_df(etl.read_traces(path), num_chunks=chunks, shuffle_seed=seed)
    X, y = load_xy()     # GAT_TRACES_NPZ or the synthetic stand-in: columns are final already
    df = X.copy()
    df["is_anomaly"] = y.values
    df.attrs[_PREPARED] = True
    return df


def preprocess_X(df):
    if df.attrs.get(_PREPARED):
        return df.drop(columns=["is_anomaly"])
    return etl.preprocess_X(df)


def preprocess_y(df):
    return df["is_anomaly"]
