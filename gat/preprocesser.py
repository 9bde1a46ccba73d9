"""Stand-ins for the reference's feature engineering (/root/reference/gat/preprocesser.py:
50-90), which is host-side pandas ETL over a CSV the reference does not ship -- out of the
hot path.  preprocess_df() returns the prepared table,
preprocess_X / preprocess_y split it, with the reference's names and return types."""
from gat_amd.trials import load_xy


def preprocess_df():
    X, y = load_xy()
    df = X.copy()
    df["is_anomaly"] = y.values
    return df


def preprocess_X(df):
    return df.drop(columns=["is_anomaly"])


def preprocess_y(df):
    return df["is_anomaly"]
