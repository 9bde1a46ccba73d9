"""CPU restatement of the reference's feature engineering.  TEST INFRASTRUCTURE ONLY (only tests/
may import it).  Follows /root/reference/gat/preprocesser.py:15-48 step for step with the same
pandas calls (groupby -> set_index -> rolling("10s").apply -> concat), minus the process pool
(chunks are handled one after the other: same chunks, same results) and minus the final shuffle.
The encoders are pinned differently: /root/reference/gat/encoder.py imports in this container, so
tests/golden/etl/etl_encoders.npz holds the outputs of the reference's OWN functions
(tests/golden/etl/make_etl_golden.py)."""
import pandas as pd


def diversity_index(series):                                  # preprocesser.py:15-16
    return series.nunique() / len(series) if len(series) > 0 else 0


def port_scan_label(df: pd.DataFrame, num_chunks: int) -> pd.DataFrame:
    df = df.copy()
    df["timestamp"] = pd.to_datetime(df["timestamp"], utc=True)       # :38
    df["destination_port_label"] = pd.to_numeric(df["destination_port_label"], errors="coerce")
    df.sort_values(by=["ip_source", "timestamp"], inplace=True)       # :39
    size = len(df) // num_chunks                                       # :31-34
    out = []
    for k in range(num_chunks):
        chunk = df.iloc[k * size:(k + 1) * size]
        for _, group in chunk.groupby("ip_source"):                    # :25-29
            group = group.set_index("timestamp")                       # :18-22
            res = group["destination_port_label"].rolling(window="10s").apply(diversity_index)
            group["diversity_index"] = res.values
            group.reset_index(inplace=True)
            out.append(group)
    df = pd.concat(out, ignore_index=True)                             # :45
    df["diversity_index"] = df["diversity_index"].fillna(0)            # :46
    return df
