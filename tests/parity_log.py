"""What the fp32 comparisons of the GPU parity tests actually achieved.  `close()` in
test_gpu_parity.py records, per compared tensor, the max-norm error (max |got - ref| over the
tensor's largest magnitude -- the quantity the 1e-5 bar is stated on) and the element-wise
relative error over the elements that carry signal (|ref| >= 1e-3 of the largest); conftest.py
prints the worst of each per kind of tensor when the session ends and writes the table to
gpurun_out/parity_report.json."""
import json
import os
import re

import torch

RECORDS = {}


def kind_of(name: str) -> str:
    """'trial 3: n=17 h=5 ... d_weight' -> 'd_weight'; the last tensor name in the label"""
    m = re.findall(r"d_[a-z_]+|log-probabilities|constant rows|\bout\b|\ba_dst\b|\bh$", name)
    return m[-1] if m else "other"


def record(name, got, ref, scale):
    if got.numel() == 0:
        return
    diff = (got.double() - ref.double()).abs()
    max_norm = float(diff.max()) / scale
    big = ref.abs() >= 1e-3 * scale
    elem = float((diff[big] / ref.double().abs()[big]).max()) if bool(big.any()) else 0.0
    strict = float((diff <= 1e-5 * ref.double().abs()).double().mean())
    r = RECORDS.setdefault(kind_of(name), dict(n=0, max_norm=0.0, elementwise=0.0, strict_frac=1.0))
    r["n"] += 1
    r["max_norm"] = max(r["max_norm"], max_norm)
    r["elementwise"] = max(r["elementwise"], elem)
    r["strict_frac"] = min(r["strict_frac"], strict)


def summary_lines():
    lines = ["fp32 parity achieved (worst over the session's comparisons; bar: max-norm 1e-5):",
             f"  {'tensor':<20}{'compared':>9}{'max-norm':>12}{'elementwise*':>14}{'strict 1e-5 frac':>18}"]
    for k, r in sorted(RECORDS.items()):
        lines.append(f"  {k:<20}{r['n']:>9}{r['max_norm']:>12.2e}{r['elementwise']:>14.2e}{r['strict_frac']:>18.4f}")
    lines.append("  * largest |got-ref|/|ref| over elements with |ref| >= 1e-3 max|ref|; strict = share of "
                 "elements within 1e-5 of their OWN size, no absolute term")
    return lines


def dump(root):
    if not RECORDS:
        return
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "parity_report.json"), "w") as f:
            json.dump(RECORDS, f, indent=1, sort_keys=True)
    except OSError:
        pass
