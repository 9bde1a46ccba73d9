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
    r = RECORDS.setdefault(kind_of(name), dict(n=0, max_norm=0.0, elementwise=0.0, strict_frac=1.0,
                                               strict_sum=0.0, strict_n=0, zero_ref=0))
    r["n"] += 1
    r["max_norm"] = max(r["max_norm"], max_norm)
    r["elementwise"] = max(r["elementwise"], elem)
    # "strict": share of elements within a relative 1e-5 of their OWN size.  Only elements whose
    # reference is not exactly zero can be asked that (a graph of self-loops alone has d_att_* == 0
    # exactly: any round-off there is 'infinitely' wrong relatively and used to pin the column at 0);
    # comparisons against an all-zero reference are counted apart and judged by max_norm alone.
    nz = ref != 0
    if bool(nz.any()):
        strict = float((diff[nz] <= 1e-5 * ref.double().abs()[nz]).double().mean())
        r["strict_frac"] = min(r["strict_frac"], strict)
        r["strict_sum"] += strict
        r["strict_n"] += 1
    else:
        r["zero_ref"] += 1


VS64 = {}


def record_vs_fp64(kind, err_hip, err_o32, elem_hip, elem_o32):
    """distance of the HIP result and of the fp32 oracle from the fp64 oracle (max-norm and element-wise)"""
    r = VS64.setdefault(kind, dict(n=0, hip=0.0, oracle32=0.0, hip_elementwise=0.0, oracle32_elementwise=0.0,
                                   worst_ratio=0.0))
    r["n"] += 1
    r["hip"], r["oracle32"] = max(r["hip"], err_hip), max(r["oracle32"], err_o32)
    r["hip_elementwise"] = max(r["hip_elementwise"], elem_hip)
    r["oracle32_elementwise"] = max(r["oracle32_elementwise"], elem_o32)
    if err_o32 > 0:
        r["worst_ratio"] = max(r["worst_ratio"], err_hip / err_o32)


def summary_lines():
    lines = ["fp32 parity achieved (worst over the session's comparisons; bar: max-norm 1e-5):",
             f"  {'tensor':<20}{'compared':>9}{'max-norm':>12}{'elementwise*':>14}{'strict worst':>14}"
             f"{'strict mean':>13}{'zero refs':>11}"]
    for k, r in sorted(RECORDS.items()):
        mean = r["strict_sum"] / r["strict_n"] if r["strict_n"] else float("nan")
        lines.append(f"  {k:<20}{r['n']:>9}{r['max_norm']:>12.2e}{r['elementwise']:>14.2e}{r['strict_frac']:>14.4f}"
                     f"{mean:>13.4f}{r['zero_ref']:>11}")
    lines.append("  * largest |got-ref|/|ref| over elements with |ref| >= 1e-3 max|ref|; strict = share of the "
                 "elements with a NONZERO reference that are within 1e-5 of their OWN size (worst / mean over "
                 "the comparisons); zero refs = comparisons whose reference is exactly 0 everywhere")
    if VS64:
        lines.append("distance from the fp64 oracle (max-norm; element-wise over |ref| >= 1e-3 max): HIP vs the fp32 oracle")
        lines.append(f"  {'tensor':<12}{'cases':>6}{'HIP':>11}{'oracle32':>11}{'worst ratio':>13}{'HIP elem.':>12}{'o32 elem.':>12}")
        for k, r in sorted(VS64.items()):
            lines.append(f"  {k:<12}{r['n']:>6}{r['hip']:>11.2e}{r['oracle32']:>11.2e}{r['worst_ratio']:>13.2f}"
                         f"{r['hip_elementwise']:>12.2e}{r['oracle32_elementwise']:>12.2e}")
    return lines


def dump(root):
    if not RECORDS:
        return
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "parity_report.json"), "w") as f:
            json.dump(dict(RECORDS, vs_fp64=VS64) if VS64 else RECORDS, f, indent=1, sort_keys=True)
    except OSError:
        pass
