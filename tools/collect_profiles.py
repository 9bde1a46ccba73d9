"""Copy the summaries of one tools/profile_round.sh run from gpurun_out/prof_<tag>/ into
profiles/<round>/ (tracked): bench JSON lines, the rocprofv3 --kernel-trace --stats kernel
table, a per-kernel PMC summary, and pmc_traffic.json (HBM bytes per launch that bench.py
reports as roofline.traffic).
usage: python tools/collect_profiles.py <tag> <round-dir> [<prefix>]"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"gatmi::(\w+)", name)
    if not m:
        return None
    # the attention kernels are templated on the table's element type: keep the two apart
    return m.group(1) + ("_bf16" if "bf16x4" in name else "")


def pmc(dirname):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if k:
                acc[k].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    tag, dst = sys.argv[1], sys.argv[2]
    prefix = sys.argv[3] if len(sys.argv) > 3 else tag
    src = os.path.join("gpurun_out", f"prof_{tag}")
    os.makedirs(dst, exist_ok=True)
    for f in glob.glob(os.path.join(src, "bench_*.json")):
        shutil.copy(f, os.path.join(dst, f"{prefix}_{os.path.basename(f)}"))
    stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(dst, f"{prefix}_kernel_stats.csv"))
    fetch, write = pmc(os.path.join(src, "pmc_fetch")), pmc(os.path.join(src, "pmc_write"))
    lines = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 2 "
             "--warmup 1, products-shaped; KiB per dispatch",
             "# HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE tallies "
             "128-B requests at 64 B, MI355X_MICROARCH.md HBM section)"]
    traffic = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        lines.append(f"{k:28s} FETCH_SIZE: avg {f:.1f} (n={nf})  WRITE_SIZE: avg {w:.1f} (n={nw})")
        traffic[k] = int((2 * f + w) * 1024)
    open(os.path.join(dst, f"pmc_{prefix}.txt"), "w").write("\n".join(lines) + "\n")
    tags = {"inagg_fwd": ["inagg_fwd_stream_kernel", "inagg_fwd3_kernel", "inagg_fwd_kernel"], "inagg_bwd_edge": ["inagg_bwd3_kernel", "inagg_bwd2_kernel", "inagg_bwd_kernel"],
            "attn_fwd": ["attn_fwd_kernel"], "attn_bwd_edge": ["attn_bwd_edge2_kernel", "attn_bwd_edge_kernel"],
            "inagg_fwd_bf16": ["inagg_fwd16_kernel"], "inagg_bwd_edge_bf16": ["inagg_bwd16_kernel"],
            "attn_fwd_bf16": ["attn_fwd_kernel_bf16"],
            "attn_bwd_edge_bf16": ["attn_bwd_edge2_kernel_bf16", "attn_bwd_edge_kernel_bf16"]}
    old = {}
    try:    # kernels this run did not launch keep their earlier measurement
        old = json.load(open(os.path.join(dst, "pmc_traffic.json"))).get("kernels", {})
    except (OSError, ValueError):
        pass
    out = {"workload": "products",
           "note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc "
                   f"passes (profiles: pmc_{prefix}.txt); FETCH_SIZE doubled per MI355X_MICROARCH.md",
           "kernels": dict(old)}
    for tag_, names in tags.items():
        for n in names:
            if n in traffic:
                out["kernels"][tag_] = traffic[n]
                break
    json.dump(out, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
    # the run.py-shaped layer at 10^6 nodes (fused tiles): kernel table + the two counter passes
    rstats = glob.glob(os.path.join(src, "stats_runpy", "**", "*kernel_stats.csv"), recursive=True)
    if rstats:
        shutil.copy(rstats[0], os.path.join(dst, f"{prefix}_runpy1m_kernel_stats.csv"))
        rf, rw = pmc(os.path.join(src, "pmc_fetch_runpy")), pmc(os.path.join(src, "pmc_write_runpy"))
        rl = ["# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --workload runpy1m --steps 2 "
              "--warmup 1; KiB per dispatch; HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024"]
        for k in sorted(set(rf) | set(rw)):
            f, nf = rf.get(k, (0.0, 0))
            w, nw = rw.get(k, (0.0, 0))
            rl.append(f"{k:28s} FETCH_SIZE: avg {f:.1f} (n={nf})  WRITE_SIZE: avg {w:.1f} (n={nw})  "
                      f"HBM bytes {int((2 * f + w) * 1024)}")
        open(os.path.join(dst, f"pmc_{prefix}_runpy1m.txt"), "w").write("\n".join(rl) + "\n")
        print("\n".join(rl))
    print(open(os.path.join(dst, f"pmc_{prefix}.txt")).read())
    print(json.dumps(out["kernels"]))


if __name__ == "__main__":
    main()
