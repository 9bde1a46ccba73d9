# Where do the GEMM waves spend their time?  One SQ pass, one TCP pass.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM -d gpurun_out/pmc_gemm_sq -o pmc --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_gemm_sq.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum -d gpurun_out/pmc_gemm_tcp -o pmc --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_gemm_tcp.log 2>&1 || exit 1
python - <<'PY'
import csv, glob, re
from collections import defaultdict
for d in ("gpurun_out/pmc_gemm_sq", "gpurun_out/pmc_gemm_tcp"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"gatmi::(\w+(?:<[^>]*>)?)", row["Kernel_Name"])
            if m and ("gemm" in m.group(1) or "inagg_bwd" in m.group(1) or "pack" in m.group(1)):
                acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc):
        print(k, {n: f"{sum(v)/len(v):.3g}" for n, v in sorted(acc[k].items())})
PY
