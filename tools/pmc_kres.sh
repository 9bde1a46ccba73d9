# Where do the K-resident GEMM waves wait?  Two SQ passes over a short bench run.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL -d gpurun_out/pmc_kres_a -o pmc --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_kres_a.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d gpurun_out/pmc_kres_b -o pmc --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_kres_b.log 2>&1 || exit 1
python - <<'PY'
import csv, glob, re
from collections import defaultdict
for d in ("gpurun_out/pmc_kres_a", "gpurun_out/pmc_kres_b"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"gatmi::(\w+(?:<[^>]*>)?)", row["Kernel_Name"])
            if m and "gemm" in m.group(1):
                acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc):
        print(k, {n: f"{sum(v)/len(v):.3g}" for n, v in sorted(acc[k].items())})
PY
