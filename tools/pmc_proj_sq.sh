# Where do the waves of the hidden-layer projection GEMMs spend their time?  (tools/proj_run.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM -d gpurun_out/pmc_proj_sq -o pmc --output-format csv -- python3 tools/proj_run.py > gpurun_out/pmc_proj_sq.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d gpurun_out/pmc_proj_sq2 -o pmc --output-format csv -- python3 tools/proj_run.py > gpurun_out/pmc_proj_sq2.log 2>&1 || exit 1
python - <<'PY'
import csv, glob, re
from collections import defaultdict
for d in ("gpurun_out/pmc_proj_sq", "gpurun_out/pmc_proj_sq2"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "gemm_f32" in row["Kernel_Name"]:
                key = re.sub(r"\(.*", "", row["Kernel_Name"].replace("void gatmi::", "")) + " grid=" + row["Grid_Size"]
                acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc):
        print(k, {n: f"{sum(v)/len(v):.4g}" for n, v in sorted(acc[k].items())})
PY
