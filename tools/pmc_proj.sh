# MFMA utilisation and clock of the hidden-layer projection GEMMs (tools/proj_run.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc_proj -o pmc --output-format csv -- python3 tools/proj_run.py > gpurun_out/pmc_proj.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d gpurun_out/trace_proj -o tr --output-format csv -- python3 tools/proj_run.py > gpurun_out/trace_proj.log 2>&1 || exit 1
python - <<'PY'
import csv, glob, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/pmc_proj/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gemm_f32" in row["Kernel_Name"]:
            key = re.sub(r"\(.*", "", row["Kernel_Name"].replace("void gatmi::", "")) + " grid=" + row["Grid_Size"]
            acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    c = {n: sum(x) / len(x) for n, x in acc[k].items()}
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    print(f"{k:60s} cycles {cyc:10.0f}  MFMA util {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.1f} %")
dur = defaultdict(list)
for f in glob.glob("gpurun_out/trace_proj/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gemm_f32" in row["Kernel_Name"]:
            key = re.sub(r"\(.*", "", row["Kernel_Name"].replace("void gatmi::", "")) + " grid=" + str(int(row["Grid_Size_X"]) * int(row["Grid_Size_Y"]) * int(row["Grid_Size_Z"]))
            dur[key].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for k in sorted(dur):
    print(f"{k:60s} avg {sum(dur[k]) / len(dur[k]):9.1f} us")
PY
