# A/B of alternative builds of the library on the split-GEMM probe: tools/ab_split.sh lib1 lib2 ... (names under
# ab_libs/, loaded through GATMI_LIB; "base" = the tree's library); prints each kernel's rocprofv3 average
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset GATMI_LIB; else export GATMI_LIB=$PWD/ab_libs/$v.so; fi
  rm -rf gpurun_out/ab_$v
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d gpurun_out/ab_$v -o p --output-format csv -- python3 -u tools/split_probe.py --sample 50000 --masks ${MASKS:-3} > gpurun_out/ab_$v.log 2>&1 || { echo "FAIL $v"; tail -5 gpurun_out/ab_$v.log; exit 1; }
  python3 - "$v" <<'PY'
import csv, sys
v = sys.argv[1]
for r in csv.DictReader(open(f"gpurun_out/ab_{v}/p_kernel_stats.csv")):
    if "kres" in r["Name"] or "dw_bf16_kernel" in r["Name"] or "gemm_f32" in r["Name"]:
        print(v, r["Name"][12:60], r["Calls"], round(float(r["AverageNs"]) / 1e6, 3), "ms")
PY
done
