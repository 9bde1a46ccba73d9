# A/B of alternative builds of the library on a bench.py workload: WL=products_hidden tools/ab_bench.sh base lib1 ...
# (names under ab_libs/, loaded through GATMI_LIB; "base" = the tree's library); prints ms/step and kernel averages
WL=${WL:-products_hidden}
for v in "$@"; do
  if [ "$v" = base ]; then unset GATMI_LIB; else export GATMI_LIB=$PWD/ab_libs/$v.so; fi
  timeout -k 10 200 python3 -u bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-extra-layers ${EXTRA:-} > gpurun_out/abb_$v.json 2> gpurun_out/abb_$v.err || { echo "FAIL $v"; tail -5 gpurun_out/abb_$v.err; exit 1; }
  python3 - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads(open(f"gpurun_out/abb_{v}.json").read().strip().splitlines()[-1])
print(v, d["ms_per_step"], {k: x["avg_ms"] for k, x in d.get("kernels", {}).items() if x["avg_ms"] > 0.3})
PY
done
