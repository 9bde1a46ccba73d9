# forward / backward edge kernels on the power-law and the uniform-degree graph, streaming and register forward
run() { label=$1; shift; wl=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-extra-layers --steps 10 --warmup 3 > gpurun_out/fa_$label.json 2> gpurun_out/fa_$label.err || { echo "FAILED $label"; tail -3 gpurun_out/fa_$label.err; exit 1; }
  python - $label <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/fa_{sys.argv[1]}.json").read().strip().splitlines()[-1])
k=d["kernels"]
print(f"{sys.argv[1]:28s} step {d['ms_per_step']:.3f}  " + "  ".join(f"{a.replace('inagg_','')} {b['avg_ms']:.3f}" for a,b in k.items()))
PY
}
run pl_auto products A=1 || exit 1
run pl_reg products GATMI_STREAM=0 || exit 1
run un_auto products_uniform A=1 || exit 1
run un_reg products_uniform GATMI_STREAM=0 || exit 1
