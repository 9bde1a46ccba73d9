# A/B runs of the backward with d_W on a side stream (GATMI_DW_OVERLAP) and smaller grids
run() { # label, env...
  label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-layers --steps 20 --warmup 5 > gpurun_out/ov_$label.json 2> gpurun_out/ov_$label.err || { echo "FAILED $label"; tail -3 gpurun_out/ov_$label.err; exit 1; }
  python - $label <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/ov_{sys.argv[1]}.json").read().strip().splitlines()[-1])
k=d["kernels"]
print(f"{sys.argv[1]:28s} step {d['ms_per_step']:.3f}  " + "  ".join(f"{a.replace('inagg_','')} {b['avg_ms']:.3f}" for a,b in k.items()))
PY
}
run base A=1 || exit 1
run b1024 GATMI_BWD_BLOCKS=1024 || exit 1
run b768 GATMI_BWD_BLOCKS=768 || exit 1
run b512 GATMI_BWD_BLOCKS=512 || exit 1
run ov GATMI_DW_OVERLAP=1 || exit 1
run ov_b1024 GATMI_DW_OVERLAP=1 GATMI_BWD_BLOCKS=1024 || exit 1
run ov_b768_w4 GATMI_DW_OVERLAP=1 GATMI_BWD_BLOCKS=768 GATMI_DW_WAVES=4 || exit 1
run ov_b512_w4 GATMI_DW_OVERLAP=1 GATMI_BWD_BLOCKS=512 GATMI_DW_WAVES=4 || exit 1
run ov_w4 GATMI_DW_OVERLAP=1 GATMI_DW_WAVES=4 || exit 1
run base2 A=1 || exit 1
