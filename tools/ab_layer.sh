# A/B of alternative builds (ab_libs/<name>.so through GATMI_LIB; "base" = the tree's library) on tools/layer_bench.py:
# ARGS="--f 100 --heads 8 --channels 32" tools/ab_layer.sh base lib1 lib2 ...
for v in "$@"; do
  if [ "$v" = base ]; then unset GATMI_LIB; else export GATMI_LIB=$PWD/ab_libs/$v.so; fi
  echo -n "$v: "
  timeout -k 10 200 python3 -u tools/layer_bench.py ${ARGS:-} 2> gpurun_out/abl_$v.err | tail -n 1 || { echo "FAIL $v"; tail -5 gpurun_out/abl_$v.err; exit 1; }
done
