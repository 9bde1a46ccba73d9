# A/B of the narrow last-layer kernels: rows in flight (ab_libs/un4.so), grid cap
for lib in base un4; do for nb in 256 512 1024 2048; do
  if [ "$lib" = base ]; then unset GATMI_LIB; else export GATMI_LIB=$PWD/ab_libs/$lib.so; fi
  echo "== $lib blocks<=$nb"
  GATMI_NARROW_BLOCKS=$nb timeout -k 10 200 python tools/bench_model.py --workload runpy1m --hidden 30 --classes 2 --epochs 10 2>&1 | grep -E "epoch|narrow|reduce_dweight"
done; done
