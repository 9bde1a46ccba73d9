set -e
out=gpurun_out/ab9.txt
: > $out
for g in runpy10k runpy100k; do
 for sc in 1 3; do
  for fz in 11 9; do
    echo "== $g x$sc GATMI_FUSED=$fz" >> $out
    GATMI_FUSED=$fz timeout -k 10 120 python tools/layer_bench.py --graph $g --scale $sc --f 25 --heads 8 --channels 30 --steps 30 >> $out 2>&1
  done
 done
done
grep -E "==|ms/step" $out
