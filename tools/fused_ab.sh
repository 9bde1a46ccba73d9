#!/bin/bash
# A/B of the fused row-per-group tiles (GATMI_FUSED=1, default) against the separate kernels (=0)
# on run.py-shaped layers; results into gpurun_out/fused_ab.txt
set -e
out=gpurun_out/fused_ab.txt
: > $out
for g in runpy10k runpy100k runpy1m; do
  for fz in 11 3 0; do
    echo "== $g GATMI_FUSED=$fz" >> $out
    GATMI_FUSED=$fz timeout -k 10 120 python tools/layer_bench.py --graph $g --f 25 --heads 8 --channels 30 --steps 20 >> $out 2>&1
  done
done
for c in 24 32 36 40; do
  for fz in 11 3 0; do
    echo "== runpy100k C=$c GATMI_FUSED=$fz" >> $out
    GATMI_FUSED=$fz timeout -k 10 120 python tools/layer_bench.py --graph runpy100k --f 25 --heads 8 --channels $c --steps 20 >> $out 2>&1
  done
done
cat $out
