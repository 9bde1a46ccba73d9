#!/bin/bash
# The products-only backward pass at two channel tiles per head (C = 34 .. 40): GATMI_FUSED = 11 against 3
# on run.py-shaped layers of 10^5 and 10^6 nodes; results into gpurun_out/wide_ab.txt
set -e
out=gpurun_out/wide_ab.txt
: > $out
for g in runpy100k runpy1m; do
  for c in ${WIDE_C:-34 36 38 40}; do
    for fz in 11 3; do
      echo "== $g C=$c GATMI_FUSED=$fz" >> $out
      GATMI_FUSED=$fz timeout -k 10 120 python tools/layer_bench.py --graph $g --f 25 --heads 8 --channels $c --steps 20 >> $out 2>&1
    done
  done
done
cat $out
