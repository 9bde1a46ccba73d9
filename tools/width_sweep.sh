#!/bin/bash
# epoch time of the 3-layer network and the hidden layer alone for the reference's search widths
# (VERDICT r2 #3).  usage: tools/width_sweep.sh <outfile> [graph]
out=$1; graph=${2:-products}
: > $out
for h in 24 28 30 32 34 36 40; do
  timeout -k 10 120 python -u tools/bench_model.py --workload $graph --hidden $h --classes 47 --epochs 5 --no-profile >> $out 2>&1 || exit 1
  timeout -k 10 120 python -u tools/layer_bench.py --graph $graph --f $((8*h)) --heads 8 --channels $h --dx >> $out 2>&1 || exit 1
done
