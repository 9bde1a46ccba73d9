# launch-bound shapes: the layer step eager against replayed as one HIP graph (bench.py --hipgraph),
# and an epoch of the 3-layer network (tools/bench_model.py)
for wl in cora runpy10k runpy100k arxiv; do
  timeout -k 10 100 python -u bench.py --workload $wl --steps 200 --warmup 20 --hipgraph --no-cpu-baseline --no-extra-layers > gpurun_out/graph_$wl.json 2> gpurun_out/graph_$wl.err || exit 1
done
for wl in runpy10k runpy100k cora; do
  for g in "" "--hipgraph"; do
    timeout -k 10 100 python -u tools/bench_model.py --workload $wl --hidden 30 --classes 2 --epochs 50 --no-profile $g || exit 1
  done
done
