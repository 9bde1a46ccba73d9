# One GPU call: bench lines for every workload, rocprofv3 kernel stats and the two PMC passes
# for the default bench command (input layer + hidden layer f32 + hidden layer bf16).  Writes under
# gpurun_out/prof_$TAG/; tools/collect_profiles.py copies the summaries into profiles/.
TAG=${1:-cur}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py > $OUT/bench_products.json 2> $OUT/bench_products.err || exit 1
echo "bench products done"
for wl in arxiv cora runpy10k runpy100k runpy1m; do
  timeout -k 10 200 python bench.py --workload $wl --steps 50 --warmup 10 --hipgraph > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err || exit 1
done
echo "bench others done"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1 || exit 1
echo "stats done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 || exit 1
echo "pmc fetch done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1 || exit 1
echo "pmc write done"
ls $OUT $OUT/stats | head -30
# the reference's own configuration: conv1 on the fused tiles (kernel stats + HBM counters at 10^6 nodes)
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $OUT/stats_runpy -o stats --output-format csv -- python3 bench.py --workload runpy1m --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats_runpy.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_runpy -o pmc --output-format csv -- python3 bench.py --workload runpy1m --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch_runpy.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_runpy -o pmc --output-format csv -- python3 bench.py --workload runpy1m --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_write_runpy.log 2>&1 || exit 1
echo "runpy1m stats + pmc done"
