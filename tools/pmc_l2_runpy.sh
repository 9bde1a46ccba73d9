# L2 requests / hits / misses of run.py's conv1 kernels at 10^6 nodes (fused forward, products-only pass, edge
# kernel): one PMC pass of three TCC counters; requests per second = TCC_REQ_sum / kernel time
# (profiles/r3/r3d_runpy1m_kernel_stats.csv)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 200 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmc_l2_runpy -o pmc --output-format csv -- python3 bench.py --workload runpy1m --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_l2_runpy.log 2>&1 || exit 1
python tools/pmc_summary.py gpurun_out/pmc_l2_runpy | grep -i "inagg_"
