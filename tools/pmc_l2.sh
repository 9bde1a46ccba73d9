# L2 requests of the edge kernels: one PMC pass (three TCC counters) over the default bench command.
# requests per second per kernel = TCC_REQ_sum / kernel time (profiles/r2/r2f_kernel_stats.csv)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 300 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmc_l2 -o pmc --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_l2.log 2>&1 || exit 1
python tools/pmc_summary.py gpurun_out/pmc_l2 | grep -i "attn_fwd\|attn_bwd_edge\|inagg_bwd3\|inagg_fwd_stream\|bwd_dst"
