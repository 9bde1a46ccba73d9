# MFMA utilisation of the fp32 GEMMs: one PMC pass per bench variant (SQ + GRBM counters only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in auto off; do
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -d gpurun_out/pmc_mfma_$v -o pmc --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --input-aggregation $v > gpurun_out/pmc_mfma_$v.log 2>&1 || exit 1
echo "== input-aggregation $v"; python tools/pmc_summary.py gpurun_out/pmc_mfma_$v | grep -i "gemm"
done
