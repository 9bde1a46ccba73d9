# Where do the waves of the fused run.py-shaped kernels spend their time?  SQ passes over tools/layer_bench.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
G=${1:-runpy1m}
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM -d gpurun_out/pmc_fused_sq1 -o pmc --output-format csv -- python3 tools/layer_bench.py --graph $G --f 25 --heads 8 --channels 30 --steps 2 > gpurun_out/pmc_fused_sq1.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAVES SQ_BUSY_CYCLES -d gpurun_out/pmc_fused_sq2 -o pmc --output-format csv -- python3 tools/layer_bench.py --graph $G --f 25 --heads 8 --channels 30 --steps 2 > gpurun_out/pmc_fused_sq2.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_FLAT SQ_VALU_MFMA_BUSY_CYCLES -d gpurun_out/pmc_fused_sq3 -o pmc --output-format csv -- python3 tools/layer_bench.py --graph $G --f 25 --heads 8 --channels 30 --steps 2 > gpurun_out/pmc_fused_sq3.log 2>&1 || exit 1
python3 tools/pmc_summary.py gpurun_out/pmc_fused_sq1 gpurun_out/pmc_fused_sq2 gpurun_out/pmc_fused_sq3 | grep -E "fwdgf|bwdgf|fwdg_|bwdg_" | tr -s ' ' | sed 's/  /\n    /g' > gpurun_out/pmc_fused.txt
cat gpurun_out/pmc_fused.txt
