#!/bin/bash
# Ablation of the fused backward's phases (timing only: results are wrong).  Needs a library built with
# -DGATMI_ABLATION (make -C gat_amd/csrc clean all EXTRA=-DGATMI_ABLATION); the tree's build ignores GATMI_FUSE_DBG.
out=gpurun_out/fused_dbg.txt
: > $out
for d in 0 1 2 8 16 32; do
  echo "== bwd GATMI_FUSE_DBG=$d" >> $out
  GATMI_FUSED=2 GATMI_FUSE_DBG=$d timeout -k 10 120 python tools/layer_bench.py --graph runpy1m --f 25 --heads 8 --channels 30 --steps 10 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
