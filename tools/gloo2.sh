# 2-rank gloo rehearsal of bench.py --gpus 2 on ONE GPU (both ranks share device 0): checks the
# multi-process path end to end.  Timings are meaningless: two processes time-slice the card, and a
# kernel that takes 0.3 ms alone was seen to take 4-20 s whenever the other rank's kernels were
# resident too -- hence the generous limits.
for mode in ${MODES:-as xs h}; do
echo "== $mode"
GAT_HANG_DUMP=${HANG:-280} GAT_DIST_BACKEND=gloo timeout -k 10 ${LIMIT:-300} python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --workload ${WL:-arxiv} --exchange $mode --partition ${PART:-edges} ${EXTRA:-} > gpurun_out/gloo2_$mode.json 2> gpurun_out/gloo2_$mode.err
rc=$?
echo "rc=$rc"; tail -c 900 gpurun_out/gloo2_$mode.json; echo; grep -v "amdgpu.ids\|socket.cpp\|Gloo" gpurun_out/gloo2_$mode.err | tail -25
if [ $rc -ne 0 ]; then exit $rc; fi   # after a failure or a time limit nothing more runs on the GPU
done
