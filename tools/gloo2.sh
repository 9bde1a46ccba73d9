# 2-rank gloo rehearsal of bench.py --gpus 2 on ONE GPU (both ranks share device 0): checks the
# multi-process path end to end; timings are not meaningful.
for mode in ${MODES:-h as xs}; do
echo "== $mode"
GAT_HANG_DUMP=${HANG:-100} GAT_DIST_BACKEND=gloo timeout -k 10 ${LIMIT:-150} python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --workload ${WL:-arxiv} --exchange $mode > gpurun_out/gloo2_$mode.json 2> gpurun_out/gloo2_$mode.err
echo "rc=$?"; tail -c 400 gpurun_out/gloo2_$mode.json; echo; grep -v "amdgpu.ids\|socket.cpp\|Gloo" gpurun_out/gloo2_$mode.err | tail -25
done
