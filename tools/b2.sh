timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 60 --timeout-method thread -k "input_aggregation" > gpurun_out/t_inagg.log 2>&1; tail -3 gpurun_out/t_inagg.log
for wl in products arxiv runpy100k; do
timeout -k 10 300 python bench.py --no-cpu-baseline --workload $wl --steps 20 --warmup 5 > gpurun_out/b2_$wl.json 2> gpurun_out/b2_$wl.err && python -c "
import json; r=json.load(open('gpurun_out/b2_$wl.json')); print('$wl', r['ms_per_step'], r['value']); print({k:v['avg_ms'] for k,v in r['kernels'].items()})"
done
