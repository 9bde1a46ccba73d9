for wl in products arxiv runpy100k; do
GATMI_LIB=$PWD/gat_amd/libgat_mi355.so timeout -k 10 300 python bench.py --no-cpu-baseline --workload $wl --steps 20 --warmup 5 > gpurun_out/b4_$wl.json 2> gpurun_out/b4_$wl.err && python -c "
import json; r=json.load(open('gpurun_out/b4_$wl.json')); print('$wl ', r['ms_per_step'], r['value']); print({k:v['avg_ms'] for k,v in r['kernels'].items()})"
done
