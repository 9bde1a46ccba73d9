for n in r4 mi355 r16; do
  GATMI_LIB=$PWD/gat_amd/libgat_$n.so timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err || exit 1
  python -c "
import json,sys; r=json.load(open('gpurun_out/ab_$n.json')); print('$n', r['ms_per_step'], {k:v['avg_ms'] for k,v in r['kernels'].items() if 'fwd' in k or 'edge' in k})"
done
