mkdir -p gpurun_out/r04g
for round in 1 2; do
for t in "" _nt1 _nt3 _nt7 _nt15 _nt31; do
  L3AC_LIB_PATH=$PWD/l3ac_amd/libl3ac_hip$t.so timeout 200 python bench.py --pipeline-only --steps 20 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k={e['name']:e['ms'] for e in d['kernels']}
print('lib$t', round(d['ms_per_step'],3), 'legacy', k.get('legacy_unit_split_kernel'), 'split24', k.get('conv_unit_split_kernel<24>'), 'rowLERP', k.get('row_kernel<LERP,CN>'), 'rowPLAIN', k.get('row_kernel<PLAIN,CN>'), 'first', k.get('first_block_kernel'), 'ring96', k.get('conv_unit_ring_kernel<96>'), 'ring48', k.get('conv_unit_ring_kernel<48>'), 'head', k.get('head_fused_kernel'), 'dwln256', k.get('dwconv_ln_split_kernel<256>'))
"
done; done 2>&1 | tee gpurun_out/r04g/nt_out.txt
