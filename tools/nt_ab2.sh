# A/B of cache-policy builds (libl3ac_hip_nt_<tag>.so): bench.py --pipeline-only, two interleaved rounds
mkdir -p gpurun_out/r04j
for round in 1 2; do
for t in "" _nt_g _nt_f _nt_r _nt_gfr; do
  L3AC_LIB_PATH=$PWD/l3ac_amd/libl3ac_hip$t.so timeout 200 python bench.py --pipeline-only --steps 20 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k={e['name']:e['ms'] for e in d['kernels']}
print('lib$t', round(d['ms_per_step'],3), 'gemm_split', k.get('gemm_split_kernel'), 'wide256', k.get('conv_unit_wide_kernel<256>'), 'front256', k.get('dwconv_ln_split_kernel<256>'), 'rowLERP', k.get('row_kernel<LERP,CN>'), 'rowPLAIN', k.get('row_kernel<PLAIN,CN>'), 'dwln', k.get('dwconv_ln_kernel'), 'ring96', k.get('conv_unit_ring_kernel<96>'))
"
done; done 2>&1 | tee gpurun_out/r04j/nt_ab2.txt
