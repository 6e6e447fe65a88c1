# A/B of L3AC_WIDE_NT builds (libl3ac_hip_wnt<mask>.so): bench.py --pipeline-only, two interleaved rounds
mkdir -p gpurun_out/r04i
for round in 1 2; do
for t in "" _wnt1 _wnt4 _wnt5 _wnt7; do
  L3AC_LIB_PATH=$PWD/l3ac_amd/libl3ac_hip$t.so timeout 200 python bench.py --pipeline-only --steps 20 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k={e['name']:e['ms'] for e in d['kernels']}
print('lib$t', round(d['ms_per_step'],3), 'wide256', k.get('conv_unit_wide_kernel<256>'), 'wide192', k.get('conv_unit_wide_kernel<192>'), 'front256', k.get('dwconv_ln_split_kernel<256>'), 'gemm_split', k.get('gemm_split_kernel'))
"
done; done 2>&1 | tee gpurun_out/r04i/wide_nt.txt
