# per-shape times of the bf16x3 GEMM launches inside the 256-clip step, for the values of an environment switch (two interleaved rounds):
#   tools/shapes_ab.sh <out.txt> <VAR> <value> [<value> ...]
out=$1; var=$2; shift 2
mkdir -p "$(dirname "$out")"
for round in 1 2; do
for v in "$@"; do
  env $var=$v timeout 200 python bench.py --pipeline-only --steps 20 --warmup 3 $BENCH_ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$v step', round(d['ms_per_step'],3))
for e in d['gemm_shapes']:
    if 'gemm_split' in e['name']: print('   %-48s x%d %.4f ms  %.1f TFLOP/s' % (e['name'], e['launches'], e['ms'], e['tflops']))
"
done; done 2>&1 | tee "$out"
