# As tools/env_ab.sh, but prints the per-shape GEMM lines (bench.py's gemm_shapes) whose name contains the substring.
out=$1; pat=$2; var=$3; shift 3
mkdir -p "$(dirname "$out")"
for round in 1 2; do
for v in "$@"; do
  env $var=$v timeout 200 python bench.py --pipeline-only --steps 20 --warmup 3 $BENCH_ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$v', round(d['ms_per_step'],3), ' '.join('%s[%d]=%.4f' % (e['name'], e['launches'], e['ms']) for e in d['gemm_shapes'] if any(p in e['name'] for p in '$pat'.split(','))))
"
done; done 2>&1 | tee "$out"
