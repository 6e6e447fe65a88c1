# A/B of tagged builds on one box: tools/lib_ab.sh <out.txt> <kernel-name-substring> <tag> [<tag> ...]   ("-" = the default library)
# BENCH_ARGS="--batch 1" etc. is passed on to bench.py
# bench.py --pipeline-only per library, two interleaved rounds; prints the step and the kernels whose name contains the substring.
out=$1; pat=$2; shift 2
mkdir -p "$(dirname "$out")"
for round in 1 2; do
for t in "$@"; do
  lib=$PWD/l3ac_amd/libl3ac_hip.so; [ "$t" != "-" ] && lib=$PWD/l3ac_amd/libl3ac_hip_$t.so
  L3AC_LIB_PATH=$lib timeout 200 python bench.py --pipeline-only --steps 20 --warmup 3 $BENCH_ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t', round(d['ms_per_step'],3), ' '.join('%s=%.4f' % (e['name'], e['ms']) for e in d['kernels'] if any(p in e['name'] for p in '$pat'.split(','))))
"
done; done 2>&1 | tee "$out"
