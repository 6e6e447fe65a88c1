#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/round_end.sh r03'): the -m gpu tests, the rocprofv3 passes (four workloads) and the
# default bench of ONE build, so that profiles/<round>/*traffic.json carry the fingerprint of the sources bench.py runs on.
# Everything is written under gpurun_out/<round>_final/ (the only directory that travels back); copy its profiles/ part
# into profiles/<round>/ and commit.
R=${1:-r03}
OUT=gpurun_out/${R}_final
mkdir -p $OUT/profiles
timeout 2400 python -m pytest tests -m gpu -x -q -s > $OUT/profiles/pytest_gpu.log 2>&1
tail -2 $OUT/profiles/pytest_gpu.log
timeout 1500 bash tools/collect_profiles.sh $OUT/raw > /dev/null 2>&1
python3 tools/summarize_profiles.py $OUT/raw $OUT/profiles | tail -12
timeout 900 python bench.py --profiles-dir $OUT/profiles > $OUT/profiles/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/profiles/bench_default.json
rm -rf $OUT/raw/*/pmc_* # per-dispatch counter dumps: tens of MB
# same-box reference: a library kept from an earlier state of the round (l3ac_amd/libl3ac_hip_base.so, git-ignored) against the final one
if [ -f l3ac_amd/libl3ac_hip_base.so ]; then
  for i in 1 2 3; do for t in base -; do lib=$PWD/l3ac_amd/libl3ac_hip.so; [ $t = base ] && lib=$PWD/l3ac_amd/libl3ac_hip_base.so
    L3AC_LIB_PATH=$lib timeout 300 python bench.py --pipeline-only --steps 60 --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t', round(d['ms_per_step'],3), 'kernel sum', round(sum(e['ms'] for e in d['kernels']),3))
"; done; done | tee $OUT/profiles/final_same_box_ab.txt
fi
