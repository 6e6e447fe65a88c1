#!/bin/bash
# Run on the GPU box: kernel-trace stats + HBM traffic counters (separate PMC passes) for bench.py.
# usage: tools/collect_profiles.sh <outdir under gpurun_out>
OUT=$GRAFT_REPO_ROOT/${1:-gpurun_out/prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --pipeline-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace_bench.json 2> $OUT/trace_err.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > /dev/null 2> $OUT/pmc_fetch_err.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > /dev/null 2> $OUT/pmc_write_err.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- $B > /dev/null 2> $OUT/pmc_sq_err.txt
python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); from bench import source_fingerprint; print(source_fingerprint())" > $OUT/source_sha256.txt
find $OUT -name "*.csv" | head -20
