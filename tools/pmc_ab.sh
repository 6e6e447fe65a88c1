#!/bin/bash
# Counter A/B of an environment switch on one box (separate --pmc passes, never combined with a trace):
#   tools/pmc_ab.sh <outdir under gpurun_out> <VAR> <value> [<value> ...]
# per value: bench.py --pipeline-only --steps 3 under an SQ pass and two cache passes; summarise with tools/pmc_ab_summary.py
# PROG="tools/hc_time.py 24480x2048x512" profiles that program (path relative to the repo) instead of bench.py
OUT=$GRAFT_REPO_ROOT/$1; VAR=$2; shift 2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE"
P="$R/bench.py --steps 3 --warmup 1 --pipeline-only $BENCH_ARGS"
[ -n "$PROG" ] && P="$R/$PROG"
for v in "$@"; do
    export $VAR=$v
    D=$OUT/${VAR}_$v
    mkdir -p $D
    rocprofv3 --pmc $SQ --output-format csv -d $D/sq -- python3 $P > /dev/null 2> $D/sq_err.txt
    rocprofv3 --pmc $SQ2 --output-format csv -d $D/sq2 -- python3 $P > /dev/null 2> $D/sq2_err.txt
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $D/tcc -- python3 $P > /dev/null 2> $D/tcc_err.txt
    rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d $D/tcp -- python3 $P > /dev/null 2> $D/tcp_err.txt
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- python3 $P > /dev/null 2> $D/fetch_err.txt
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- python3 $P > /dev/null 2> $D/write_err.txt
    python3 $R/tools/pmc_ab_summary.py $D "$PMC_PATTERN" > $OUT/${VAR}_$v.txt 2>&1
    find $D -name "*.csv" -size +200k -delete   # raw per-dispatch dumps do not travel back (the summary does)
    tail -n +1 $OUT/${VAR}_$v.txt
done
