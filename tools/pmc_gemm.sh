#!/bin/bash
# PMC counters of the GEMM micro-benchmark (run on the GPU box): tools/pmc_gemm.sh <outdir>
set -e
OUT=${1:-gpurun_out/pmc_gemm}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$OUT -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py > $GRAFT_REPO_ROOT/$OUT/stdout.txt 2> $GRAFT_REPO_ROOT/$OUT/stderr.txt || true
ls -R $GRAFT_REPO_ROOT/$OUT | head
