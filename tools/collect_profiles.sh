#!/bin/bash
# Run on the GPU box: kernel-trace stats + HBM traffic + MFMA/VALU counters (separate PMC passes, as the guide's
# rocprofv3 section prescribes: --pmc never combined with a trace domain) for FIVE workloads:
#   main   bench.py --pipeline-only                 1kbps, 256 x 1 s (the headline step)
#   3kbps  bench.py --config 3kbps --pipeline-only  BASELINE config 3
#   b1     tools/b1_profile.py                      one 1 s clip (the streaming chunk of BASELINE config 5), eager
#   vq     tools/vq_argmin_bench.py --quick         explicit-codebook argmin, K = 250 047, N = 42 752
#   fsq    tools/fsq_profile.py                     the quantiser alone at 2^22 tokens (the north_star's HBM kernel) + its copy ceiling
# usage: tools/collect_profiles.sh <outdir under gpurun_out> [workloads...]      (default: all four)
# The program goes directly after `--` (no env / bash -c hop: the profiler has initialised the GPU by then).
OUT=$GRAFT_REPO_ROOT/${1:-gpurun_out/prof}
[ $# -gt 0 ] && shift
WL=${@:-main 3kbps b1 vq fsq}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# the fingerprint of the sources being profiled is written FIRST: a run cut short by a timeout must not be stamped later with
# the fingerprint of whatever tree the summary is made on
python3 -c "import sys; sys.path.insert(0, '$R'); from bench import source_fingerprint; print(source_fingerprint())" > $OUT/source_sha256.txt
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
for w in $WL; do
    case $w in
        main)  P="$R/bench.py --steps 3 --warmup 1 --pipeline-only"; PT="$R/bench.py --steps 12 --warmup 3 --pipeline-only" ;;  # (the trace pass runs as many steps as a bench run: its averages then include the same warm state)
        3kbps) P="$R/bench.py --config 3kbps --steps 3 --warmup 1 --pipeline-only"; PT="$R/bench.py --config 3kbps --steps 12 --warmup 3 --pipeline-only" ;;
        b1)    P="$R/tools/b1_profile.py"; PT=$P ;;
        vq)    P="$R/tools/vq_argmin_bench.py --quick"; PT=$P ;;
        fsq)   P="$R/tools/fsq_profile.py"; PT=$P ;;
        *) echo "unknown workload $w"; continue ;;
    esac
    D=$OUT/$w
    mkdir -p $D
    rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $PT > $D/trace_stdout.txt 2> $D/trace_err.txt
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -- python3 $P > /dev/null 2> $D/pmc_fetch_err.txt
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -- python3 $P > /dev/null 2> $D/pmc_write_err.txt
    rocprofv3 --pmc $SQ --output-format csv -d $D/pmc_sq -- python3 $P > /dev/null 2> $D/pmc_sq_err.txt
done
find $OUT -name "*_kernel_stats.csv" | head -20
