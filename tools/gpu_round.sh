#!/bin/bash
# One gpurun call of a measurement round: GPU tests, bench.py, the rocprofv3 kernel-trace stats of the same bench command,
# and the PMC passes (MFMA instructions / HBM bytes) of tools/profile_step.py. Outputs under gpurun_out/$1/.
# A step that times out or is killed stops the script (no further GPU step after a hang).
#   usage: tools/gpu_round.sh <tag> [tests] [bench] [stats] [pmc] [micro]
set -u
TAG=$1; shift
WHAT=" $* "
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT

step() {  # step <name> <timeout> <cmd...>: stop the whole script when the step was killed
    local name=$1 limit=$2; shift 2
    echo "== $name" >&2
    timeout -k 10 $limit "$@"
    local rc=$?
    echo "== $name rc=$rc" >&2
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== $name was killed: stopping"; exit $rc; fi
    return $rc
}

if [[ "$WHAT" == *" tests "* ]]; then
    step tests 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/tests.log 2>&1
    tail -15 $OUT/tests.log
fi
if [[ "$WHAT" == *" bench "* ]]; then
    step bench 400 python bench.py --dump-conv $OUT/conv_layers.json > $OUT/bench_n1.json 2> $OUT/bench_n1.err
    tail -c 600 $OUT/bench_n1.json
fi
if [[ "$WHAT" == *" stats "* ]]; then
    step stats 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --cpu-images 0 --alt-precision none --in-flight 1 > $OUT/kt.json 2> $OUT/kt.err
    find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
    find $OUT/kt -name "*_kernel_trace.csv" -delete   # large; the stats summary is what is kept
fi
if [[ "$WHAT" == *" pmc "* ]]; then
    for grp in "mfma:SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
        name=${grp%%:*}; ctrs=${grp#*:}
        mkdir -p $OUT/pmc_$name
        step pmc_$name 500 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/pmc_$name -o p -- python3 tools/profile_step.py --steps 2 --meta $OUT/pmc_$name/meta.json > $OUT/pmc_$name.log 2>&1 || tail -5 $OUT/pmc_$name.log
        find $OUT/pmc_$name -name "*_kernel_trace.csv" -delete
    done
    f=$(find $OUT/pmc_mfma -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 profiles/summarize_pmc.py mfma $f $OUT/pmc_mfma/meta.json > $OUT/mfma_util.json
    ff=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1)
    fw=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
    [ -n "$ff" ] && [ -n "$fw" ] && python3 profiles/summarize_pmc.py traffic $ff $fw $OUT/pmc_fetch/meta.json > $OUT/hbm_traffic.json
    find $OUT -name "*counter_collection.csv" -size +20M -delete
    head -c 1500 $OUT/mfma_util.json
fi
if [[ "$WHAT" == *" micro "* ]]; then
    step micro 300 python tools/microbench.py > $OUT/microbench.jsonl 2> $OUT/microbench.err
fi
ls $OUT
