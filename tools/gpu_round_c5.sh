#!/bin/bash
# configs[4] geometry (R101-FPN, 832x1344, batch 8, plain fp16): bench line + per-layer table, rocprofv3 kernel stats of the same
# command, and the MFMA-instruction counter pass of tools/profile_step.py. Outputs under gpurun_out/$1/.
set -u
TAG=$1
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C5="--arch resnet101 --height 832 --width 1344 --precision f16 --cpu-images 0 --alt-precision none --alt-config5 0"
step() { local name=$1 limit=$2; shift 2; echo "== $name" >&2; timeout -k 10 $limit "$@"; local rc=$?; echo "== $name rc=$rc" >&2
         if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== $name was killed: stopping"; exit $rc; fi; return $rc; }
step bench 300 python bench.py $C5 --dump-conv $OUT/config5_f16_conv_layers.json > $OUT/config5_f16.json 2> $OUT/config5_f16.err
step stats 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py $C5 --reps 1 > $OUT/kt.json 2> $OUT/kt.err
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/config5_f16_kernel_stats.csv \;
find $OUT/kt -name "*_kernel_trace.csv" -delete
mkdir -p $OUT/pmc_mfma
step pmc_mfma 500 rocprofv3 --pmc SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -o p -- python3 tools/profile_step.py --steps 2 --arch resnet101 --height 832 --width 1344 --precision f16 --meta $OUT/pmc_mfma/meta.json > $OUT/pmc_mfma.log 2>&1 || tail -5 $OUT/pmc_mfma.log
find $OUT/pmc_mfma -name "*_kernel_trace.csv" -delete
f=$(find $OUT/pmc_mfma -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 profiles/summarize_pmc.py mfma $f $OUT/pmc_mfma/meta.json > $OUT/config5_f16_mfma_util.json
find $OUT -name "*counter_collection.csv" -size +20M -delete
head -c 1200 $OUT/config5_f16_mfma_util.json
ls $OUT
