#!/bin/bash
# Timing-only ablation builds of bottleneck_c2_f16 (csrc/bottleneck_f16.hip: MRCNN_BF16_ABL), each rebuilt on the box and timed with
# tools/c2_f16_probe.py on both block kinds. The library is restored to the product build at the end.
set -u
TAG=${1:-r05abl}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
for abl in 0 1 2 3 4 8 16 28 31 0; do
  MRCNN_BF16_ABL=$abl python3 maskrcnn_amd/build.py > $OUT/build_$abl.log 2>&1 || { tail -5 $OUT/build_$abl.log; exit 1; }
  for kind in identity first; do
    C2_PROBE_PER_LAYER=0 timeout -k 10 120 python3 tools/c2_f16_probe.py $kind 30 | sed "s/^{/{\"abl\": $abl, /" >> $OUT/ablate.jsonl
  done
done
python3 maskrcnn_amd/build.py > $OUT/build_restore.log 2>&1
cat $OUT/ablate.jsonl
