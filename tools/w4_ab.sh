#!/bin/bash
# same-box A/B of conv3x3_wino4_f32 builds on the three in-step shapes (tools/w4_busy_probe.py): product vs variant libraries, alternating
#   usage: tools/w4_ab.sh <tag> <variant> [<variant> ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/maskrcnn_amd/csrc/build/variants
for rep in 1 2 3; do
  echo "product:" $(timeout -k 10 120 python3 tools/w4_busy_probe.py 20 2>/dev/null | tr '\n' ' ') | tee -a $OUT/ab.txt
  for v in "$@"; do
    echo "$v:" $(MRCNN_LIB=$V/$v/libmaskrcnn_hip.so timeout -k 10 120 python3 tools/w4_busy_probe.py 20 2>/dev/null | tr '\n' ' ') | tee -a $OUT/ab.txt
  done
done
