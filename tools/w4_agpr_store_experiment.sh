#!/bin/bash
# Does the rare wrong tile of conv3x3_wino4_f32 (DESIGN 5.1a'') come from LDS stores that source ACCUMULATOR registers (ds_write_b32 v, aN)?
# Builds of the kernel with / without the wait states of the round-5 fix and with the epilogue's Z stores sourcing VGPRs instead,
# each soaked on constant inputs (tools/soak_probe.py). The library is restored to the product build at the end.
set -u
TAG=${1:-w4agpr}; N=${2:-80000}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
soak() {  # soak <name> <env...>
  local name=$1; shift
  env "$@" python3 maskrcnn_amd/build.py > $OUT/build_$name.log 2>&1 || { tail -3 $OUT/build_$name.log; return; }
  timeout -k 10 400 python3 tools/soak_probe.py $N wino4_plain_both,wino4_plain_relu_kblocked 2>/dev/null | sed "s/^{/{\"build\": \"$name\", /" | tee -a $OUT/soak.jsonl | cut -c1-160
}
soak no_fix MRCNN_W4_NO_RACE_FIX=1
soak no_fix_vgpr_stores MRCNN_W4_NO_RACE_FIX=1 MRCNN_W4_Z_FROM_VGPR=1
soak staging_nops_only MRCNN_W4_NO_EPILOGUE_NOPS=1
soak staging_nops_vgpr_stores MRCNN_W4_NO_EPILOGUE_NOPS=1 MRCNN_W4_Z_FROM_VGPR=1
python3 maskrcnn_amd/build.py > $OUT/build_restore.log 2>&1
