#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06e; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -p no:cacheprovider -k "detection_agreement_rate" > $OUT/test_rate.log 2>&1; tail -12 $OUT/test_rate.log
cp gpurun_out/parity_fullsize.json $OUT/parity_rate.json 2>/dev/null
bash tools/w4_busy_pmc.sh r06e
