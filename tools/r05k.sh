set -u
OUT=gpurun_out/r05k; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MRCNN_W4_DIAG_BUILD=1 timeout -k 10 400 python maskrcnn_amd/build.py > $OUT/build.log 2>&1; echo "rebuild rc=$?"
timeout -k 10 1000 python tools/w4_diag_soak.py 60000 > $OUT/diag_soak.jsonl 2> $OUT/diag_soak.err; echo "soak rc=$?"; cat $OUT/diag_soak.jsonl | cut -c1-400
