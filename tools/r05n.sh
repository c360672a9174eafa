set -u
OUT=gpurun_out/r05n; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MRCNN_W4_DIAG_BUILD=1 timeout -k 10 400 python maskrcnn_amd/build.py > $OUT/build.log 2>&1; echo "rebuild rc=$?"
timeout -k 10 900 python tools/w4_diag_soak.py 40000 0 plain_both >> $OUT/diag_soak.jsonl 2>> $OUT/diag_soak.err; echo "soak rc=$?"; tail -1 $OUT/diag_soak.jsonl | cut -c1-300
