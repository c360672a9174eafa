set -u
OUT=gpurun_out/r05x; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -8 $OUT/tests.log
cp gpurun_out/parity_fullsize.json $OUT/parity_fullsize.json 2>/dev/null
