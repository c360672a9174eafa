set -u
OUT=gpurun_out/r05g; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_dist.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -k "not config5 and not level" > $OUT/tests1.log 2>&1; echo "split rc=$?"; tail -4 $OUT/tests1.log
MRCNN_CONV_NO_SPLIT=1 timeout -k 10 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_dist.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -k "not config5 and not level" > $OUT/tests2.log 2>&1; echo "nosplit rc=$?"; tail -4 $OUT/tests2.log
