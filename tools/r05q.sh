set -u
OUT=gpurun_out/r05q; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MRCNN_W4_NO_RACE_FIX=1 timeout -k 10 400 python maskrcnn_amd/build.py > $OUT/build_old.log 2>&1
W4_BUILD=without_fix timeout -k 10 300 python tools/w4_race_probe.py 20000 2 > $OUT/old.jsonl 2>&1; grep differed $OUT/old.jsonl | cut -c1-120
timeout -k 10 400 python maskrcnn_amd/build.py > $OUT/build_new.log 2>&1
timeout -k 10 900 python tools/soak_probe.py 100000 wino4_plain_both,wino4_plain_relu_kblocked,wino4_heads,wino4_conv3 > $OUT/new_soak.jsonl 2>&1; cat $OUT/new_soak.jsonl | cut -c1-200
MRCNN_W4_NO_RACE_FIX=1 timeout -k 10 400 python maskrcnn_amd/build.py > $OUT/build_old2.log 2>&1
W4_BUILD=without_fix_again timeout -k 10 300 python tools/w4_race_probe.py 20000 2 > $OUT/old2.jsonl 2>&1; grep differed $OUT/old2.jsonl | cut -c1-120
