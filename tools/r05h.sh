set -u
OUT=gpurun_out/r05h; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/w4_race_probe.py 3000 2 > $OUT/race_default.jsonl 2>&1; cat $OUT/race_default.jsonl | cut -c1-600
MRCNN_W4_VMCNT0=1 timeout -k 10 400 python maskrcnn_amd/build.py > $OUT/build.log 2>&1; echo "rebuild rc=$?"
W4_BUILD=vmcnt0 timeout -k 10 300 python tools/w4_race_probe.py 3000 2 > $OUT/race_vmcnt0.jsonl 2>&1; cat $OUT/race_vmcnt0.jsonl | cut -c1-600
