set -u
OUT=gpurun_out/r05o; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  timeout -k 10 400 python maskrcnn_amd/build.py > $OUT/build_p$round.log 2>&1
  W4_BUILD=product_$round timeout -k 10 300 python tools/w4_race_probe.py 20000 2 >> $OUT/ab.jsonl 2>&1
  MRCNN_W4_DIAG_BUILD=1 timeout -k 10 400 python maskrcnn_amd/build.py > $OUT/build_d$round.log 2>&1
  W4_BUILD=diag0_$round timeout -k 10 300 python tools/w4_race_probe.py 20000 2 >> $OUT/ab.jsonl 2>&1
  grep -h differed $OUT/ab.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['flag'], d['mode'], d['differed'])
" | tail -4
done
