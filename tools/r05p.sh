set -u
OUT=gpurun_out/r05p; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # run <tag> <env assignments...>
  local tag=$1; shift
  env "$@" timeout -k 10 400 python maskrcnn_amd/build.py > $OUT/build_$tag.log 2>&1
  W4_BUILD=$tag timeout -k 10 300 python tools/w4_race_probe.py 20000 2 >> $OUT/ab.jsonl 2>&1
}
for round in 1 2; do
  run product_$round X=1
  run fixA_$round MRCNN_W4_FIX_A=1
  run fixB_$round MRCNN_W4_FIX_B=1
  run fixAB_$round MRCNN_W4_FIX_A=1 MRCNN_W4_FIX_B=1
done
grep -h differed $OUT/ab.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['flag'], d['mode'], d['differed'])
"
