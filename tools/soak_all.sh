set -u
OUT=gpurun_out/${1:-soak_all}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 tools/soak_probe.py 20000 > $OUT/soak.jsonl 2> $OUT/soak.err; echo "soak rc=$?"; cat $OUT/soak.jsonl; tail -2 $OUT/soak.err
