set -u
OUT=gpurun_out/r05soak; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 tools/soak_probe.py 20000 c2_f16_block_identity,c2_f16_block_first > $OUT/soak.jsonl 2> $OUT/soak.err; echo "soak rc=$?"; cat $OUT/soak.jsonl; tail -3 $OUT/soak.err
