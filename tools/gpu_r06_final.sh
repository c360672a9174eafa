#!/bin/bash
# round-6 final measurements, part $1 (two gpurun calls: a = tests + bench + kernel stats + counter passes + torchrun N = 1;
# b = microbenchmarks, configs[4] fp16 line + stats + counters, soak of every heavy kernel)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "$1" = a ]; then
  bash tools/gpu_round.sh r06z tests bench stats pmc || exit $?
  OUT=gpurun_out/r06z
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --cpu-images 0 --alt-precision none --alt-config5 0 --roofline-steps 0 > $OUT/bench_torchrun_n1.json 2> $OUT/bench_torchrun_n1.err
  python3 -c "
import json
for f in ('bench_n1','bench_torchrun_n1'):
    try:
        d=json.loads(open('$OUT/'+f+'.json').read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
    except Exception as e: print(f, 'failed', e)
"
else
  bash tools/gpu_round.sh r06z micro || exit $?
  bash tools/gpu_round_c5.sh r06z_c5 || exit $?
  bash tools/soak_all.sh r06z_soak
fi
