#!/bin/bash
# Round-4 supplementary evidence, one gpurun call (outputs under gpurun_out/r04x/):
#  (1) LDS activity counters of the F(4x4) heads launch on the P2 RPN layer (what the cross-wave probe says is NOT its bound)
#  (2) rocprofv3 kernel stats of bench.py as ONE rank of an nccl process group (RANK / WORLD_SIZE exported here, no launcher
#      between rocprofv3 and python): the RCCL all-gather of the detections shows up among the step's kernels
set -u
OUT=gpurun_out/r04x; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
d=$OUT/w4_lds; mkdir -p $d
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $d -o p -- python3 tools/w4_time.py 8 256 256 256 512 > $d.log 2>&1 || tail -3 $d.log
find $d -name "*_kernel_trace.csv" -delete
python3 - <<PY > $OUT/w4_lds_counters.json
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob("$OUT/w4_lds/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "wino4_f32" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, dct in acc.items():
        m = {c: sum(v[1:]) / max(1, len(v) - 1) for c, v in dct.items()}
        if m.get("SQ_BUSY_CYCLES"):
            m["lds_idx_active_over_busy"] = m.get("SQ_LDS_IDX_ACTIVE", 0) / m["SQ_BUSY_CYCLES"]
            m["bank_conflict_over_lds_active"] = m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, m.get("SQ_LDS_IDX_ACTIVE", 0))
        out[k] = m
print(json.dumps(out, indent=1))
PY
cat $OUT/w4_lds_counters.json | head -40
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 MRCNN_FORCE_COLLECTIVE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_nccl -o kt -- python3 bench.py --gpus 1 --steps 10 --reps 1 --cpu-images 0 --alt-precision none --alt-config5 0 --in-flight 1 --measure-traffic 0 --roofline-steps 0 > $OUT/bench_nccl_rank.json 2> $OUT/bench_nccl_rank.err || tail -5 $OUT/bench_nccl_rank.err
find $OUT/kt_nccl -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_nccl_world1.csv \;
find $OUT/kt_nccl -name "*_kernel_trace.csv" -delete
grep -i "nccl\|rccl" $OUT/kernel_stats_nccl_world1.csv | cut -c1-200
head -c 400 $OUT/bench_nccl_rank.json
