#!/bin/bash
# Busy counters of the three conv3x3_wino4_f32 instantiations (tools/w4_busy_probe.py): two SQ passes, one process each
# (rocprofv3 --pmc with --kernel-trace only), summarised per instantiation: gpurun_out/$1/w4_busy.json
set -u
TAG=${1:-r06busy}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 200 python3 tools/w4_busy_probe.py 5 > $OUT/w4_busy_time.txt 2>&1; cat $OUT/w4_busy_time.txt
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA" \
            "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1)); d=$OUT/busy_pass$i; mkdir -p $d
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -o p -- python3 tools/w4_busy_probe.py 2 > $d.log 2>&1 || tail -3 $d.log
  find $d -name "*_kernel_trace.csv" -delete
done
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/busy_pass*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv3x3_wino4_f32" not in k: continue
        t = k.split("<")[1].split(">")[0].replace(" ", "").split(",")
        name = "heads" if t[1] in ("true", "1") else "conv3" if len(t) > 3 and t[3] in ("true", "1") else "plain"
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
for k, d in out.items():
    wc = d.get("SQ_WAVE_CYCLES")
    if wc:
        d["derived"] = {"per_wave_cycle_fractions (SQ quad-cycle counters / SQ_WAVE_CYCLES)": {c[3:]: round(d[c] / wc, 4) for c in
                        ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS") if c in d}}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d:
            d["derived"]["mfma_busy_cycles_over_4xSQ_WAVE_CYCLES"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * wc), 4)
json.dump(out, open("$OUT/w4_busy.json", "w"), indent=1)
for k, d in out.items(): print(k, json.dumps(d.get("derived")))
PY
