#!/bin/bash
# Fabric bytes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) and time of one ResNet C2 identity Bottleneck in its
# three forms -> gpurun_out/$1/bottleneck_traffic.json (copied to profiles/r03_bottleneck_traffic.json).
set -u
OUT=gpurun_out/${1:-bt}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for form in native three whole_block; do
    timeout -k 10 200 python3 tools/bottleneck_block.py $form > $OUT/time_$form.json 2> $OUT/time_$form.err || exit 1
    for c in FETCH_SIZE WRITE_SIZE; do
        d=$OUT/${form}_$c; mkdir -p $d
        timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o p -- python3 tools/bottleneck_block.py $form --meta $d/meta.json > $d/log.txt 2>&1 || { tail -5 $d/log.txt; exit 1; }
        find $d -name "*_kernel_trace.csv" -delete
    done
    ff=$(find $OUT/${form}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
    fw=$(find $OUT/${form}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
    python3 profiles/summarize_pmc.py traffic $ff $fw $OUT/${form}_FETCH_SIZE/meta.json > $OUT/traffic_$form.json || exit 1
    find $OUT -name "*counter_collection.csv" -delete
done
python3 - $OUT <<'PY'
import json, sys
out = sys.argv[1]
res = {"what": "one ResNet C2 identity Bottleneck (model.py:190-211), batch 8 x 256 x 256 x 256 channels: fabric bytes per block "
               "(rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes) and device time per block (hipGraph replay of 20 blocks)",
       "forms": {}}
for form in ("native", "three", "whole_block"):
    t = json.loads(open(f"{out}/time_{form}.json").read().strip().splitlines()[-1])
    tr = json.load(open(f"{out}/traffic_{form}.json"))
    conv = {k: v for k, v in tr["per_kernel"].items() if any(s in k for s in ("conv", "wino", "bottleneck"))}
    res["forms"][form] = {"ms_per_block": t["ms_per_block"], "bytes_per_block": sum(v["hbm_bytes_per_step"] for v in conv.values()),
                          "block_min_bytes": t["block_min_bytes"], "checksum": t["checksum"],
                          "per_kernel": {k: {"launches_per_block": v["launches_per_step"], "bytes_per_block": v["hbm_bytes_per_step"]}
                                         for k, v in conv.items()}}
    res["kernel_source_sha16"] = t["kernel_source_sha16"]
json.dump(res, open(f"{out}/bottleneck_traffic.json", "w"), indent=1)
print(json.dumps({f: (v["ms_per_block"], round(v["bytes_per_block"] / 1e9, 3)) for f, v in res["forms"].items()}))
PY
