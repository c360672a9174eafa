"""Summarise the rocprofv3 --pmc passes of tools/crop_pmc.sh (both NCHW crop kernels on the BASELINE configs[1] shape):
python profiles/summarize_crop_pmc.py gpurun_out/<tag> <level> > profiles/r03_crop_counters_<level>.json"""
import collections
import csv
import glob
import json
import sys

tag, lvl = sys.argv[1], sys.argv[2]
out = {"shape": f"256 RoIs x 256 ch x 14x14 on a 1x256x{lvl}x{lvl} map (BASELINE configs[1] on that level)",
       "note": "mean over launches 2..6 of each kernel; FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE reports half the bytes of 128-byte "
               "requests: TCC_EA0_RDREQ_128B x 128 B is the fabric read volume); durations from the dispatch timestamps of the "
               "same passes", "kernels": {}}
for f in sorted(glob.glob(f"{tag}/crop_pmc_{lvl}_*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "crop" not in k:
            continue
        k = "crop_forward_nchw_staged" if "staged" in k else "crop_forward_nchw (gather)"
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, d in acc.items():
        e = out["kernels"].setdefault(k, {"us": []})
        e["us"].append(round(sorted(dur[k])[len(dur[k]) // 2], 1))
        for c, v in d.items():
            e[c] = round(sum(v[1:]) / max(1, len(v) - 1), 1)
for k, e in out["kernels"].items():
    e["us"] = round(sum(e["us"]) / len(e["us"]), 1)
    if "TCC_EA0_RDREQ_128B_sum" in e:
        e["fabric_read_MB"] = round((e["TCC_EA0_RDREQ_128B_sum"] * 128 + e.get("TCC_EA0_RDREQ_64B_sum", 0) * 64) / 1e6, 1)
    if "WRITE_SIZE" in e:
        e["write_MB"] = round(e["WRITE_SIZE"] * 1024 / 1e6, 1)
    if "TCC_HIT_sum" in e:
        e["l2_hit_rate"] = round(e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"]), 3)
print(json.dumps(out, indent=1))
