#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, separate runs) for the conv kernels:
HBM bytes of all conv launches of ONE batch-8 step. gfx950 corrections per MI355X_MICROARCH.md §HBM:
counters are in KiB; FETCH_SIZE under-reports wide coalesced reads by exactly 2x → doubled."""
import csv
import json
import re
import sys


def step_sum(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    # every launch of the conv path: the direct implicit-GEMM kernel, the Winograd kernel and its k-blocking pre-pass
    conv = [r for r in rows if any(k in r["Kernel_Name"] for k in ("conv_igemm_f32", "conv3x3_wino", "stem7x7", "kblock_kernel"))]
    # a step starts at the stem launch (its own kernel, or the GENERIC instantiation of the direct kernel — template
    # argument MODE = 1). bench.py also runs batch-1 passes (head calibration, the final statistics pass): the batch-8
    # steps are the ones whose launches carry the most workgroups in total; the last of them is taken.
    stem_re = re.compile(r"conv_igemm_f32<\d+, \d+, \d+, \d+, \d+, (1|true), \d+>")
    stems = [i for i, r in enumerate(conv) if "stem7x7" in r["Kernel_Name"] or stem_re.search(r["Kernel_Name"])]
    bounds = list(zip(stems, stems[1:] + [len(conv)]))
    size = [sum(int(r["Grid_Size"]) for r in conv[a:b]) for a, b in bounds]
    big = max(size)
    i0, i1 = [bd for bd, sz in zip(bounds, size) if sz == big][-1]
    step = conv[i0:i1]
    return sum(float(r["Counter_Value"]) for r in step) * 1024.0, len(step)


if __name__ == "__main__":
    fetch, n1 = step_sum(sys.argv[1], "FETCH_SIZE")
    write, n2 = step_sum(sys.argv[2], "WRITE_SIZE")
    out = {"conv_launches_per_step": n1, "fetch_bytes_raw": fetch, "fetch_bytes_corrected_x2": 2 * fetch,
           "write_bytes": write, "hbm_bytes_per_step": 2 * fetch + write,
           "hbm_bytes_per_launch_avg": (2 * fetch + write) / max(n1, 1)}
    print(json.dumps(out, indent=1))
