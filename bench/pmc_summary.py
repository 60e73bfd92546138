#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc CSV (counter_collection + kernel_trace) per kernel: per-wave shares and MFMA busy."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
meta = {}
for r in csv.DictReader(open(cc)):
    if pat in r["Kernel_Name"]:
        agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        meta[r["Dispatch_Id"]] = (r["Kernel_Name"][:60], r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"],
                                  (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for did, c in agg.items():
    name, grid, wg, vg, ag, sg, us = meta[did]
    waves = int(grid) // 64
    out = {"kernel": name, "us": round(us, 1), "waves": waves, "vgpr": vg, "agpr": ag, "sgpr": sg}
    wc = c.get("SQ_WAVE_CYCLES", 0)
    for k, v in c.items():
        out[k] = int(v)
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if k in c:
                out[k + "_share"] = round(c[k] / wc, 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            # WAVE_CYCLES counts quad-cycles summed over waves; 1024 SIMDs, waves/1024 resident waves per SIMD
            simd_cycles = 4.0 * wc / max(1, waves) * 1024
            out["mfma_busy_share"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, 3)
    print(out)
