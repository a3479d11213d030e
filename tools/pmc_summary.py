#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel.  usage: pmc_summary.py DIR [DIR...]"""
import collections, csv, glob, json, sys
out = {}
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
        for k, v in agg.items():
            out.setdefault(k, {}).update(v); out[k]["dispatches"] = len(disp[k])
for k, v in out.items():
    if "VALU" in "".join(v.keys()) and v.get("SQ_WAVE_CYCLES"):
        wc = v["SQ_WAVE_CYCLES"]
        v["_wait_any_frac"] = v.get("SQ_WAIT_ANY", 0) / wc
        v["_wait_inst_frac"] = v.get("SQ_WAIT_INST_ANY", 0) / wc
        v["_active_frac"] = v.get("SQ_ACTIVE_INST_ANY", 0) / wc
        if v.get("SQ_ACTIVE_INST_VALU"): v["_valu_lane_util"] = v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64)
print(json.dumps(out, indent=1))
