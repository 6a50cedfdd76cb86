#!/usr/bin/env python3
"""Median per launch of every counter of tools/c4_pmc2.sh for the kernels of the screened config-4 pass.
usage: c4_pmc2_summary.py <gpurun_out/DIR> OUT.json"""
import csv
import glob
import json
import os
import statistics
import sys

KERNELS = ["convert_x_group_kernel", "encode_group_kernel", "sparse_finish_many_kernel<true>", "rowdot_kernel", "lens_assemble_kernel"]
base, out_path = sys.argv[1], sys.argv[2]
vals = {}
for path in glob.glob(os.path.join(base, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        for k in KERNELS:
            if k in row["Kernel_Name"]:
                vals.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
out = {"method": "TSC_LENS_STREAMS=1 rocprofv3 --pmc <one set per run> --kernel-trace -- python3 tools/prof_config4.py 2 (tools/c4_pmc2.sh); "
                 "median per launch; 512 rows x 24 dictionaries", "kernels": {}}
for k, cs in vals.items():
    e = {c: statistics.median(v) for c, v in cs.items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("SQ_BUSY_CU_CYCLES"):
        e["mfma_busy_fraction"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * e["SQ_BUSY_CU_CYCLES"])
    if e.get("SQ_ACTIVE_INST_LDS"):
        e["lds_bank_conflict_fraction"] = e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_ACTIVE_INST_LDS"]
    if e.get("TCC_REQ_sum"):
        e["l2_hit_fraction"] = e.get("TCC_HIT_sum", 0.0) / (e.get("TCC_HIT_sum", 0.0) + e.get("TCC_MISS_sum", 0.0) or 1.0)
    out["kernels"][k] = e
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out, indent=1))
