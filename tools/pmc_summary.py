#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per pass, csv output) of tools/prof_encode.py into one JSON:
per counter the per-launch values of the encode kernel, and the derived traffic per launch with the gfx950
corrections of MI355X_MICROARCH.md (FETCH_SIZE is in KiB and counts a wide streaming read at half its bytes: x1024 x2;
WRITE_SIZE in KiB: x1024).  usage: pmc_summary.py OUT.json LABEL=DIR[,DIR...] ...  (LABEL like "24576x65536")"""
import csv
import glob
import json
import os
import sys


def collect(dirs, kernel_substr="encode_f32_kernel"):
    counters = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                if kernel_substr in row["Kernel_Name"]:
                    counters.setdefault(row["Counter_Name"], []).append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    return {k: [v for _, v in sorted(vals)] for k, vals in counters.items()}


def main():
    out_path, specs = sys.argv[1], sys.argv[2:]
    out = {"method": "rocprofv3 --pmc <one counter set per pass> --kernel-trace --output-format csv -- python3 tools/prof_encode.py ...; "
                     "values of the LAST launch of each pass (lists warm); FETCH_SIZE KiB x1024 x2 (gfx950 wide-read correction), "
                     "WRITE_SIZE KiB x1024; these are L2 fabric-side requests, Infinity Cache hits included",
           "shapes": {}}
    for spec in specs:
        label, dirs = spec.split("=", 1)
        d_sae, tokens = (int(v) for v in label.split("x")[:2])
        c = collect(dirs.split(","))
        entry = {"d_sae": d_sae, "tokens_per_launch": tokens, "counters_per_launch": c}
        fetch = c.get("FETCH_SIZE", [None])[-1]
        write = c.get("WRITE_SIZE", [None])[-1]
        if fetch is not None:
            entry["fetch_bytes_per_launch"] = fetch * 1024 * 2
        if write is not None:
            entry["write_bytes_per_launch"] = write * 1024
        if fetch is not None:
            entry["bytes_per_launch"] = entry["fetch_bytes_per_launch"] + entry.get("write_bytes_per_launch", 0.0)
        entry["algorithmic_bytes_per_launch"] = 4 * 768 * (tokens + d_sae)  # x once + W_enc once (lists: written through, small)
        entry["operand_footprint_bytes"] = 4 * 768 * (tokens + d_sae)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CU_CYCLES" in c:
            entry["mfma_busy_fraction"] = c["SQ_VALU_MFMA_BUSY_CYCLES"][-1] / (4 * c["SQ_BUSY_CU_CYCLES"][-1])
        if fetch is not None:
            entry["traffic_over_algorithmic"] = entry["bytes_per_launch"] / entry["algorithmic_bytes_per_launch"]
        out["shapes"][label] = entry
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters_per_launch"} for k, v in out["shapes"].items()}, indent=1))


if __name__ == "__main__":
    main()
