#!/usr/bin/env python3
"""Fabric-side bytes of one screened config-4 pass from the two rocprofv3 --pmc runs of tools/c4_pmc.sh:
per kernel of the pass the median FETCH_SIZE / WRITE_SIZE per launch, with the gfx950 corrections of
MI355X_MICROARCH.md (FETCH_SIZE in KiB, a wide streaming read counted at half its bytes: x1024 x2; WRITE_SIZE in KiB:
x1024), and their sum.  usage: c4_pmc_summary.py <gpurun_out/DIR> OUT.json"""
import csv
import glob
import json
import os
import statistics
import sys

PASS_KERNELS = {  # kernels of ONE screened direction-lens pass (grouped form), by substring
    "convert_x_group_kernel": "convert_x_group_kernel (x - b_dec -> bf16, 24 dictionaries)",
    "encode_group_kernel": "encode_group_kernel (bf16 screen emit, 24 dictionaries)",
    "sparse_finish_many_kernel<true>": "sparse_finish_many_kernel<SCREENED> (exact recompute + decode + scores)",
    "rowdot_kernel": "rowdot_kernel (auxiliary rows . direction)",
    "lens_assemble_kernel": "lens_assemble_kernel (normalise + top-k)",
}


def per_kernel(directory, counter):
    vals = {}
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] != counter:
                continue
            for sub, label in PASS_KERNELS.items():
                if sub in row["Kernel_Name"]:
                    vals.setdefault(label, []).append(float(row["Counter_Value"]))
    return vals


def main():
    base, out_path = sys.argv[1], sys.argv[2]
    fetch = per_kernel(os.path.join(base, "pmc_FETCH_SIZE"), "FETCH_SIZE")
    write = per_kernel(os.path.join(base, "pmc_WRITE_SIZE"), "WRITE_SIZE")
    out = {"method": "TSC_LENS_STREAMS=1 rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python3 tools/prof_config4.py 2 "
                     "(tools/c4_pmc.sh, one counter per run); per launch, median over the launches of the run; FETCH_SIZE KiB "
                     "x1024 x2 (gfx950 wide-read correction of MI355X_MICROARCH.md), WRITE_SIZE KiB x1024; 512 rows x 24 dictionaries",
           "kernels": {}}
    total = 0.0
    for label in PASS_KERNELS.values():
        if label not in fetch:
            continue
        f = statistics.median(fetch[label]) * 1024 * 2
        w = statistics.median(write.get(label, [0.0])) * 1024
        out["kernels"][label] = {"fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "launches_seen": len(fetch[label])}
        total += f + w
    out["pass_bytes"] = total
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
