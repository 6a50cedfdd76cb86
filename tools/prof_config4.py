#!/usr/bin/env python3
"""Profiling target for BASELINE configs[3] (the batched CircuitLens pass): builds the bench's config-4 world once
and runs the pass a few times.  Used under rocprofv3 --kernel-trace --stats; not a test."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "transcoder-sae-circuits_amd"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n_prompts = int(sys.argv[2]) if len(sys.argv) > 2 else 512
print(json.dumps(bench.config4_side_object(torch.device("cuda", 0), n_prompts=n_prompts, reps=reps)))

import circuit_lens  # noqa: E402

if circuit_lens._TRACE:  # TSC_LENS_TRACE=1: phase times of the last screened pass (device: between events; host: perf_counter)
    tr = circuit_lens._TRACE
    starts = [i for i, t in enumerate(tr) if t[0] == "lens: start"]
    # the screened passes come first, then the plain ones: take the last screened pass = pass number reps (0-based: warm-up is 0)
    i0 = starts[reps]
    i1 = starts[reps + 1] if reps + 1 < len(starts) else len(tr)
    seg = tr[i0:i1]
    for (la, ea, ha), (lb, eb, hb) in zip(seg[:-1], seg[1:]):
        print(f"  {la:26s} -> {lb:26s} device {ea.elapsed_time(eb) * 1e3:8.1f} us   host {(hb - ha) * 1e6:8.1f} us", file=sys.stderr)
