#!/usr/bin/env python3
"""The grouped bf16 screen emit of a CircuitLens pass by itself: 24 dictionaries at their real widths (bench.py's),
R rows each, N calls, HIP-event time per call.  TSC_HIP_LIBRARY selects an ablation build (tools/build_variant.sh).  Not a test."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "transcoder-sae-circuits_amd"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402

import tsc_hip  # noqa: E402
from bench import dictionary_widths, make_device_dictionary  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 512
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 10
b_enc = float(sys.argv[3]) if len(sys.argv) > 3 else -0.70
dev = torch.device("cuda", 0)
handles, xs, keep = [], [], []
g = torch.Generator(device=dev).manual_seed(5)
for i, (_, _, d_sae) in enumerate(dictionary_widths()):
    W, be, bd = make_device_dictionary(d_sae, 100 + i, dev)
    be.fill_(b_enc)
    Wd = torch.zeros(d_sae, 768, device=dev)
    handles.append(tsc_hip.DeviceDict(W, be, bd, Wd, bd, compute_dtype=tsc_hip.F32_SCREEN))
    xs.append(torch.randn(rows, 768, generator=g, device=dev))
bounds = torch.stack([tsc_hip.screen_row_bound(x, h._keep[2]) for x, h in zip(xs, handles)])
cap = 512
feat = torch.empty(len(handles), rows, cap, dtype=torch.int32, device=dev)
val = torch.empty(len(handles), rows, cap, dtype=torch.float32, device=dev)
count = torch.zeros(len(handles), rows, dtype=torch.int32, device=dev)
grp = tsc_hip.LensGroup(handles, xs)
times = []
for i in range(calls + 2):
    count.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    grp.emit(feat, val, count, row_bounds=bounds)
    e1.record()
    torch.cuda.synchronize()
    if i >= 2:
        times.append(e0.elapsed_time(e1) * 1e3)
flops = sum(2.0 * 768 * w * rows for _, _, w in dictionary_widths())
times.sort()
med = times[len(times) // 2]
print(json.dumps({"rows": rows, "us_per_call_median": med, "us_min": times[0], "TFLOP_per_s": flops / med / 1e6,
                  "candidates_per_row_and_dictionary": float(count.float().mean()),
                  "library": os.path.basename(tsc_hip.library_path())}))
