#!/usr/bin/env python3
"""Experiment: the grouped lens-pass emit (tsc_sparse_lens_emit_many) over 24 DIFFERENT 24576-wide dictionaries (0.9 GB of
bf16 W_enc, cold) against the same launch over ONE dictionary listed 24 times (38 MB, cache-hot): what cold W costs."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "transcoder-sae-circuits_amd"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402

import tsc_hip  # noqa: E402
from bench import make_device_dictionary  # noqa: E402

dev = torch.device("cuda", 0)
M, cap, n = 512, 512, 24
hs = []
for i in range(n):
    W, be, bd = make_device_dictionary(24576, 100 + i, dev)
    be.fill_(-0.7)
    hs.append(tsc_hip.DeviceDict(W, be, bd, W_dec=torch.zeros(24576, 768, device=dev), compute_dtype=tsc_hip.F32_SCREEN))
g = torch.Generator(device=dev).manual_seed(1)
xs = [torch.randn(M, 768, generator=g, device=dev) for _ in range(n)]
feat = torch.empty(n, M, cap, dtype=torch.int32, device=dev)
val = torch.empty(n, M, cap, device=dev)
count = torch.zeros(n, M, dtype=torch.int32, device=dev)
bounds = torch.stack([tsc_hip.screen_row_bound(x, h._keep[2]) for x, h in zip(xs, hs)])
# (a handle may be listed once per call: "hot" = few dictionaries, whose W_enc stays in the 256-MiB Infinity Cache between calls)
for name, k in (("24 different dictionaries (0.9 GB of bf16 W_enc: cold)", 24), ("4 dictionaries (151 MB: Infinity Cache)", 4),
                ("2 dictionaries (75 MB)", 2)):
    handles, inputs, bnd = hs[:k], xs[:k], bounds[:k].contiguous()
    f, v, c = feat[:k], val[:k], count[:k]
    for _ in range(3):
        c.zero_()
        tsc_hip.sparse_lens_emit_many(handles, inputs, f, v, c, row_bounds=bnd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(10):
        c.zero_()
        e0.record()
        tsc_hip.sparse_lens_emit_many(handles, inputs, f, v, c, row_bounds=bnd)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    print(f"{name}: {tot / 10 * 1e3:.1f} us per call, {tot / 10 * 1e3 / k:.1f} us per dictionary")
