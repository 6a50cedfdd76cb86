#!/usr/bin/env python3
"""Experiment: how the lens pass's two kernels scale with the number of rows of one launch (is a grouped launch over
the 24 dictionaries of a pass worth building?).  Run under rocprofv3 --kernel-trace --stats."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "transcoder-sae-circuits_amd"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402

import tsc_hip  # noqa: E402
from bench import make_device_dictionary  # noqa: E402

dev = torch.device("cuda", 0)
W, be, bd = make_device_dictionary(24576, 1, dev)
be.fill_(-0.7)
W_dec = torch.randn(24576, 768, device=dev)
d = tsc_hip.DeviceDict(W, be, bd, W_dec=W_dec, out_bias=torch.zeros(768, device=dev), compute_dtype=tsc_hip.F32_SCREEN)
g = torch.Generator(device=dev).manual_seed(3)
for rows in (512, 1024, 2048, 4096, 12288):
    x = torch.randn(rows, 768, generator=g, device=dev)
    dirs = torch.randn(rows, 768, generator=g, device=dev)
    for _ in range(3):
        d.sparse_lens_pass(x, 512, dirs=dirs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = d.sparse_lens_pass(x, 512, dirs=dirs)
    e1.record()
    torch.cuda.synchronize()
    print(f"rows {rows}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per pass ({e0.elapsed_time(e1) / 10 * 1e3 / rows * 512:.1f} us per 512 rows), "
          f"active per row {float(out[2].float().mean()):.1f}", flush=True)
