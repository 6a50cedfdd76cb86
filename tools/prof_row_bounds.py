#!/usr/bin/env python3
"""Candidates of the bf16 screen with one margin for the whole call (the longest row's) against a margin per row, on
inputs whose row norms differ: rows scaled by factors from 0.2 to 3 (z inputs of different positions).  Not a test."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "transcoder-sae-circuits_amd"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402

import tsc_hip  # noqa: E402
from bench import make_device_dictionary  # noqa: E402

dev = torch.device("cuda", 0)
W, be, bd = make_device_dictionary(24576, 3, dev)
Wd = torch.randn(24576, 768, device=dev) / 28.0
h = tsc_hip.DeviceDict(W, be, bd, Wd, bd, compute_dtype=tsc_hip.F32_SCREEN)
g = torch.Generator(device=dev).manual_seed(1)
out = {}
for name, scale in (("equal_norms", torch.ones(512, device=dev)), ("norms_0.2_to_3", torch.linspace(0.2, 3.0, 512, device=dev))):
    x = torch.randn(512, 768, generator=g, device=dev) * scale[:, None]
    res = {}
    for per_row in (False, True):
        rb = tsc_hip.screen_row_bound(x, bd, per_row=per_row)
        f, v, c, _, _ = h.sparse_lens_pass(x, 2048, want_recon=False, row_bound=rb, emit_only=True)
        res["per_row" if per_row else "one_margin"] = float(c.float().mean())
    f, v, c, _, _ = h.sparse_lens_pass(x, 2048, want_recon=False, screen=False)
    res["active"] = float(c.float().mean())
    out[name] = res
print(json.dumps(out))
