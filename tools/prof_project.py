#!/usr/bin/env python3
"""Timing of tsc_project_features (activations of a few selected features over many tokens).  Not a test."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "transcoder-sae-circuits_amd"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402
import tsc_hip  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
M = 65536
x = torch.randn(M, 768, generator=g, device=dev)
bd = torch.randn(768, generator=g, device=dev) * 0.05
for F in (1, 4, 6, 16, 64):
    Pt = torch.randn(F, 768, generator=g, device=dev)
    bias = torch.randn(F, generator=g, device=dev)
    proj = tsc_hip.FeatureProjector(Pt, bias, bd)
    out = proj(x, relu=True)
    ref = torch.relu((x - bd) @ Pt.t() + bias)
    err = float((out - ref).abs().max())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        proj(x, relu=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    by = M * (768 * 4 + F * 4)
    print(f"F={F:3d}: {ms * 1e3:7.1f} us, {by / ms / 1e6:6.0f} GB/s algorithmic, max err vs torch {err:.2e}")
