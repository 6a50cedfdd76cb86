#!/usr/bin/env python3
"""Micro-benchmark of tsc_lens_assemble at config-4 shape (512 rows, 24 groups, capacity 512)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "transcoder-sae-circuits_amd"))
import torch  # noqa: E402

import tsc_hip  # noqa: E402

dev = torch.device("cuda", 0)
R, G, cap = 512, 24, 512
g = torch.Generator(device=dev).manual_seed(0)
scores = torch.randn(G, R, cap, generator=g, device=dev)
count = torch.randint(40, 100, (G, R), generator=g, device=dev, dtype=torch.int32)
head = torch.randn(R, 2, generator=g, device=dev)
aux = torch.randn(R, 60, generator=g, device=dev)
for k, dense in ((10, True), (10, False), (1, True), (1, False), (40, False)):
    for _ in range(3):
        tsc_hip.lens_assemble(head, aux, [3, 2] * 12, scores, count, k=k, dense=dense)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        tsc_hip.lens_assemble(head, aux, [3, 2] * 12, scores, count, k=k, dense=dense)
    e1.record()
    torch.cuda.synchronize()
    print(f"k={k} dense={dense}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us (allocation of the outputs included)")
