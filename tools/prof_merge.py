#!/usr/bin/env python3
"""Timing of tsc_topk_finalize / tsc_topk_merge on a filled state.  Not a test."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "transcoder-sae-circuits_amd"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402
import tsc_hip  # noqa: E402

dev = torch.device("cuda", 0)
for d_sae, k, slots in ((24576, 20, 8), (49152, 20, 4), (24576, 64, 16), (131072, 20, 1)):
    st = tsc_hip.TopkState(d_sae, k, dev, n_slots=slots)
    st.vals.copy_(torch.rand_like(st.vals) + 0.1)
    st.idx.copy_(torch.randint(0, 1 << 40, st.idx.shape, device=dev))
    st.finalize()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        v, i = st.finalize()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    by = slots * d_sae * k * 12 + d_sae * k * 12
    ok = bool((v[:, :-1] >= v[:, 1:]).all())
    print(f"d_sae={d_sae} k={k} slots={slots}: {ms * 1e3:.0f} us, {by / ms / 1e6:.0f} GB/s algorithmic, sorted={ok}")
