#!/usr/bin/env python3
"""How evenly the screened top-k's candidate queues fill (one queue per workgroup of K1 / K2): per call the mean and the
maximum of the per-workgroup counts, and K2's share of the call.  Not a test."""
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
d_sae, B = 24576, 65536
W, be, bd = make_device_dictionary(d_sae, 1, dev)
d = tsc_hip.DeviceDict(W, be, bd, compute_dtype=tsc_hip.F32_SCREEN)
st = tsc_hip.TopkState(d_sae, 20, dev)
g = torch.Generator(device=dev).manual_seed(3)
xs = [torch.randn(B, 768, generator=g, device=dev) for _ in range(4)]
d.encode_topk(xs[0][:16384], 0, st)
n_wg = (d_sae // 256) * st.n_slots
for i in range(1, 12):
    x = xs[i % 4]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    d.encode_topk_screened(x, i * B, st)
    e1.record()
    torch.cuda.synchronize()
    ws = st.screen_workspace()
    c = ws.view(torch.uint8)[256:256 + 4 * n_wg].view(torch.int32).float()
    print(json.dumps({"call": i, "ms": e0.elapsed_time(e1), "candidates": int(c.sum()), "mean_per_queue": float(c.mean()),
                      "max_per_queue": float(c.max()), "max_over_mean": float(c.max() / c.mean())}), flush=True)
