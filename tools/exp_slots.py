#!/usr/bin/env python3
"""Fused top-k rate against the number of list slots (= token-tile groups per column tile, EncodeArgs::m_split): with
s slots the 32 CUs of an XCD hold 32/s column tiles x s token-tile streams instead of 32 column tiles.
usage: exp_slots.py d_sae tokens bf16(0|1) slots...   Not a test."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "transcoder-sae-circuits_amd"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402

import tsc_hip  # noqa: E402
from bench import D_IN, make_device_dictionary  # noqa: E402

d_sae, tokens, bf16 = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
W_enc, b_enc, b_dec = make_device_dictionary(d_sae, 555, dev)
d = tsc_hip.DeviceDict(W_enc, b_enc, b_dec, compute_dtype=tsc_hip.BF16 if bf16 else tsc_hip.F32)
g = torch.Generator(device=dev).manual_seed(777)
xs = [torch.randn(tokens, D_IN, generator=g, device=dev) for _ in range(4)]
for slots in [int(v) for v in sys.argv[4:]]:
    st = tsc_hip.TopkState(d_sae, 20, dev, n_slots=slots)
    n = 0
    for _ in range(3):
        d.encode_topk(xs[n % 4], n * tokens, st); n += 1
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L = 8
    for _ in range(L):
        d.encode_topk(xs[n % 4], n * tokens, st); n += 1
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / L
    print(json.dumps({"d_sae": d_sae, "tokens": tokens, "bf16": bf16, "slots": slots, "ms_per_launch": t * 1e3,
                      "TFLOP_per_s": 2.0 * D_IN * d_sae * tokens / t / 1e12}), flush=True)
    del st
