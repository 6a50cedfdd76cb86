#!/usr/bin/env python3
"""SparseTranscoder.forward on 16384 x 24576: decoder half as the dense product (reference mlp_transcoder.py:128-138)
against tsc_dense_to_sparse + tsc_decode_sparse.  HIP-event milliseconds per call, median of 7.  Not a test."""
import json
import os
import sys
from types import SimpleNamespace

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "transcoder-sae-circuits_amd"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402

import tsc_hip  # noqa: E402
from mlp_transcoder import SparseTranscoder  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
cfg = SimpleNamespace(d_in=768, d_sae=24576, l1_coefficient=1e-3, dtype=torch.float32, device=str(dev), is_transcoder=True,
                      d_out=768, is_sparse_connection=False, use_ghost_grads=False, hook_point="blocks.8.ln2.hook_normalized",
                      hook_point_layer=8)
torch.manual_seed(0)
tc = SparseTranscoder(cfg)
tc.b_enc.fill_(-0.70)
x = torch.randn(m, 768, device=dev)


def timed(fn, n=7):
    fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[n // 2]


out = {"rows": m, "active_per_row": float((tc.encode(x) > 0).sum(1).float().mean())}
out["encode_only_ms"] = timed(lambda: tc.encode(x))
out["forward_sparse_decode_ms"] = timed(lambda: tc(x))
a = tc(x)[0]
tsc_hip.DeviceDict.DECODE_SPARSE_MIN_ROWS = 1 << 60
out["forward_dense_decode_ms"] = timed(lambda: tc(x))
b = tc(x)[0]
out["max_abs_difference_of_sae_out"] = float((a - b).abs().max())
print(json.dumps(out))
