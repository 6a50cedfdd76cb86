#!/usr/bin/env python3
"""Timing of the CircuitLens contribution pass (BASELINE config 4 shape): sparse emit, sparse decode and
contribution scores for M positions of one dictionary.  Prints per-kernel time and the algorithmic HBM rate
(n_active * (d_out*4 + 8) + n_dir * d_out * 4 bytes per position).  Not a test."""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "transcoder-sae-circuits_amd"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--d-sae", type=int, default=24576)
ap.add_argument("--positions", type=int, default=512)
ap.add_argument("--cap", type=int, default=512)
ap.add_argument("--n-dir", type=int, default=1)
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()

import tsc_hip  # noqa: E402
from bench import make_device_dictionary  # noqa: E402

dev = torch.device("cuda", 0)
W, be, bd = make_device_dictionary(args.d_sae, 1, dev)
g = torch.Generator(device=dev).manual_seed(7)
Wd = torch.randn(args.d_sae, 768, generator=g, device=dev)
Wd /= Wd.norm(dim=1, keepdim=True)
d = tsc_hip.DeviceDict(W, be, bd, Wd, bd)
x = torch.randn(args.positions, 768, generator=g, device=dev)
dirs = torch.randn(args.n_dir, 768, generator=g, device=dev)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps, out


t_enc, (feat, val, count) = timed(lambda: d.encode_sparse(x, args.cap))
n_active = int(count.sum())
t_dec, _ = timed(lambda: d.decode_sparse(feat, val, count))
t_con, _ = timed(lambda: d.contrib_scores(feat, val, count, dirs))
row_bytes = n_active * (768 * 4 + 8)
print(f"positions={args.positions} d_sae={args.d_sae} cap={feat.shape[1]} active/position={n_active / args.positions:.1f}")
print(f"encode_sparse  {t_enc * 1e3:9.1f} us  ({2.0 * 768 * args.d_sae * args.positions / t_enc / 1e9:.1f} TFLOP/s incl. count reset + sort)")
print(f"decode_sparse  {t_dec * 1e3:9.1f} us  ({(row_bytes + args.positions * 768 * 8) / t_dec / 1e6:.0f} GB/s algorithmic)")
print(f"contrib n_dir={args.n_dir} {t_con * 1e3:7.1f} us  ({(row_bytes + args.n_dir * 768 * 4) / t_con / 1e6:.0f} GB/s algorithmic)")
