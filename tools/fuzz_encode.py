#!/usr/bin/env python3
"""Randomised numerics check of tsc_encode_dense / tsc_encode_dense_pre against an fp64 restatement of the reference
formula relu((x - b_dec) @ W_enc + b_enc) (src/mlp_transcoder.py:194-213, src/z_sae.py:73-77): fp32 handles within
1e-4 relative of the largest term magnitude (the north star's tolerance), bf16 handles within 2^-7 of the restatement
on bf16-rounded operands.  Every staging variant (d_in 4 ... 1028), odd widths, ragged row counts, row scales.
Usage: fuzz_encode.py [seconds] [seed] [max cases]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "transcoder-sae-circuits_amd"))
import torch  # noqa: E402

import tsc_hip  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_cases = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 60  # stop after this many cases (a deterministic run)
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(seed)
t0, n_cases, worst32, worst16 = time.time(), 0, 0.0, 0.0


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


while time.time() - t0 < budget and n_cases < max_cases:
    d_in = [4, 36, 64, 100, 128, 192, 256, 768, 772, 1028][ri(0, 9)]
    d_sae = [4, 132, 260, 1536, 4100, 24576][ri(0, 5)]
    m = [1, 3, 127, 128, 129, 255, 257, 1000, 4097][ri(0, 8)]
    if m * d_sae > 3e7:
        m = 257
    bf16 = d_in % 128 == 0 and ri(0, 2) == 0
    gd = torch.Generator(device=dev).manual_seed(seed * 15485863 + n_cases)
    W = (torch.randn(d_in, d_sae, generator=gd, device=dev) / d_in ** 0.5).contiguous()
    b_enc = torch.randn(d_sae, generator=gd, device=dev) * 0.3
    b_dec = torch.randn(d_in, generator=gd, device=dev) * [0.0, 0.1, 2.0][ri(0, 2)]
    x = (torch.randn(m, d_in, generator=gd, device=dev) * (0.1 + 3 * torch.rand(m, 1, generator=gd, device=dev))).contiguous()
    d = tsc_hip.DeviceDict(W, b_enc, b_dec, compute_dtype=tsc_hip.BF16 if bf16 else tsc_hip.F32)
    out, pre = d.encode_dense(x), d.encode_pre(x)
    if bf16:
        xc = (x - b_dec).to(torch.bfloat16).double()
        ref = xc @ W.to(torch.bfloat16).double() + b_enc.double()
        mag = (xc.abs() @ W.to(torch.bfloat16).double().abs() + b_enc.double().abs())
        tol = 2.0 ** -7
    else:
        ref = (x.double() - b_dec.double()) @ W.double() + b_enc.double()
        mag = (x.double() - b_dec.double()).abs() @ W.double().abs() + b_enc.double().abs() + (b_dec.double().abs() @ W.double().abs())
        tol = 1e-4
    err_pre = ((pre.double() - ref).abs() / mag.clamp(min=1e-30)).max().item()
    err_out = ((out.double() - ref.clamp(min=0)).abs() / mag.clamp(min=1e-30)).max().item()
    if not (err_pre <= tol and err_out <= tol) or not bool(((out == 0) | (out == pre)).all()):
        print(f"MISMATCH case {n_cases}: d_in {d_in} d_sae {d_sae} m {m} bf16 {bf16}: relative error pre {err_pre:.3e} out {err_out:.3e}", flush=True)
        sys.exit(1)
    if bf16:
        worst16 = max(worst16, err_pre)
    else:
        worst32 = max(worst32, err_pre)
    n_cases += 1
print(f"ok: {n_cases} cases; worst error relative to the term magnitude: fp32 {worst32:.2e} (bar 1e-4), bf16 {worst16:.2e} (bar 7.8e-3) (seed {seed})")
