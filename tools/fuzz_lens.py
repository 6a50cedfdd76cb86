#!/usr/bin/env python3
"""Randomised checks of tsc_sparse_lens_pass: (1) plain fp32 handles -- the emitted (feature, value) lists are exactly
the positives of the library's own dense activations, value-descending, and recon / scores agree with a torch
restatement on them; (2) TSC_F32_SCREEN handles -- the bf16 screen + exact recompute loses no feature whose fp32
activation is above the rounding noise of the two summation orders, and invents none.  Random shapes, bias scales,
row scales, b_dec magnitudes (the screen's error bound is what is under test).  Usage: fuzz_lens.py [seconds] [seed] [max cases]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "transcoder-sae-circuits_amd"))
import torch  # noqa: E402

import tsc_hip  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_cases = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 60  # stop after this many cases (a deterministic run)
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(seed)
t0, n_cases, n_active = time.time(), 0, 0


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


def fail(msg):
    print("MISMATCH " + msg, flush=True)
    sys.exit(1)


while time.time() - t0 < budget and n_cases < max_cases:
    d_in = [4, 36, 128, 256, 768][ri(0, 4)]
    d_sae = [4, 8, 260, 1536, 4100, 24576][ri(0, 5)]
    d_out = [4, 8, 64, 256, 768][ri(0, 4)]
    m = [1, 5, 130, 512, 1000][ri(0, 4)]
    scale = [1.0, 1e-2, 20.0][ri(0, 2)]
    bdec_scale = [0.0, 0.1, 4.0][ri(0, 2)]
    gd = torch.Generator(device=dev).manual_seed(seed * 7919 + n_cases)
    W = (torch.randn(d_in, d_sae, generator=gd, device=dev) / d_in ** 0.5).contiguous()
    b_dec = torch.randn(d_in, generator=gd, device=dev) * bdec_scale * scale
    b_enc = -(torch.rand(d_sae, generator=gd, device=dev) * 1.0 + (1.8 if d_sae > 64 else 0.0)) * scale  # a percent or two of the features fire (more in tiny dictionaries)
    W_dec = torch.randn(d_sae, d_out, generator=gd, device=dev).contiguous()
    out_bias = torch.randn(d_out, generator=gd, device=dev)
    x = torch.randn(m, d_in, generator=gd, device=dev) * scale * (0.2 + 2.8 * torch.rand(m, 1, generator=gd, device=dev))
    if ri(0, 1):
        x = x + b_dec
    x = x.contiguous()
    dirs = torch.randn(m, d_out, generator=gd, device=dev)
    cap = 2048 if d_sae > 2048 else min(2048, d_sae)
    plain = tsc_hip.DeviceDict(W, b_enc, b_dec, W_dec=W_dec, out_bias=out_bias)
    fa, va, ca, ra, sa = plain.sparse_lens_pass(x, cap, dirs=dirs)
    if int(ca.max()) > cap:
        continue
    dense = plain.encode_dense(x)
    # (1) exactly the positives of the own dense activations, sorted
    pos = dense > 0
    if not torch.equal(pos.sum(1).int(), ca):
        fail(f"case {n_cases}: active counts differ from the dense positives")
    dv, di = torch.sort(torch.where(pos, dense, torch.full_like(dense, -1.0)), dim=1, descending=True, stable=True)
    slot = torch.arange(cap, device=dev)[None, :]
    live = slot < ca[:, None]
    if not torch.equal(torch.where(live, va, torch.zeros_like(va)), torch.where(live, dv[:, :cap], torch.zeros_like(va))):
        fail(f"case {n_cases}: emitted values are not the sorted dense positives")
    if not torch.equal(torch.where(live, fa.long(), torch.zeros_like(di[:, :cap])), torch.where(live, di[:, :cap], torch.zeros_like(di[:, :cap]))):
        fail(f"case {n_cases}: emitted features are not the sorted dense positives (ties by feature id)")
    acts = torch.where(pos, dense, torch.zeros_like(dense))
    ref_recon = acts.double() @ W_dec.double() + out_bias.double()
    tol = 1e-5 * float(ref_recon.abs().max()) + 1e-6
    if float((ra.double() - ref_recon).abs().max()) > 20 * tol:
        fail(f"case {n_cases}: reconstruction off by {float((ra.double() - ref_recon).abs().max()):.3e}")
    rows = W_dec[fa.clamp(min=0).long()]                                            # [M, cap, d_out]
    ref_scores = torch.where(live, va.double() * (rows.double() * dirs[:, None, :].double()).sum(-1), torch.zeros_like(va, dtype=torch.float64))
    stol = 1e-5 * float(ref_scores.abs().max()) + 1e-6
    if float((sa.double() - ref_scores).abs().max()) > 20 * stol:
        fail(f"case {n_cases}: scores off by {float((sa.double() - ref_scores).abs().max()):.3e}")
    # (2) the screened handle: same active set up to the rounding noise of two fp32 summation orders
    if d_in % 128 == 0 and m <= 32768:
        screen = tsc_hip.DeviceDict(W, b_enc, b_dec, W_dec=W_dec, out_bias=out_bias, compute_dtype=tsc_hip.F32_SCREEN)
        fb, vb, cb, rb, sb = screen.sparse_lens_pass(x, cap, dirs=dirs)
        if int(cb.max()) <= cap:
            # noise of a d_in-term fp32 dot product of these magnitudes
            mag = float(((x - b_dec).abs() @ W.abs()).max()) + float(b_enc.abs().max())
            noise = 4 * d_in * 2.0 ** -24 * mag
            got = torch.zeros_like(dense)
            got.scatter_add_(1, fb.clamp(min=0).long(), torch.where(slot < cb[:, None], vb, torch.zeros_like(vb)))  # padding adds 0 to column 0
            missed = (acts > noise) & (got == 0)
            invented = (got > noise) & (acts == 0)
            if bool(missed.any()) or bool(invented.any()):
                fail(f"case {n_cases}: screened pass missed {int(missed.sum())} / invented {int(invented.sum())} features "
                     f"(d_in {d_in} d_sae {d_sae} m {m} scale {scale} b_dec {bdec_scale}, noise {noise:.2e})")
            both = (acts > 0) & (got > 0)
            if float((acts - got)[both].abs().max() if bool(both.any()) else 0.0) > noise:
                fail(f"case {n_cases}: screened values differ by more than the rounding noise")
        del screen
    n_active += int(ca.sum())
    n_cases += 1
    del plain
print(f"ok: {n_cases} cases, {n_active} active (row, feature) pairs checked (seed {seed})")
