#!/usr/bin/env python3
"""Randomised check of the grouped lens-pass calls (tsc_sparse_lens_emit_many + tsc_sparse_lens_finish_many) against
tsc_sparse_lens_pass dictionary by dictionary: 2..6 dictionaries of mixed widths, ragged row counts, screened and plain
handles, with and without scores / reconstruction -- every output bit for bit.
Usage: fuzz_grouped.py [seconds] [seed] [max cases]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "transcoder-sae-circuits_amd"))
import torch  # noqa: E402

import tsc_hip  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_cases = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 60
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(seed)
t0, n_cases = time.time(), 0


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


while time.time() - t0 < budget and n_cases < max_cases:
    n = ri(2, 6)
    d_in = [128, 768][ri(0, 1)]
    d_out = [64, 768][ri(0, 1)]
    m = [1, 77, 300, 700, 2100][ri(0, 4)]
    screened = bool(ri(0, 1))
    want_scores, want_recon = bool(ri(0, 3)), bool(ri(0, 3))
    cap = 512
    gd = torch.Generator(device=dev).manual_seed(seed * 49979687 + n_cases)
    hs, xs, dirs = [], [], []
    for i in range(n):
        d_sae = [260, 1000, 1536, 4100, 24576][ri(0, 4)]
        W = (torch.randn(d_in, d_sae, generator=gd, device=dev) / d_in ** 0.5).contiguous()
        b_enc = -(torch.rand(d_sae, generator=gd, device=dev) + (2.0 if d_sae > 2000 else 1.2))
        b_dec = torch.randn(d_in, generator=gd, device=dev) * 0.1
        hs.append(tsc_hip.DeviceDict(W, b_enc, b_dec, W_dec=torch.randn(d_sae, d_out, generator=gd, device=dev),
                                     out_bias=torch.randn(d_out, generator=gd, device=dev),
                                     compute_dtype=tsc_hip.F32_SCREEN if screened else tsc_hip.F32))
        xs.append(torch.randn(m, d_in, generator=gd, device=dev))
        dirs.append(torch.randn(m, d_out, generator=gd, device=dev))
    ref = [h.sparse_lens_pass(x, cap, dirs=d if want_scores else None, want_recon=want_recon) for h, x, d in zip(hs, xs, dirs)]
    if max(int(r[2].max()) for r in ref) > cap:
        for h in hs:
            h.close()
        continue
    feat = torch.empty(n, m, cap, dtype=torch.int32, device=dev)
    val = torch.empty(n, m, cap, device=dev)
    count = torch.zeros(n, m, dtype=torch.int32, device=dev)
    recon = torch.empty(n, m, d_out, device=dev) if want_recon else None
    scores = torch.empty(n, m, cap, device=dev) if want_scores else None
    tsc_hip.sparse_lens_emit_many(hs, xs, feat, val, count)
    tsc_hip.sparse_lens_finish_many(hs, xs, feat, val, count, dirs=dirs if want_scores else None, recon=recon, scores=scores)
    for i, (f, v, c, r, s) in enumerate(ref):
        w = f.shape[1]  # the single-dictionary call clamps its capacity to d_sae; beyond it the grouped buffers hold padding
        ok = torch.equal(count[i], c) and torch.equal(feat[i][:, :w], f) and torch.equal(val[i][:, :w], v)
        ok = ok and bool((feat[i][:, w:] == -1).all()) and bool((val[i][:, w:] == 0).all())
        ok = ok and (not want_recon or torch.equal(recon[i], r))
        ok = ok and (not want_scores or (torch.equal(scores[i][:, :w], s) and bool((scores[i][:, w:] == 0).all())))
        if not ok:
            print(f"MISMATCH case {n_cases} dictionary {i}: n {n} d_in {d_in} d_out {d_out} m {m} screened {screened} "
                  f"d_sae {hs[i].d_sae} scores {want_scores} recon {want_recon}", flush=True)
            sys.exit(1)
    for h in hs:
        h.close()
    n_cases += 1
print(f"ok: {n_cases} cases, grouped == per dictionary bit for bit (seed {seed})")
