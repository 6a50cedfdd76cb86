#!/usr/bin/env python3
"""Randomised comparison of tsc_encode_topk_screened against tsc_encode_topk (finished lists must be bit-identical)
over shapes the fixed tests do not hit: odd d_sae, several d_in, k 1..64, 1..8 slots, ragged row counts, cold and warm
lists, duplicated rows (exact ties), scaled inputs.  Usage: fuzz_screened.py [seconds] [seed] [max cases]"""
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
t0, n_cases, n_calls = time.time(), 0, 0


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


while time.time() - t0 < budget and n_cases < max_cases:
    d_in = [128, 256, 384, 768, 1024][ri(0, 4)]
    d_sae = [132, 1536, 4100, 6144, 24576, 24580][ri(0, 5)]
    k = [1, 5, 20, 33, 64][ri(0, 4)]
    n_slots = [1, 2, 3, 4, 8][ri(0, 4)]
    if n_slots * k > 2048:
        continue
    scale = [1.0, 1e-3, 37.0][ri(0, 2)]
    gd = torch.Generator(device=dev).manual_seed(seed * 1000 + n_cases)
    W = (torch.randn(d_in, d_sae, generator=gd, device=dev) / d_in ** 0.5).contiguous()
    b_enc = torch.randn(d_sae, generator=gd, device=dev) * 0.3 - [0.0, 1.0, 2.0][ri(0, 2)]
    b_dec = torch.randn(d_in, generator=gd, device=dev) * [0.0, 0.1, 5.0][ri(0, 2)]
    plain = tsc_hip.DeviceDict(W, b_enc, b_dec)
    screen = tsc_hip.DeviceDict(W, b_enc, b_dec, compute_dtype=tsc_hip.F32_SCREEN)
    ref = tsc_hip.TopkState(d_sae, k, dev, n_slots=n_slots)
    st = tsc_hip.TopkState(d_sae, k, dev, n_slots=n_slots)
    base = 0
    for call in range(ri(2, 5)):
        m = [1, 7, 255, 1000, 4097, 20000, 40000][ri(0, 6)]
        if m * d_sae > 6e8:
            m = 4097
        x = torch.randn(m, d_in, generator=gd, device=dev) * scale + b_dec * [0.0, 1.0][ri(0, 1)]
        if m > 10 and ri(0, 1):
            x[m // 2:m // 2 + m // 4] = x[:m // 4]  # duplicated rows: exact ties across token ids
        plain.encode_topk(x, base, ref)
        if call == 0 and ri(0, 1):
            screen.encode_topk(x, base, st)       # warm on the fp32 kernel
        else:
            screen.encode_topk_screened(x, base, st)
        base += m
        n_calls += 1
    rv, ri_ = ref.finalize()
    v, i = st.finalize()
    if not (torch.equal(v, rv) and torch.equal(i, ri_)):
        print(f"MISMATCH case {n_cases}: d_in {d_in} d_sae {d_sae} k {k} slots {n_slots} scale {scale}: "
              f"{int((i != ri_).sum())} ids, {int((v != rv).sum())} values differ", flush=True)
        sys.exit(1)
    n_cases += 1
    del plain, screen, ref, st
print(f"ok: {n_cases} cases, {n_calls} calls, all lists bit-identical (seed {seed})")
