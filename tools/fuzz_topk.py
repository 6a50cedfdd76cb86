#!/usr/bin/env python3
"""Randomised check of tsc_encode_topk (fp32 and bf16 handles) against the exact ranking of the library's OWN dense
activations (value descending, token id ascending, positives only) on shapes the fixed tests do not hit: every staging
variant (d_in % 64 == 0: LDS-DMA; other multiples of 4: register staging, k tails), odd d_sae, k 1..64, 1..8 slots,
ragged row counts over several calls, duplicated rows.  Usage: fuzz_topk.py [seconds] [seed] [max cases]"""
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
t0, n_cases = time.time(), 0


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


def rank(acts, k, base):
    """[d_sae, k] exact lists of a dense [M, d_sae] matrix, on the device."""
    m = acts.shape[0]
    a = torch.where(acts > 0, acts, torch.full_like(acts, float("-inf"))).t().contiguous()  # [F, M]
    v, i = torch.sort(a, dim=1, descending=True, stable=True)  # stable: ties keep ascending token order
    v, i = v[:, :k], i[:, :k]
    if v.shape[1] < k:
        pad = k - v.shape[1]
        v = torch.cat([v, torch.full((v.shape[0], pad), float("-inf"), device=dev)], 1)
        i = torch.cat([i, torch.zeros(i.shape[0], pad, dtype=i.dtype, device=dev)], 1)
    live = torch.isfinite(v)
    return torch.where(live, v, torch.zeros_like(v)), torch.where(live, i + base, torch.full_like(i, -1))


while time.time() - t0 < budget and n_cases < max_cases:
    d_in = [4, 8, 28, 36, 100, 128, 192, 256, 768, 772, 1024][ri(0, 10)]
    d_sae = [4, 8, 132, 260, 1536, 4100, 6144, 24576][ri(0, 7)]
    k = [1, 5, 20, 33, 64][ri(0, 4)]
    n_slots = [1, 2, 3, 4, 8][ri(0, 4)]
    bf16 = d_in % 128 == 0 and ri(0, 2) == 0
    gd = torch.Generator(device=dev).manual_seed(seed * 1000 + n_cases)
    W = (torch.randn(d_in, d_sae, generator=gd, device=dev) / d_in ** 0.5).contiguous()
    b_enc = torch.randn(d_sae, generator=gd, device=dev) * 0.3 - [0.0, 1.0, 2.0][ri(0, 2)]
    b_dec = torch.randn(d_in, generator=gd, device=dev) * [0.0, 0.1][ri(0, 1)]
    d = tsc_hip.DeviceDict(W, b_enc, b_dec, compute_dtype=tsc_hip.BF16 if bf16 else tsc_hip.F32)
    st = tsc_hip.TopkState(d_sae, k, dev, n_slots=n_slots)
    xs = []
    for call in range(ri(1, 4)):
        m = [1, 7, 255, 1000, 4097, 9000][ri(0, 5)]
        if m * d_sae > 6e7:
            m = 1000
        x = torch.randn(m, d_in, generator=gd, device=dev)
        if m > 10 and ri(0, 1):
            x[m // 2:m // 2 + m // 4] = x[:m // 4]
        xs.append(x)
    base0 = ri(0, 1000)
    base = base0
    for x in xs:
        d.encode_topk(x, base, st)
        base += x.shape[0]
    # one dense call over everything: the fused kernel's value of an element does not depend on the launch shape
    ev, ei = rank(d.encode_dense(torch.cat(xs)), k, base0)
    v, i = st.finalize()
    if not (torch.equal(v, ev) and torch.equal(i, ei)):
        print(f"MISMATCH case {n_cases}: d_in {d_in} d_sae {d_sae} k {k} slots {n_slots} bf16 {bf16} rows {[x.shape[0] for x in xs]}: "
              f"{int((i != ei).sum())} ids, {int((v != ev).sum())} values differ", flush=True)
        sys.exit(1)
    n_cases += 1
print(f"ok: {n_cases} cases, all lists equal the exact ranking of the library's own dense activations (seed {seed})")
