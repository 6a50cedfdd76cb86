#!/usr/bin/env python3
"""Randomised check of tsc_topk_merge (the all-gather merge of config 3 and tsc_topk_finalize's engine) against an
exact torch merge: n_lists 1..64, k 1..64 (n_lists * k <= 2048), odd d_sae, lists of every fill level (padding: value
0 / id -1), exact value ties across lists (token ids decide), int64 ids beyond 2^32.  Usage: fuzz_merge.py [seconds] [seed] [max cases]"""
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
t0, n_cases = time.time(), 0


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


while time.time() - t0 < budget and n_cases < max_cases:
    k = [1, 2, 5, 20, 33, 64][ri(0, 5)]
    n_lists = min([1, 2, 3, 8, 17, 64][ri(0, 5)], 2048 // k)
    d_sae = [1, 7, 260, 4100, 24576][ri(0, 4)]
    gd = torch.Generator(device=dev).manual_seed(seed * 104729 + n_cases)
    # a pool of (value, id) pairs per feature with coarse values (many exact ties) and distinct ids, dealt to the lists
    pool = n_lists * k
    vals = torch.randint(1, 40 if ri(0, 1) else 100000, (d_sae, pool), generator=gd, device=dev).float() * 0.25
    ids = torch.argsort(torch.rand(d_sae, pool, generator=gd, device=dev), dim=1) + (ri(0, 1) * (1 << 33))
    fill = torch.randint(0, k + 1, (n_lists, d_sae), generator=gd, device=dev)           # live entries per list
    lv = vals.view(d_sae, n_lists, k).permute(1, 0, 2).contiguous()
    li = ids.view(d_sae, n_lists, k).permute(1, 0, 2).contiguous()
    # each list sorted (value descending, id ascending) and padded after its fill level, as finalized lists are
    key = torch.argsort(li, dim=2, stable=True)
    lv, li = torch.gather(lv, 2, key), torch.gather(li, 2, key)
    key = torch.argsort(lv, dim=2, descending=True, stable=True)
    lv, li = torch.gather(lv, 2, key), torch.gather(li, 2, key)
    slot = torch.arange(k, device=dev)[None, None, :]
    live = slot < fill[:, :, None]
    lv = torch.where(live, lv, torch.zeros_like(lv)).contiguous()
    li = torch.where(live, li, torch.full_like(li, -1)).contiguous()
    ov, oi = tsc_hip.topk_merge(lv, li)
    # exact merge
    av = torch.where(live, lv, torch.full_like(lv, float("-inf"))).permute(1, 0, 2).reshape(d_sae, pool)
    ai = torch.where(live, li, torch.full_like(li, torch.iinfo(torch.int64).max)).permute(1, 0, 2).reshape(d_sae, pool)
    key = torch.argsort(ai, dim=1, stable=True)
    av, ai = torch.gather(av, 1, key), torch.gather(ai, 1, key)
    key = torch.argsort(av, dim=1, descending=True, stable=True)
    av, ai = torch.gather(av, 1, key)[:, :k], torch.gather(ai, 1, key)[:, :k]
    ok = torch.isfinite(av)
    ev, ei = torch.where(ok, av, torch.zeros_like(av)), torch.where(ok, ai, torch.full_like(ai, -1))
    if not (torch.equal(ov, ev) and torch.equal(oi, ei)):
        print(f"MISMATCH case {n_cases}: n_lists {n_lists} k {k} d_sae {d_sae}: {int((oi != ei).sum())} ids, "
              f"{int((ov != ev).sum())} values differ", flush=True)
        sys.exit(1)
    n_cases += 1
print(f"ok: {n_cases} cases, every merged list equals the exact merge (seed {seed})")
