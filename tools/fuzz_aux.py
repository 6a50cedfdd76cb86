#!/usr/bin/env python3
"""Randomised checks of the small entry points against fp64 torch restatements: tsc_project_features /
FeatureProjector (both routes), tsc_decode_rows (padding ids -1 and out-of-range ids decode to zero),
tsc_decode_sparse, tsc_contrib_scores (1..8 directions) and tsc_lens_assemble (random layouts).
Usage: fuzz_aux.py [seconds] [seed] [max cases]"""
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


def fail(msg):
    print(f"MISMATCH case {n_cases}: {msg}", flush=True)
    sys.exit(1)


def close(a, b, what, rtol=2e-5, scale=None):
    # scale: the magnitude of the summed terms (a sum that cancels is only accurate relative to those)
    scale = (float(b.abs().max()) if scale is None else float(scale)) + 1e-30
    err = float((a.double() - b.double()).abs().max())
    if err > rtol * scale + 1e-7:
        fail(f"{what}: error {err:.3e} at scale {scale:.3e}")


while time.time() - t0 < budget and n_cases < max_cases:
    gd = torch.Generator(device=dev).manual_seed(seed * 32452843 + n_cases)
    kind = ri(0, 3)
    if kind == 0:  # selected-feature projection
        d_in = [4, 36, 64, 128, 768, 1028][ri(0, 5)]
        F = [1, 3, 5, 6, 7, 64, 65, 130][ri(0, 7)]
        m = [1, 3, 1000, 1024, 3000][ri(0, 4)]
        Pt = torch.randn(F, d_in, generator=gd, device=dev) / d_in ** 0.5
        bias = torch.randn(F, generator=gd, device=dev)
        b_dec = torch.randn(d_in, generator=gd, device=dev) if ri(0, 1) else None
        x = torch.randn(m, d_in, generator=gd, device=dev)
        relu = bool(ri(0, 1))
        ref = (x.double() - (b_dec.double() if b_dec is not None else 0)) @ Pt.double().t() + bias.double()
        if relu:
            ref = ref.clamp(min=0)
        mag = (x.double().abs() + (b_dec.double().abs() if b_dec is not None else 0)) @ Pt.double().abs().t() + bias.double().abs()
        out = tsc_hip.project_features(x, Pt, bias, b_dec, relu=relu)
        if tuple(out.shape) != (m, F) or float(((out.double() - ref).abs() / mag).max()) > 1e-5:
            fail(f"project_features d_in {d_in} F {F} m {m}: {float(((out.double() - ref).abs() / mag).max()):.3e}")
    elif kind in (1, 2):  # decoder-side gathers
        d_sae = [4, 260, 4100, 24576][ri(0, 3)]
        d_out = [4, 8, 64, 768, 1024, 4096][ri(0, 5)]
        if d_sae * d_out > 3e7:
            d_out = 768
        m = [1, 5, 300, 2500][ri(0, 3)]
        cap = min([1, 7, 64, 300][ri(0, 3)], d_sae)
        W_enc = torch.zeros(4, d_sae, device=dev)
        W_dec = torch.randn(d_sae, d_out, generator=gd, device=dev)
        ob = torch.randn(d_out, generator=gd, device=dev) if ri(0, 1) else None
        d = tsc_hip.DeviceDict(W_enc, torch.zeros(d_sae, device=dev), torch.zeros(4, device=dev), W_dec=W_dec, out_bias=ob)
        count = torch.randint(0, cap + 1, (m,), generator=gd, device=dev, dtype=torch.int32)
        feat = torch.randint(0, d_sae, (m, cap), generator=gd, device=dev, dtype=torch.int32)
        val = torch.rand(m, cap, generator=gd, device=dev) + 0.1
        live = torch.arange(cap, device=dev)[None, :] < count[:, None]
        feat = torch.where(live, feat, torch.full_like(feat, -1))
        val = torch.where(live, val, torch.zeros_like(val))
        rows = W_dec[feat.clamp(min=0).long()].double() * val.double()[..., None]          # [m, cap, d_out], 0 on padding
        if kind == 1:
            bad = feat.clone().reshape(-1)
            if bad.numel() > 3:
                bad[1] = d_sae + 5      # out of range: decodes to zero, like the padding id
            exp = torch.where(((bad >= 0) & (bad < d_sae))[:, None], W_dec[bad.clamp(0, d_sae - 1).long()].double() * val.reshape(-1).double()[:, None], torch.zeros(1, dtype=torch.float64, device=dev))
            close(d.decode_rows(bad, val.reshape(-1)), exp, f"decode_rows d_sae {d_sae} d_out {d_out}")
            if d_out <= 4096:
                close(d.decode_sparse(feat, val, count), rows.sum(1) + (ob.double() if ob is not None else 0), f"decode_sparse d_out {d_out} cap {cap}",
                      scale=(rows.abs().sum(1) + (ob.double().abs() if ob is not None else 0)).max())
        else:
            n_dir = ri(1, 8)
            if n_dir * d_out * 4 > 60 * 1024:
                n_dir = 1
            dirs = torch.randn(n_dir, d_out, generator=gd, device=dev)
            exp = torch.einsum("mcd,nd->mcn", rows, dirs.double())
            mag = torch.einsum("mcd,nd->mcn", rows.abs(), dirs.double().abs()).max()
            close(d.contrib_scores(feat, val, count, dirs), exp, f"contrib_scores n_dir {n_dir} d_out {d_out}", rtol=1e-5, scale=mag)
        d.close()
    else:  # lens_assemble on a random layout
        G = ri(1, 24)
        cap = [1, 5, 64, 512][ri(0, 3)]
        R = [1, 3, 40][ri(0, 2)]
        n_head = ri(0, 3)
        group_aux = [ri(0, 4) for _ in range(G)]
        k = min([1, 4, 10, 64][ri(0, 3)], 256)
        scores = torch.randn(G, R, cap, generator=gd, device=dev)
        count = torch.randint(0, cap + 2, (G, R), generator=gd, device=dev, dtype=torch.int32)
        head = torch.randn(R, n_head, generator=gd, device=dev) + 25.0
        aux = torch.randn(R, sum(group_aux), generator=gd, device=dev)
        out = tsc_hip.lens_assemble(head, aux, group_aux, scores, count, k=k)
        parts, masks, off = [head], [torch.ones_like(head, dtype=torch.bool)], 0
        slot = torch.arange(cap, device=dev)
        for gi, a in enumerate(group_aux):
            parts.append(aux[:, off:off + a]); masks.append(torch.ones(R, a, dtype=torch.bool, device=dev)); off += a
            lv = slot[None, :] < count[gi].clamp(max=cap)[:, None]
            parts.append(torch.where(lv, scores[gi], torch.zeros_like(scores[gi]))); masks.append(lv)
        raw, mask = torch.cat(parts, 1), torch.cat(masks, 1)
        if not (torch.equal(out["raw"], raw) and torch.equal(out["mask"], mask)):
            fail("lens_assemble raw / mask")
        total = raw.double().sum(1, keepdim=True)
        if float(total.abs().min()) < 1e-3:
            n_cases += 1
            continue
        acts = (raw.double() / total).float()
        cond = float((raw.double().abs().sum(1, keepdim=True) / total.abs()).max())  # cancellation in the fp32 row sum
        close(out["activations"], torch.where(mask, acts, torch.zeros_like(acts)), "lens_assemble activations",
              rtol=2e-5 + 4e-7 * cond)
        # ranking of the kernel's OWN activations (the row sum's rounding can merge or split near-ties)
        own = torch.where(mask, out["activations"], torch.full_like(acts, float("-inf")))
        v, c = torch.sort(own, dim=1, descending=True, stable=True)
        v, c = v[:, :k], c[:, :k]
        if v.shape[1] < k:
            v = torch.cat([v, torch.full((R, k - v.shape[1]), float("-inf"), device=dev)], 1)
            c = torch.cat([c, torch.zeros(R, k - c.shape[1], dtype=c.dtype, device=dev)], 1)
        c = torch.where(torch.isfinite(v), c, torch.full_like(c, -1))
        if not (torch.equal(out["top_columns"].long(), c) and torch.equal(out["top_values"], v)):
            fail(f"lens_assemble top-k (G {G} cap {cap} R {R} k {k} aux {group_aux})")
    n_cases += 1
print(f"ok: {n_cases} cases (seed {seed})")
