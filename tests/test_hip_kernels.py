"""GPU parity of the HIP kernels (through the C-ABI, via ``tsc_hip``) against the CPU oracle and
the golden vectors.  Tolerance for floating point: rtol 1e-4 (north star) with a small absolute
floor for values that are sums of cancelling terms; indices bit-exact."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import encode_ref as E
from oracle import lens_ref as L
from oracle import topk_ref as T

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 2e-5
DEV = "cuda:0"


def _dd(w, **kw):
    import tsc_hip

    g = {k: v.to(DEV) for k, v in w.items()}
    return tsc_hip.DeviceDict(g["W_enc"], g["b_enc"], g["b_dec"], g["W_dec"], g.get("b_dec_out", g["b_dec"]), **kw), g


def _close(a, b, rtol=RTOL, atol=ATOL):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b.detach().cpu() if torch.is_tensor(b) else b), dtype=torch.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.allclose(a, b, rtol=rtol, atol=atol), float((a - b).abs().max())


def _assert_topk_equal(vals, idx, ref_vals, ref_idx, ref_acts=None, base=0):
    """Values within tolerance at every rank; indices bit-exact wherever both sides rank the same
    values (north star).  The MFMA sums k in a different order than the CPU sgemm, so two tokens
    whose oracle activations differ by less than the tolerance may swap ranks (or swap across the
    list's end): a differing index is accepted only if the oracle's own activation of the token the
    GPU chose is within tolerance of the oracle's value at that rank (``ref_acts [tokens, d_sae]``)."""
    vals, idx = vals.cpu(), idx.cpu()
    _close(vals, ref_vals)
    same = idx == ref_idx
    if bool(same.all()):
        return 0
    assert ref_acts is not None, f"{int((~same).sum())} indices differ and no dense oracle activations were given"
    bad_f, bad_j = torch.nonzero(~same, as_tuple=True)
    for f, j in zip(bad_f.tolist(), bad_j.tolist()):
        tok = int(idx[f, j])
        assert tok >= 0, f"feature {f} rank {j}: GPU list ends early"
        v_ref_rank = float(ref_vals[f, j])
        v_ref_tok = float(ref_acts[tok - base, f])
        assert abs(v_ref_tok - v_ref_rank) <= RTOL * abs(v_ref_rank) + ATOL, \
            f"feature {f} rank {j}: token {tok} (oracle act {v_ref_tok}) vs oracle rank value {v_ref_rank}"
    # near-tie swaps must stay rare
    assert int((~same).sum()) <= max(4, same.numel() // 200), int((~same).sum())
    return int((~same).sum())


def test_encode_dense_matches_golden_and_oracle():
    g, wt, wz, x = gi.g1_inputs()
    dt, _ = _dd(wt)
    dz, _ = _dd(wz)
    xg = x.to(DEV)
    _close(dt.encode_dense(xg), g["tc_encode"])
    _close(dz.encode_dense(xg), g["zs_encode"])
    _close(dt.encode_dense(xg.view(4, 16, 768)), g["tc_encode_3d"])
    full = gi.g1_full_weights()
    df, _ = _dd(full)
    _close(df.encode_dense(xg[:4].contiguous()), g["full_encode"])


@pytest.mark.parametrize("m", [1, 127, 128, 129, 1000])
def test_encode_dense_ragged_rows(m):
    _, wt, _, _ = gi.g1_inputs()
    dt, _ = _dd(wt)
    x = torch.randn(m, 768, generator=torch.Generator().manual_seed(m))
    _close(dt.encode_dense(x.to(DEV)), E.encode_ref(x, wt["W_enc"], wt["b_enc"], wt["b_dec"]))


def test_encode_dense_odd_shapes():
    # d_in not a multiple of the K step, d_sae not a multiple of the column tile
    from standins.synthetic_dicts import make_dictionary_weights

    w = make_dictionary_weights(d_in=100, d_sae=200, seed=5, b_enc=-0.1)
    d, _ = _dd(w)
    x = torch.randn(77, 100, generator=torch.Generator().manual_seed(9))
    _close(d.encode_dense(x.to(DEV)), E.encode_ref(x, w["W_enc"], w["b_enc"], w["b_dec"]))
    assert d.encode_dense(torch.empty(0, 100, device=DEV)).shape == (0, 200)


@pytest.mark.parametrize("d_in,d_sae,bf16", [(64, 1000, False), (768, 1000, False), (192, 132, False),
                                             (128, 200, True), (768, 1000, True), (96, 260, False),
                                             (4, 4, False), (8, 132, False), (28, 260, False), (36, 8, False)])
def test_all_epilogues_on_ragged_dictionary_shapes(d_in, d_sae, bf16):
    """Shapes that exercise the tails of every kernel variant: d_sae not a multiple of the 128-wide
    column tile (clamped W rows, masked columns), the shortest k-loops of the LDS-DMA kernels
    (d_in = 64 / 128), a d_in that falls back to register staging (96), ragged token tiles, and the smallest
    legal shapes (d_in < 32: a whole k-step of clamped x columns -- a 32-bit offset once went negative there and
    the load faulted; d_sae = 4: one partial column tile)."""
    import tsc_hip
    from standins.synthetic_dicts import make_dictionary_weights

    w = make_dictionary_weights(d_in=d_in, d_sae=d_sae, seed=d_in + d_sae, b_enc=-0.2)
    d, g = _dd(w, compute_dtype=tsc_hip.BF16 if bf16 else tsc_hip.F32)
    m = 391
    x = torch.randn(m, d_in, generator=torch.Generator().manual_seed(d_sae))
    x[300:330] = x[5:35]
    ref = (E.encode_bf16_ref if bf16 else E.encode_ref)(x, w["W_enc"], w["b_enc"], w["b_dec"])
    xg = x.to(DEV)
    dense = d.encode_dense(xg)
    _close(dense, ref)
    k = 9
    exp_v, exp_i = _rank_own_dense(dense, k, base=3)
    st = tsc_hip.TopkState(d_sae, k, DEV, n_slots=3)
    d.encode_topk(xg[:129], 3, st)
    d.encode_topk(xg[129:], 3 + 129, st)
    v, i = st.finalize()
    assert torch.equal(v.cpu(), exp_v) and torch.equal(i.cpu(), exp_i)
    feat, val, count = d.encode_sparse(xg, 64)
    assert torch.equal(count.long(), (dense > 0).sum(dim=1))
    row = 390
    c = int(count[row])
    order = torch.sort(dense[row], descending=True, stable=True)
    assert torch.equal(val[row, :c], order.values[:c]) and torch.equal(feat[row, :c].long(), order.indices[:c])


_VARIANTS = [
    {"f32_tile": 128}, {"f32_tile": 256}, {"f32_tile": 128, "lists": 1}, {"f32_tile": 128, "staging": 1},
    {"f32_tile": 256, "staging": 1}, {"bf16_tile": 128}, {"bf16_tile": 256}, {"bf16_tile": 128, "lists": 1},
]


@pytest.mark.parametrize("variant", _VARIANTS, ids=lambda v: ",".join(f"{k}={x}" for k, x in v.items()))
def test_every_kernel_variant_gives_the_same_results(variant):
    """Tile geometry, staging mode and list placement are speed choices: forced one at a time
    (``tsc_set_tuning``), each must reproduce the oracle (dense) and the exact ranking of its own dense
    activations (fused top-k, sparse emit), on a ragged shape."""
    import tsc_hip
    from standins.synthetic_dicts import make_dictionary_weights

    bf16 = any(k.startswith("bf16") for k in variant)
    w = make_dictionary_weights(d_in=768, d_sae=1000, seed=77, b_enc=-0.9)
    m = 700
    x = torch.randn(m, 768, generator=torch.Generator().manual_seed(78))
    x[500:560] = x[20:80]
    ref = (E.encode_bf16_ref if bf16 else E.encode_ref)(x, w["W_enc"], w["b_enc"], w["b_dec"])
    try:
        for key, value in variant.items():
            tsc_hip.set_tuning(key, value)
        d, _ = _dd(w, compute_dtype=tsc_hip.BF16 if bf16 else tsc_hip.F32)
        xg = x.to(DEV)
        dense = d.encode_dense(xg)
        _close(dense, ref)
        for k, n_slots, cuts in ((20, 2, [0, 300, m]), (33, 3, [0, m])):
            exp_v, exp_i = _rank_own_dense(dense, k, base=11)
            st = tsc_hip.TopkState(1000, k, DEV, n_slots=n_slots)
            for a, b in zip(cuts[:-1], cuts[1:]):
                d.encode_topk(xg[a:b], 11 + a, st)
            v, i = st.finalize()
            assert torch.equal(v.cpu(), exp_v) and torch.equal(i.cpu(), exp_i)
        feat, val, count = d.encode_sparse(xg, 128)
        assert torch.equal(count.long(), (dense > 0).sum(dim=1))
        for row in (0, 255, 256, m - 1):
            c = int(count[row])
            order = torch.sort(dense[row], descending=True, stable=True)
            assert torch.equal(val[row, :c], order.values[:c]) and torch.equal(feat[row, :c].long(), order.indices[:c])
    finally:
        for key in ("f32_tile", "bf16_tile", "staging", "lists"):
            tsc_hip.set_tuning(key, 0)


def test_folded_decoder_bias_under_cancellation():
    """The LDS-DMA kernels contract the raw x and add b_enc - b_dec @ W_enc instead of subtracting b_dec
    first.  Worst case for that: activations that sit close to a large b_dec, as in
    a trained dictionary whose b_dec is the mean activation -- x @ W_enc and b_dec @ W_enc then cancel to
    two digits.  The result must still meet the fp32 gate against the reference order of operations."""
    from standins.synthetic_dicts import make_dictionary_weights

    w = make_dictionary_weights(768, 1536, seed=31, b_enc=0.0)
    gen = torch.Generator().manual_seed(32)
    w["b_dec"] = 3.0 * torch.randn(768, generator=gen)
    x = w["b_dec"] + 0.05 * torch.randn(600, 768, generator=gen)
    ref = E.encode_ref(x, w["W_enc"], w["b_enc"], w["b_dec"])
    assert float(ref.abs().max()) < 0.1 * float((x @ w["W_enc"]).abs().max())  # the cancellation is real
    d, _ = _dd(w)
    out = d.encode_dense(x.to(DEV)).cpu()
    scale = float(ref.abs().max())
    assert float((out - ref).abs().max()) <= 1e-4 * scale, float((out - ref).abs().max()) / scale


def test_fused_topk_matches_golden():
    import tsc_hip

    g, w, xs, n_seq, n_pos = gi.g2_inputs()
    d, _ = _dd(w)
    ref_vals = torch.from_numpy(g["vals"])
    ref_idx = torch.from_numpy(g["idx"].astype(np.int64))
    xg = xs.to(DEV)
    acts = E.encode_ref(xs, w["W_enc"], w["b_enc"], w["b_dec"])
    for n_slots, batch in ((None, 8 * n_pos), (1, 1000), (3, 4096), (16, 333)):
        st = tsc_hip.TopkState(1536, 20, DEV, n_slots=n_slots)
        for s in range(0, xg.shape[0], batch):
            d.encode_topk(xg[s:s + batch], s, st)
        vals, idx = st.finalize()
        _assert_topk_equal(vals, idx, ref_vals, ref_idx, acts)
        # padding of short lists
        short = torch.from_numpy(g["n_active"] < 20)
        assert bool((idx.cpu()[short][:, -1] == -1).all()) and bool((vals.cpu()[short][:, -1] == 0).all())


def test_fused_topk_descending_stream_forces_slow_path():
    """Every token beats the running k-th best (activations grow with the token id), so the
    candidate queue overflows and the multi-round drain is exercised."""
    import tsc_hip
    from standins.synthetic_dicts import make_dictionary_weights

    w = make_dictionary_weights(768, 256, seed=3, b_enc=0.0)
    w["W_enc"] = w["W_enc"].abs()
    w["b_dec"].zero_()
    m = 1024
    x = (torch.rand(m, 768, generator=torch.Generator().manual_seed(1)) + 0.5) * torch.linspace(1.0, 3.0, m)[:, None]
    d, _ = _dd(w)
    st = tsc_hip.TopkState(256, 8, DEV, n_slots=2)
    d.encode_topk(x.to(DEV), 1000, st)
    vals, idx = st.finalize()
    rv, ri = T.topk_stream_ref(x, w, k=8, token_id_base=1000)
    _assert_topk_equal(vals, idx, rv, ri, E.encode_ref(x, w["W_enc"], w["b_enc"], w["b_dec"]), base=1000)
    assert len(set(idx.cpu()[0].tolist())) == 8  # no token listed twice


def test_fused_topk_exact_ties_prefer_smaller_token_id():
    import tsc_hip

    _, wt, _, _ = gi.g1_inputs()
    d, _ = _dd(wt)
    row = torch.randn(1, 768, generator=torch.Generator().manual_seed(4))
    x = row.repeat(300, 1)  # 300 identical tokens: every activation ties
    st = tsc_hip.TopkState(1536, 5, DEV, n_slots=4)
    d.encode_topk(x[:200].to(DEV), 0, st)
    d.encode_topk(x[200:].to(DEV), 200, st)
    vals, idx = st.finalize()
    acts = E.encode_ref(row, wt["W_enc"], wt["b_enc"], wt["b_dec"])[0]
    live = acts > 0
    expect = torch.arange(5).expand(1536, 5)
    assert torch.equal(idx.cpu()[live], expect[live])
    assert bool((idx.cpu()[~live] == -1).all())


def _rank_own_dense(acts, k, base=0):
    """Top-k per feature of a dense activation matrix under the reference's ranking (value
    descending, token id ascending, positives only): the expected lists when the values are the
    GPU's own, so the comparison is bit-exact and isolates the list logic from sgemm rounding."""
    a = acts.cpu().numpy()
    m, f = a.shape
    vals = np.zeros((f, k), np.float32)
    idx = np.full((f, k), -1, np.int64)
    ids = np.arange(m)
    for j in range(f):
        col = a[:, j]
        order = np.lexsort((ids, -col))[:k]
        order = order[col[order] > 0]
        vals[j, :len(order)] = col[order]
        idx[j, :len(order)] = order + base
    return torch.from_numpy(vals), torch.from_numpy(idx)


@pytest.mark.parametrize("k,n_slots,batch", [(20, 8, 1024), (20, 1, 700), (7, 3, 333), (33, 2, 4096)])
def test_fused_topk_bit_exact_on_duplicated_tokens(k, n_slots, batch):
    """Real token streams repeat contexts (every sequence starts with the same BOS activation), so
    exact value ties are common: blocks of tokens are repeated here, in the middle of the stream
    too, and the lists must equal the ranking of the GPU's own dense activations bit for bit
    (k = 33 takes the kernel variant whose lists are not mirrored in LDS)."""
    import tsc_hip

    g, w, xs, n_seq, n_pos = gi.g2_inputs()
    gen = torch.Generator().manual_seed(11)
    a, b, c = xs[:900], xs[900:1500], xs[1500:2300]
    x = torch.cat([a, b, a[:500], c, b, a[200:700], c[:50].repeat(6, 1)])
    x = x[torch.randperm(x.shape[0], generator=gen)].contiguous()
    d, _ = _dd(w)
    xg = x.to(DEV)
    exp_v, exp_i = _rank_own_dense(d.encode_dense(xg), k, base=50)
    for floor in (True, False):  # with and without the optional cross-slot floor scratch
        st = tsc_hip.TopkState(1536, k, DEV, n_slots=n_slots, cross_slot_floor=floor)
        for s in range(0, xg.shape[0], batch):
            d.encode_topk(xg[s:s + batch], 50 + s, st)
        vals, idx = st.finalize()
        assert torch.equal(vals.cpu(), exp_v)
        assert torch.equal(idx.cpu(), exp_i)


BF16_RTOL = 2.0 ** -7  # relative to the fp32 restatement on bf16-rounded operands


def test_bf16_encode_dense_matches_restatement():
    import tsc_hip

    g, wt, wz, x = gi.g1_inputs()
    for w in (wt, wz):
        d, _ = _dd(w, compute_dtype=tsc_hip.BF16)
        for m in (1, 64, 200):
            xm = x.repeat(4, 1)[:m].contiguous()
            ref = E.encode_bf16_ref(xm, w["W_enc"], w["b_enc"], w["b_dec"])
            out = d.encode_dense(xm.to(DEV)).cpu()
            # products of bf16 numbers are exact in fp32: only the summation order differs
            assert torch.allclose(out, ref, rtol=BF16_RTOL, atol=1e-5), float((out - ref).abs().max())
            assert torch.allclose(out, ref, rtol=1e-4, atol=2e-5), float((out - ref).abs().max())
        # and the mode stays within bf16 precision of the reference's fp32 encode
        full = E.encode_ref(x, w["W_enc"], w["b_enc"], w["b_dec"])
        out = d.encode_dense(x.to(DEV)).cpu()
        assert float((out - full).abs().max()) <= 0.05 * float(full.abs().max())


@pytest.mark.parametrize("k,n_slots,batch", [(20, 4, 1024), (33, 1, 5000)])
def test_bf16_fused_topk_bit_exact_against_own_dense(k, n_slots, batch):
    import tsc_hip

    g, w, xs, n_seq, n_pos = gi.g2_inputs()
    x = torch.cat([xs, xs[:700]])  # with exact ties
    d, _ = _dd(w, compute_dtype=tsc_hip.BF16)
    xg = x.to(DEV)
    exp_v, exp_i = _rank_own_dense(d.encode_dense(xg), k)
    st = tsc_hip.TopkState(1536, k, DEV, n_slots=n_slots)
    for s in range(0, xg.shape[0], batch):
        d.encode_topk(xg[s:s + batch], s, st)
    vals, idx = st.finalize()
    assert torch.equal(vals.cpu(), exp_v)
    assert torch.equal(idx.cpu(), exp_i)


def test_bf16_wide_dictionary_config5_shape():
    """d_sae = 131072 (BASELINE config 5), chunked input (more rows than the bf16 workspace holds)."""
    import tsc_hip
    from standins.synthetic_dicts import make_dictionary_weights

    w = make_dictionary_weights(768, 131072, seed=5, b_enc=-0.35)
    d, _ = _dd(w, compute_dtype=tsc_hip.BF16)
    m = 32768 + 1000
    x = torch.randn(m, 768, generator=torch.Generator().manual_seed(2))
    st = tsc_hip.TopkState(131072, 20, DEV)
    d.encode_topk(x.to(DEV), 7, st)
    vals, idx = st.finalize()
    # oracle on a slice of the features (the full CPU sgemm would take minutes)
    sl = slice(65536 - 512, 65536 + 512)
    ws = {"W_enc": w["W_enc"][:, sl].contiguous(), "b_enc": w["b_enc"][sl].contiguous(), "b_dec": w["b_dec"]}
    acts = E.encode_bf16_ref(x, ws["W_enc"], ws["b_enc"], ws["b_dec"])
    rv, ri = _rank_own_dense(acts, 20, base=7)
    v, i = vals.cpu()[sl], idx.cpu()[sl]
    assert torch.allclose(v, rv, rtol=BF16_RTOL, atol=1e-5)
    same = i == ri
    if not bool(same.all()):  # near-ties may swap; the GPU's token must have (almost) the rank's value
        for f, j in zip(*torch.nonzero(~same, as_tuple=True)):
            tok = int(i[f, j]) - 7
            assert tok >= 0 and abs(float(acts[tok, f]) - float(rv[f, j])) <= 1e-4 * abs(float(rv[f, j])) + 2e-5
        assert int((~same).sum()) <= same.numel() // 200


def test_fused_topk_many_calls_with_kept_floor():
    """More calls than the floor warm-up: most of them reuse an older cross-slot floor
    (TSC_TOPK_FLOOR_KEEP); the result must still be the exact ranking, also after a resume."""
    import tsc_hip

    g, w, xs, n_seq, n_pos = gi.g2_inputs()
    d, _ = _dd(w)
    x = torch.cat([xs, xs.flip(0) * 1.01, xs[::2] * 0.99])
    xg = x.to(DEV)
    exp_v, exp_i = _rank_own_dense(d.encode_dense(xg), 20)
    st = tsc_hip.TopkState(1536, 20, DEV, n_slots=4)
    batch = 256  # 40 calls
    half = (xg.shape[0] // 2 // batch) * batch
    for s in range(0, half, batch):
        d.encode_topk(xg[s:s + batch], s, st)
    assert st._calls > st.FLOOR_WARM_CALLS and st._struct.flags in (0, 1)
    v, i = st.finalize()
    st2 = tsc_hip.TopkState(1536, 20, DEV, n_slots=4)
    st2.load(v, i)  # resume: the floor starts over
    assert st2._calls == 0
    for s in range(half, xg.shape[0], batch):
        d.encode_topk(xg[s:s + batch], s, st2)
    v, i = st2.finalize()
    assert torch.equal(v.cpu(), exp_v) and torch.equal(i.cpu(), exp_i)


@pytest.mark.parametrize("n_feat", [1, 5, 7, 64, 70])
def test_feature_projector_matches_oracle(n_feat):
    """Selected encoder columns over a long token stream: the skinny-GEMM route (>= 6 columns, padded to a
    multiple of 4) and the per-row kernel must both give raw_feature_scores_ref (autointerpretability.py:
    226-238) and, with the ReLU, the columns of the full encode."""
    import tsc_hip

    g, w, xs, n_seq, n_pos = gi.g2_inputs()
    feats = torch.arange(3, 3 + 19 * n_feat, 19) % 1536
    Pt = w["W_enc"][:, feats].t().contiguous().to(DEV)
    bias = w["b_enc"][feats].contiguous().to(DEV)
    proj = tsc_hip.FeatureProjector(Pt, bias, w["b_dec"].to(DEV))
    x = xs[:3000]
    raw = E.raw_feature_scores_ref(x, w, feats)
    for rows in (3000, 100):  # long input: GEMM route when eligible; short: per-row kernel
        _close(proj(x[:rows].to(DEV), relu=False), raw[:rows])
        _close(proj(x[:rows].to(DEV), relu=True), torch.clamp_min(raw[:rows], 0))
    proj.close()


def test_topk_merge_and_resume():
    import tsc_hip

    g, w, xs, n_seq, n_pos = gi.g2_inputs()
    d, _ = _dd(w)
    half = xs.shape[0] // 2
    parts = []
    for base, chunk in ((0, xs[:half]), (half, xs[half:])):
        st = tsc_hip.TopkState(1536, 20, DEV)
        d.encode_topk(chunk.to(DEV), base, st)
        parts.append(st.finalize())
    mv, mi = tsc_hip.topk_merge(torch.stack([parts[1][0], parts[0][0]]), torch.stack([parts[1][1], parts[0][1]]))
    acts = E.encode_ref(xs, w["W_enc"], w["b_enc"], w["b_dec"])
    _assert_topk_equal(mv, mi, torch.from_numpy(g["vals"]), torch.from_numpy(g["idx"].astype(np.int64)), acts)
    # resume: load the first half's lists, stream the second half
    st = tsc_hip.TopkState(1536, 20, DEV)
    st.load(*parts[0])
    d.encode_topk(xs[half:].to(DEV), half, st)
    rv, ri = st.finalize()
    _assert_topk_equal(rv, ri, torch.from_numpy(g["vals"]), torch.from_numpy(g["idx"].astype(np.int64)), acts)


def test_sparse_emit_decode_and_contrib():
    g, wt, wz, x = gi.g1_inputs()
    w = gi.g1_full_weights()
    d, gw = _dd(w)
    xg = x.to(DEV)
    feat, val, count = d.encode_sparse(xg, max_active=16)  # forces the capacity retry
    acts = E.encode_ref(x, w["W_enc"], w["b_enc"], w["b_dec"])
    dense = d.encode_dense(xg).cpu()
    dirs = torch.randn(3, 768, generator=torch.Generator().manual_seed(8))
    scores = d.contrib_scores(feat, val, count, dirs.to(DEV)).cpu()
    recon = d.decode_sparse(feat, val, count).cpu()
    feat, val, count = feat.cpu(), val.cpu(), count.cpu()
    for m in range(x.shape[0]):
        c = int(count[m])
        # the GPU's own dense and sparse forms agree exactly (same kernel arithmetic)
        nz = dense[m].nonzero().flatten()
        assert c == nz.numel()
        order = torch.sort(dense[m][nz], descending=True, stable=True).indices
        assert torch.equal(feat[m, :c].long(), nz[order])
        assert torch.equal(val[m, :c], dense[m][nz][order])
        # against the oracle: values within tolerance, same set up to sub-tolerance activations
        rv, ri = L.sparse_desc(acts[m])
        big = rv > 10 * ATOL
        on_gpu = dict(zip(feat[m, :c].tolist(), val[m, :c].tolist()))
        for f_, v_ in zip(ri[big].tolist(), rv[big].tolist()):
            assert abs(on_gpu[f_] - v_) <= RTOL * abs(v_) + ATOL
        ref_sc = L.contribution_scores_ref(val[m, :c], feat[m, :c].long(), w["W_dec"], dirs)
        _close(scores[m, :c], ref_sc, atol=1e-5)
        assert bool((scores[m, c:] == 0).all())
    _close(recon, E.decode_ref(dense, w["W_dec"], w["b_dec_out"]), atol=1e-4)
    rows = d.decode_rows(feat[0, : int(count[0])].to(DEV), val[0, : int(count[0])].to(DEV)).cpu()
    _close(rows, w["W_dec"][feat[0, : int(count[0])].long()] * val[0, : int(count[0]), None], atol=1e-6)


def test_errors_are_raised_not_swallowed():
    import tsc_hip

    _, wt, _, _ = gi.g1_inputs()
    with pytest.raises(tsc_hip.TscError):
        tsc_hip.DeviceDict(wt["W_enc"], wt["b_enc"], wt["b_dec"])  # CPU tensors: no fallback
    d, _ = _dd(wt)
    with pytest.raises(tsc_hip.TscError):
        d.encode_dense(torch.zeros(4, 768))  # CPU input
    st = tsc_hip.TopkState(512, 20, DEV)
    with pytest.raises(tsc_hip.TscError):
        d.encode_topk(torch.zeros(4, 768, device=DEV), 0, st)  # d_sae mismatch
