"""``tsc_encode_topk_screened``: the fused top-k with the contraction on the bf16 MFMA as a screen and an exact fp32
recompute of the candidates must give the lists of ``tsc_encode_topk`` BIT FOR BIT -- values and token ids -- whatever
the data: warm lists, cold lists (queue overflow -> the fp32 fallback launch), exact ties, ragged shapes, several
slots, calls of both kinds interleaved on one state."""
import pytest
import torch

import golden_inputs as gi
from test_hip_kernels import _dd, _rank_own_dense

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pair(w):
    import tsc_hip

    plain, _ = _dd(w)
    screen, _ = _dd(w, compute_dtype=tsc_hip.F32_SCREEN)
    return plain, screen


@pytest.mark.parametrize("k,n_slots,batch", [(20, 4, 1024), (20, 1, 700), (7, 3, 333), (33, 2, 4096)])
def test_screened_topk_is_bit_identical_small(k, n_slots, batch):
    import tsc_hip

    g, w, xs, n_seq, n_pos = gi.g2_inputs()  # d_sae 1536: 128-wide tiles, short queues
    gen = torch.Generator().manual_seed(11)
    a, b, c = xs[:900], xs[900:1500], xs[1500:2300]
    x = torch.cat([a, b, a[:500], c, b, a[200:700], c[:50].repeat(6, 1)])  # repeated contexts: exact ties
    x = x[torch.randperm(x.shape[0], generator=gen)].contiguous()
    plain, screen = _pair(w)
    xg = x.to(DEV)
    exp_v, exp_i = _rank_own_dense(plain.encode_dense(xg), k, base=50)
    ref = tsc_hip.TopkState(1536, k, DEV, n_slots=n_slots)
    all_screened = tsc_hip.TopkState(1536, k, DEV, n_slots=n_slots)   # from the first call on: cold lists, overflow path
    mixed = tsc_hip.TopkState(1536, k, DEV, n_slots=n_slots)          # fp32 first, then screened, then fp32 again
    for j, s in enumerate(range(0, xg.shape[0], batch)):
        chunk = xg[s:s + batch]
        plain.encode_topk(chunk, 50 + s, ref)
        screen.encode_topk_screened(chunk, 50 + s, all_screened)
        if j == 0 or j % 3 == 2:
            screen.encode_topk(chunk, 50 + s, mixed)
        else:
            screen.encode_topk_screened(chunk, 50 + s, mixed)
    rv, ri = ref.finalize()
    assert torch.equal(rv.cpu(), exp_v) and torch.equal(ri.cpu(), exp_i)
    for st in (all_screened, mixed):
        v, i = st.finalize()
        assert torch.equal(v, rv), float((v - rv).abs().max())
        assert torch.equal(i, ri), int((i != ri).sum())


@pytest.mark.parametrize("d_sae", [24576, 49152])
def test_screened_topk_is_bit_identical_full_width(d_sae):
    """Real widths (256-wide tiles, 8 / 4 slots), 40 000 rows per call (two chunks of the staging buffer), lists warm
    after the first call, exact ties across calls, a row bound handed in for the second half."""
    import tsc_hip
    from standins.synthetic_dicts import make_dictionary_weights

    w = make_dictionary_weights(768, d_sae, seed=21, b_enc=-0.70 if d_sae == 24576 else -0.5)
    plain, screen = _pair(w)
    gen = torch.Generator(device=DEV).manual_seed(d_sae)
    ref = tsc_hip.TopkState(d_sae, 20, DEV)
    scr = tsc_hip.TopkState(d_sae, 20, DEV)
    first = None
    n_calls, rows = 5, 40000
    for j in range(n_calls):
        x = torch.randn(rows, 768, generator=gen, device=DEV)
        if j == 0:
            first = x[:3000].clone()
        elif j % 2 == 0:
            x[500:3500] = first  # the same contexts again
        plain.encode_topk(x, j * rows, ref)
        if j == 0:
            screen.encode_topk(x, j * rows, scr)          # warm-up on the fp32 kernel, as a sweep would
        elif j < 3:
            screen.encode_topk_screened(x, j * rows, scr)
        else:
            bound = tsc_hip.screen_row_bound(x, w["b_dec"].to(DEV))
            screen.encode_topk_screened(x, j * rows, scr, row_bound=bound)
    rv, ri = ref.finalize()
    v, i = scr.finalize()
    assert torch.equal(v, rv), float((v - rv).abs().max())
    assert torch.equal(i, ri), int((i != ri).sum())
    assert bool((ri[:, -1] >= 0).all())  # 200 000 tokens: every list is full


def test_screened_topk_argument_checks():
    import tsc_hip

    g, w, xs, n_seq, n_pos = gi.g2_inputs()
    plain, screen = _pair(w)
    x = xs[:64].to(DEV)
    st = tsc_hip.TopkState(1536, 20, DEV)
    with pytest.raises(tsc_hip.TscError):
        plain.encode_topk_screened(x, 0, st)             # not a screening handle
    with pytest.raises(tsc_hip.TscError):
        screen.encode_topk_screened(x, 0, tsc_hip.TopkState(1536, 20, DEV, cross_slot_floor=False))
    assert tsc_hip.lib().tsc_topk_screen_workspace_bytes(24576, 8) > 8 * 96 * 8192 * 4


def test_screened_topk_with_non_finite_rows_matches_the_fp32_path():
    """Rows holding NaN (their activations are NaN on both paths and enter no list; the row bound skips them) and a
    row holding +inf (the bound becomes infinite, every pair is a candidate, the queues overflow and the fp32 launch
    does the chunk) must leave the same lists as ``tsc_encode_topk``."""
    import tsc_hip

    g, w, xs, n_seq, n_pos = gi.g2_inputs()
    plain, screen = _pair(w)
    x = xs[:3000].clone()
    x[17, 5] = float("nan")
    x[1200] = float("nan")
    xg = x.to(DEV)
    ref = tsc_hip.TopkState(1536, 20, DEV, n_slots=2)
    st = tsc_hip.TopkState(1536, 20, DEV, n_slots=2)
    plain.encode_topk(xg[:1000], 0, ref)   # warm both states on the fp32 kernel
    screen.encode_topk(xg[:1000], 0, st)
    plain.encode_topk(xg[1000:], 1000, ref)
    screen.encode_topk_screened(xg[1000:], 1000, st)   # NaN row 1200 inside a warm, screened call
    y = xs[3000:4000].clone()
    y[3, 100] = float("inf")
    yg = y.to(DEV)
    plain.encode_topk(yg, 3000, ref)
    screen.encode_topk_screened(yg, 3000, st)
    rv, ri = ref.finalize()
    v, i = st.finalize()
    assert torch.equal(i, ri), int((i != ri).sum())
    assert torch.equal(torch.nan_to_num(v, nan=-7.0), torch.nan_to_num(rv, nan=-7.0))
    assert not bool((ri == 1200).any()) and not bool((ri == 17).any())   # NaN rows are in no list


def _midpoint_case(n_rows=640, d_sae=1536, seed=9, exps=(-6, 6)):
    """Operands on bf16 rounding midpoints with aligned signs (round-2 advisor's counter-example): each product loses
    the full 2^-7 in the screen and nothing cancels.  Rows ``hot`` carry exactly the sign pattern of the columns:
    their exact pre-activation is +0.5 in every column while the bf16 one is about -5.5 * 2^e of the column."""
    from standins.synthetic_dicts import make_dictionary_weights

    K = 768
    gen = torch.Generator().manual_seed(seed)
    mid = 1.0 + 2.0 ** -8
    sign = torch.where(torch.rand(K, generator=gen) < 0.5, -1.0, 1.0)
    w = make_dictionary_weights(K, d_sae, seed=seed, b_enc=0.0)
    scale = 2.0 ** torch.arange(*exps).repeat(d_sae // (exps[1] - exps[0]))
    w["W_enc"] = ((mid * sign)[:, None] * scale[None, :]).contiguous()
    w["b_dec"] = torch.zeros(K)
    hot_row = mid * sign
    w["b_enc"] = (-(hot_row.double() @ w["W_enc"].double()) + 0.5).float()
    x = mid * torch.where(torch.rand(n_rows, K, generator=gen) < 0.5, -1.0, 1.0)   # same norm, other sign patterns
    hot = [h for h in (3, 77, 200, 201, 639) if h < n_rows]
    x[hot] = hot_row
    return w, x.contiguous(), hot


def test_screened_topk_keeps_candidates_at_bf16_rounding_midpoints():
    import tsc_hip

    w, x, hot = _midpoint_case()
    plain, screen = _pair(w)
    xg = x.to(DEV)
    ref = tsc_hip.TopkState(1536, 20, DEV, n_slots=2)
    scr = tsc_hip.TopkState(1536, 20, DEV, n_slots=2)
    plain.encode_topk(xg, 10, ref)
    screen.encode_topk_screened(xg, 10, scr)
    rv, ri = ref.finalize()
    v, i = scr.finalize()
    assert len(hot) == 5 and sorted(ri[5, :5].tolist()) == [10 + h for h in hot] and float(rv[5, 0]) > 0.25   # the case is what it claims
    assert torch.equal(v, rv), float((v - rv).abs().max())
    assert torch.equal(i, ri), int((i != ri).sum())


def test_screened_lens_pass_keeps_features_at_bf16_rounding_midpoints():
    w, x, hot = _midpoint_case(n_rows=96, exps=(-4, 2))   # column scales up to 2: the fp32 sums of ~1500 keep +0.5 visible
    plain, screen = _pair(w)
    xg = x.to(DEV)
    fa, va, ca, _, _ = plain.sparse_lens_pass(xg, 2048, want_recon=False)
    fb, vb, cb, _, _ = screen.sparse_lens_pass(xg, 2048, want_recon=False)
    assert int(ca[hot[0]]) >= 1536 // 2            # the hot rows are active (by +0.5) in every column the fp32 sum keeps
    assert int(cb.max()) <= 2048
    for m in range(x.shape[0]):
        a = dict(zip(fa[m, :int(ca[m])].tolist(), va[m, :int(ca[m])].tolist()))
        b = dict(zip(fb[m, :int(cb[m])].tolist(), vb[m, :int(cb[m])].tolist()))
        # the columns with large scale carry fp32 summation noise of ~768 * 2^e * 2^-24 around +0.5: compare generously,
        # but every feature clearly active on one side must be on the other
        for f in set(a) | set(b):
            if max(a.get(f, 0.0), b.get(f, 0.0)) > 0.05:
                assert f in a and f in b, (m, f, a.get(f), b.get(f))
                assert abs(a[f] - b[f]) <= 0.03, (m, f, a[f], b[f])
