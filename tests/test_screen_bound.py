"""CPU check of the error bound behind the bf16 screen of ``tsc_sparse_lens_pass`` (include/tsc_hip.h,
DESIGN.md section 4.3): for every (row, feature), ``|exact fp32 pre-activation - bf16 pre-activation| <= margin`` with
``margin[n] = bound[0] max(||W_enc[:, n]||, ||bf16(W_enc[:, n])||) + bound[1] ||W_enc[:, n] - bf16(W_enc[:, n])|| + 2 d_in 2^-24
|b_enc[n]|`` and ``bound = tsc_hip.screen_row_bound(x, b_dec)`` (the actual rounding errors of the rows and the columns).
The bf16 side is emulated with torch (operands rounded to bf16, fp32 accumulation), the exact side is the oracle."""
import torch

from oracle import encode_ref as E


def _margin(x, w):
    import tsc_hip

    bound = tsc_hip.screen_row_bound(x, w["b_dec"])
    W = w["W_enc"].double()
    Wt = w["W_enc"].to(torch.bfloat16).double()
    wnorm = torch.maximum(W.norm(dim=0), Wt.norm(dim=0)).float() * (1 + 1e-6)   # what col_norm_kernel stores
    wdiff = (W - Wt).norm(dim=0).float() * (1 + 1e-6)
    g = 2.0 * x.shape[-1] * 2.0 ** -24
    return bound[0] * wnorm + bound[1] * wdiff + g * w["b_enc"].abs()


def _pre(x, w):
    return (x - w["b_dec"]) @ w["W_enc"] + w["b_enc"]


def _pre_bf16(x, w):
    centred = (x - w["b_dec"]).to(torch.bfloat16).float()
    return centred @ w["W_enc"].to(torch.bfloat16).float() + w["b_enc"]


def test_screen_margin_bounds_the_bf16_error():
    from standins.synthetic_dicts import make_dictionary_weights

    gen = torch.Generator().manual_seed(0)
    cases = []
    w = make_dictionary_weights(768, 2048, seed=1, b_enc=-0.7)
    cases.append((w, torch.randn(256, 768, generator=gen)))
    wc = make_dictionary_weights(768, 2048, seed=2, b_enc=-0.02)   # inputs that cancel against a large b_dec
    wc["b_dec"] = 3.0 * torch.randn(768, generator=gen)
    cases.append((wc, wc["b_dec"] + 0.05 * torch.randn(256, 768, generator=gen)))
    ws = make_dictionary_weights(768, 2048, seed=3, b_enc=-0.3)    # rows of very different norms, heavy-tailed entries
    cases.append((ws, torch.randn(256, 768, generator=gen) * torch.logspace(-2, 2, 256)[:, None]))
    wt = make_dictionary_weights(768, 2048, seed=4, b_enc=-0.1)
    wt["W_enc"] = wt["W_enc"] * torch.logspace(-3, 1, 2048)[None, :]  # feature norms over four decades
    cases.append((wt, torch.randn(128, 768, generator=gen).pow(3)))
    for w, x in cases:
        margin = _margin(x, w)
        exact64 = _pre(x.double(), {k: v.double() for k, v in w.items()})
        exact32 = _pre(x, w)
        folded = x @ w["W_enc"] + (w["b_enc"] - w["b_dec"] @ w["W_enc"])   # what the fp32 kernels compute
        approx = _pre_bf16(x, w)
        for exact in (exact64.float(), exact32, folded):
            err = (exact - approx).abs()
            assert bool((err <= margin[None, :]).all()), float((err / margin[None, :]).max())
        # and the bound is not absurd: Cauchy-Schwarz over 768 terms is loose by tens, not thousands
        assert float((margin[None, :] / (exact32 - approx).abs().clamp_min(1e-30)).min()) < 200
        # consequence: nothing active in the exact encode is screened out
        acts = E.encode_ref(x, w["W_enc"], w["b_enc"], w["b_dec"])
        assert bool(((approx + margin[None, :]) > 0)[acts > 0].all())
        # and the screen stays selective on ordinary inputs
        if w is cases[0][0]:
            assert float(((approx + margin[None, :]) > 0).float().mean()) < 2.5 * float((acts > 0).float().mean()) + 1e-4


def test_screen_margin_holds_at_bf16_rounding_midpoints():
    """Adversarial case (round-2 advisor): every operand sits on a bf16 rounding midpoint with aligned signs, so BOTH
    roundings of every product lose the full 2^-8 and nothing cancels: the exact pre-activation is +0.5 while the
    screened one is about -5.5.  A margin built on 2^-9 per rounding drops this active feature; 2^-8 per rounding
    (2^-7 per product) keeps it."""
    K, N = 768, 64
    gen = torch.Generator().manual_seed(5)
    mid = 1.0 + 2.0 ** -8                      # ties-to-even rounds it DOWN to 1.0: relative error 2^-8 / (1 + 2^-8)
    sign = torch.where(torch.rand(K, generator=gen) < 0.5, -1.0, 1.0)
    x = (mid * sign)[None, :].repeat(2, 1)
    x[1] = (1.0 + 3 * 2.0 ** -8) * sign       # the other tie direction: rounds UP to 1 + 2^-6
    W = (mid * sign)[:, None].repeat(1, N) * (2.0 ** torch.arange(-8, 8).repeat(N // 16))[None, :]  # exact scalings
    b_dec = torch.zeros(K)
    exact = (x.double() @ W.double())
    b_enc = (-(exact[0]) + 0.5).float()       # row 0 is active by +0.5 in every column
    w = {"W_enc": W, "b_enc": b_enc, "b_dec": b_dec}
    margin = _margin(x, w)
    approx = _pre_bf16(x, w)
    pre64 = exact + b_enc.double()
    assert bool((pre64[0] > 0).all())
    # the old coefficient (half this margin) would have lost the wide columns
    assert float((approx[0] + 0.5 * margin).min()) < 0
    err = (pre64.float() - approx).abs()
    assert bool((err <= margin[None, :]).all()), float((err / margin[None, :]).max())
    assert bool(((approx + margin[None, :]) > 0)[pre64 > 0].all())
    # and the bound is nearly attained here (nothing to give away): within 2 % of the worst row
    assert float((err / margin[None, :]).max()) > 0.9


def test_prompt_cache_view_slices_like_a_single_prompt_cache():
    from circuit_lens import _PromptCacheView
    from standins.synthetic_model import SyntheticHookedTransformer

    model = SyntheticHookedTransformer(n_layers=2, d_vocab=200, n_ctx=16, seed=1, d_mlp=64)
    prompts = ["a b c d e", "x y"]
    tokens = model.to_tokens(prompts)
    _, cache = model.run_with_cache(tokens)
    for i, p in enumerate(prompts):
        single_tokens = model.to_tokens(p)
        n = single_tokens.shape[1]
        _, single = model.run_with_cache(single_tokens)
        view = _PromptCacheView(cache, i, n)
        for key in (("z", 1), ("normalized", 0, "ln2"), ("pattern", 1), ("mlp_out", 1), "embed", ("resid_pre", 1), ("k", 0)):
            a, b = view[key], single[key]
            assert a.shape == b.shape, key
            assert torch.allclose(a, b, atol=1e-5), key  # right padding does not reach the real positions of a causal model
