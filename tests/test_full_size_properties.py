"""Parity at BASELINE.json's full dictionary widths through size-independent properties (the CPU
oracle cannot finish these sizes in seconds): the fused top-k must equal the ranking of the library's
own dense activations bit for bit, must not depend on how the token stream is cut into calls or on
the number of slots, and the dense activations must agree with the oracle on a sampled block."""
import numpy as np
import pytest
import torch

from oracle import encode_ref as E
from standins.synthetic_dicts import make_dictionary_weights

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RTOL, ATOL = 1e-4, 2e-5  # north star: activations within 1e-4 relative (fp32)


def _gpu_rank(acts: torch.Tensor, k: int, base: int = 0):
    """(value desc, token id asc) top-k of every column of a device matrix, positives only."""
    m, n = acts.shape
    ids = torch.arange(m, device=acts.device)
    # stable sort by value descending keeps ascending ids among equal values
    order = torch.sort(acts, dim=0, descending=True, stable=True).indices[:k]  # [k, n]
    vals = torch.gather(acts, 0, order)
    live = vals > 0
    vals = torch.where(live, vals, torch.zeros_like(vals))
    idx = torch.where(live, order + base, torch.full_like(order, -1))
    return vals.t().contiguous(), idx.t().contiguous()


@pytest.mark.parametrize("d_sae", [24576, 49152])
def test_fused_topk_full_width(d_sae):
    import tsc_hip

    w = make_dictionary_weights(768, d_sae, seed=21, b_enc=-0.70 if d_sae == 24576 else -0.5)
    g = {k: v.to(DEV) for k, v in w.items()}
    d = tsc_hip.DeviceDict(g["W_enc"], g["b_enc"], g["b_dec"])
    m = 6000  # ragged: 46.9 tiles of 128
    x = torch.randn(m, 768, generator=torch.Generator().manual_seed(d_sae))
    x[4000:4500] = x[100:600]  # exact ties across calls
    xg = x.to(DEV)
    dense = d.encode_dense(xg)

    # dense vs the oracle on a block of rows and a strided sample of columns
    cols = torch.arange(0, d_sae, 97)
    ref = E.encode_ref(x[:256], w["W_enc"][:, cols], w["b_enc"][cols], w["b_dec"])
    got = dense[:256].cpu()[:, cols]
    assert torch.allclose(got, ref, rtol=RTOL, atol=ATOL), float((got - ref).abs().max())
    frac_active = float((dense > 0).float().mean())
    assert 1e-4 < frac_active < 0.2, frac_active

    exp_v, exp_i = _gpu_rank(dense, 20, base=1_000_000)
    results = []
    for n_slots, cuts in ((None, [0, m]), (1, [0, 1, 130, 4097, m]), (5, [0, 3000, m])):
        st = tsc_hip.TopkState(d_sae, 20, DEV, n_slots=n_slots)
        for a, b in zip(cuts[:-1], cuts[1:]):
            d.encode_topk(xg[a:b], 1_000_000 + a, st)
        v, i = st.finalize()
        assert torch.equal(v, exp_v), (n_slots, float((v - exp_v).abs().max()))
        assert torch.equal(i, exp_i), (n_slots, int((i != exp_i).sum()))
        # lists are sorted (value desc, id asc) and padded with (0, -1)
        assert bool((v[:, :-1] >= v[:, 1:]).all())
        tie = (v[:, :-1] == v[:, 1:]) & (i[:, 1:] >= 0)
        assert bool((i[:, :-1][tie] < i[:, 1:][tie]).all())
        results.append((v, i))

    # merging the lists of two half-streams equals the whole stream (the multi-GPU rule)
    halves = []
    for a, b in ((0, 3000), (3000, m)):
        st = tsc_hip.TopkState(d_sae, 20, DEV)
        d.encode_topk(xg[a:b], 1_000_000 + a, st)
        halves.append(st.finalize())
    mv, mi = tsc_hip.topk_merge(torch.stack([halves[1][0], halves[0][0]]), torch.stack([halves[1][1], halves[0][1]]))
    assert torch.equal(mv, exp_v) and torch.equal(mi, exp_i)


def test_sparse_emit_full_width_matches_dense():
    import tsc_hip

    d_sae = 24576
    w = make_dictionary_weights(768, d_sae, seed=22, b_enc=-0.70)
    g = {k: v.to(DEV) for k, v in w.items()}
    d = tsc_hip.DeviceDict(g["W_enc"], g["b_enc"], g["b_dec"], g["W_dec"], g.get("b_dec_out", g["b_dec"]))
    x = torch.randn(300, 768, generator=torch.Generator().manual_seed(3)).to(DEV)
    dense = d.encode_dense(x)
    feat, val, count = d.encode_sparse(x, 256)
    assert torch.equal(count.long(), (dense > 0).sum(dim=1))
    for m in (0, 127, 128, 299):
        c = int(count[m])
        order = torch.sort(dense[m], descending=True, stable=True)
        assert torch.equal(val[m, :c], order.values[:c])
        assert torch.equal(feat[m, :c].long(), order.indices[:c])
    # decode of the sparse set == dense decode (linearity of the decoder in the activations)
    recon = d.decode_sparse(feat, val, count)
    ref = dense @ g["W_dec"] + g.get("b_dec_out", g["b_dec"])
    assert torch.allclose(recon, ref, rtol=1e-4, atol=1e-4), float((recon - ref).abs().max())


def test_long_stream_running_topk_full_width():
    """A longer sweep at the real width (20 calls: past the floor warm-up, so most calls reuse an older
    cross-slot floor): the fused lists must equal an exact running top-k kept with torch on the library's
    own dense activations, bit for bit."""
    import tsc_hip

    d_sae, k, batch, n_batches = 24576, 20, 8192, 20
    w = make_dictionary_weights(768, d_sae, seed=41, b_enc=-0.70)
    g = {kk: v.to(DEV) for kk, v in w.items()}
    d = tsc_hip.DeviceDict(g["W_enc"], g["b_enc"], g["b_dec"])
    st = tsc_hip.TopkState(d_sae, k, DEV)
    gen = torch.Generator(device=DEV).manual_seed(5)
    run_v = torch.zeros(k, d_sae, device=DEV)
    run_i = torch.full((k, d_sae), -1, dtype=torch.int64, device=DEV)
    first = None
    for b in range(n_batches):
        x = torch.randn(batch, 768, generator=gen, device=DEV)
        if b == 0:
            first = x[:512].clone()
        elif b % 5 == 0:
            x[100:612] = first  # the same contexts again: exact ties with older tokens
        base = b * batch
        d.encode_topk(x, base, st)
        dense = d.encode_dense(x)
        ids = torch.arange(base, base + batch, device=DEV)[:, None].expand(-1, d_sae)
        cat_v = torch.cat([run_v, dense])          # older entries first: the stable sort keeps smaller ids ahead
        cat_i = torch.cat([run_i, ids])
        order = torch.sort(cat_v, dim=0, descending=True, stable=True).indices[:k]
        run_v = torch.gather(cat_v, 0, order)
        run_i = torch.gather(cat_i, 0, order)
        del dense, cat_v, cat_i, ids
    live = run_v > 0
    exp_v = torch.where(live, run_v, torch.zeros_like(run_v)).t().contiguous()
    exp_i = torch.where(live, run_i, torch.full_like(run_i, -1)).t().contiguous()
    v, i = st.finalize()
    assert st._calls == n_batches
    assert torch.equal(v, exp_v), float((v - exp_v).abs().max())
    assert torch.equal(i, exp_i), int((i != exp_i).sum())


def test_all_24_dictionaries_full_width_step_loop_is_verified():
    """The benchmarked workload itself (BASELINE configs[1]: 12 transcoders + 12 z-SAEs at their real widths): a
    few steps of the bench's step loop, then EVERY dictionary's lists are checked against the library's own dense
    activations of EVERY step's batch with the bench's exact membership test (bench.verify_slice)."""
    import bench

    engine = bench.HipEngine(0, True)
    widths = bench.dictionary_widths()
    assert len(widths) == 24 and sum(w for _, _, w in widths) == 26 * 24576
    B, steps = 4096, 3
    gen = torch.Generator(device=DEV).manual_seed(99)
    dicts, states = [], []
    for i, (_, _, d_sae) in enumerate(widths):
        dicts.append(engine.make_dict(*bench.make_device_dictionary(d_sae, 500 + i, DEV)))
        states.append(engine.make_state(d_sae, 20))
    xs = [torch.randn(len(widths), B, 768, generator=gen, device=DEV) for _ in range(steps)]
    xs[2][:, 100:400] = xs[0][:, 100:400]  # the same contexts again: exact ties with older tokens
    for s in range(steps):
        for i, d in enumerate(dicts):
            engine.encode_topk(d, xs[s][i], 7 + s * B, states[i])
    for i, d in enumerate(dicts):
        v, ix = engine.finalize(states[i])
        assert float((ix[:, -1] >= 0).float().mean()) > 0.9  # 12288 tokens at 0.26 % density: ~32 candidates per feature
        for s in range(steps):
            assert bench.verify_slice(engine, d, xs[s][i], 7 + s * B, v, ix), (i, s)
    # and the check is not vacuous: a corrupted list is caught
    v, ix = engine.finalize(states[0])
    v2 = v.clone()
    v2[5, 3] = v2[5, 3] * 1.5
    assert not bench.verify_slice(engine, dicts[0], xs[0][0], 7, v2, ix)
