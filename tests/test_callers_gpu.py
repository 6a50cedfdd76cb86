"""GPU parity of the caller-level drop-ins against ``g5_callers.npz`` -- the outputs of the reference's OWN
``MaxActAnalysis``, ``get_feature_scores``, ``get_top_k_activating_examples``, ``TaskEvaluation.logit_difference``
and ``CircuitLens`` run unmodified (tests/golden/make_golden_callers.py) -- plus the batched multi-prompt
CircuitLens, the fused lens pass, the hooked sweep and the local-file loaders."""
import json

import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import encode_ref as E
from oracle import lens_ref as L
from replay_model import ReplayModel
from test_hip_kernels import _close, _dd, _rank_own_dense

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ATOL = 1e-4  # the reference's own tolerance, test_circuit_lens.py:7


@pytest.fixture(autouse=True)
def _fresh_cache():
    import load_model_and_encoders as lme

    lme.clear_cache()
    prev = torch.is_grad_enabled()
    torch.set_grad_enabled(False)
    yield
    torch.set_grad_enabled(prev)
    lme.clear_cache()


def _replay_setup():
    import load_model_and_encoders as lme

    g5, w, wz, xs, n_seq, n_pos, prompts = gi.g5_inputs()
    model = ReplayModel(xs.to(DEV), n_seq, n_pos)
    tc = lme.build_transcoder(w, 0, DEV)
    zs = lme.build_zsae(wz, 0, DEV)
    lme.model_encoder_cache = (model, [zs], [tc])
    return g5, w, wz, xs, n_seq, n_pos, model, tc, zs


def test_max_act_analysis_against_the_reference_class():
    from max_act_analysis import MaxActAnalysis, get_top_k_tokens_for_features, top_k_token_ids

    g5, w, wz, xs, n_seq, n_pos, model, tc, zs = _replay_setup()
    tokens = model.token_matrix()
    acts = E.encode_ref(xs, w["W_enc"], w["b_enc"], w["b_dec"])
    for kind, f, key in (("mlp", 3, "maa_seq_pos_3"), ("mlp", 700, "maa_seq_pos_700"), ("mlp", 1535, "maa_seq_pos_1535"),
                         ("attn", 11, "maa_attn_seq_pos_11")):
        ma = MaxActAnalysis(kind, 0, f, batch_size=8, tokens=tokens)
        seq_pos, vals, scores = ma.active_examples
        ref_sp, ref_v = g5[key].astype(np.int64), g5[key.replace("seq_pos", "vals")]
        _close(vals, ref_v)                                   # values within the fp32 gate at every rank
        assert ma.num_examples == ref_sp.shape[0]
        diff = (seq_pos.numpy() != ref_sp).any(axis=1)
        if diff.any():  # a differing (seq, pos) must be a near-tie: the oracle's activation there equals the rank's value
            ref_acts = acts if kind == "mlp" else E.encode_ref(xs, wz["W_enc"], wz["b_enc"], wz["b_dec"])
            for r in np.nonzero(diff)[0]:
                tok = int(seq_pos[r, 0]) * n_pos + int(seq_pos[r, 1])
                assert abs(float(ref_acts[tok, f]) - float(ref_v[r])) <= 1e-4 * abs(float(ref_v[r])) + 2e-5
            assert diff.sum() <= 4
        if kind == "mlp" and ref_sp.shape[0]:
            s, p, v = ma.get_active_example(0)
            assert [int(s), int(p)] == g5[f"maa_example0_{f}"].tolist()
    # get_top_k_tokens: VALUES, for an mlp and an attn feature (max_act_analysis.py:156-170)
    for kind, f in (("mlp", 3), ("mlp", 700), ("attn", 11), ("attn", 1000)):
        toks = MaxActAnalysis(kind, 0, f, tokens=tokens).get_top_k_tokens(k=5)
        assert [int(t) for t in toks] == g5[f"top_tokens_{kind}_{f}"].tolist()
    # the batched [F, 768] x W_U form gives the same lists, and logits that match the per-feature GEMV
    many = get_top_k_tokens_for_features(model, zs, "attn", 0, [11, 1000, 5], k=5)
    assert [int(t) for t in many[0]] == g5["top_tokens_attn_11"].tolist()
    assert [int(t) for t in many[1]] == g5["top_tokens_attn_1000"].tolist()
    logits, ids = top_k_token_ids(model, tc, "mlp", 0, [3, 700], k=5)
    ref = (tc.W_dec[[3, 700]] @ model.W_U).topk(5, dim=1)
    assert torch.equal(ids, ref.indices) and torch.allclose(logits, ref.values, rtol=1e-5, atol=1e-6)


def test_feature_scores_all_modes_and_top_examples_against_the_reference():
    from autointerpretability import get_feature_scores, get_top_k_activating_examples

    g5, w, wz, xs, n_seq, n_pos, model, tc, zs = _replay_setup()
    tokens = model.token_matrix()
    g2 = gi.load("g2_topk.npz")
    post = (torch.randn(768, 3, generator=torch.Generator().manual_seed(77)) * 0.05).to(DEV)
    raw = get_feature_scores(model, tc, tokens, [5, 6, 7], batch_size=8, use_raw_scores=True, feature_post=post)
    _close(raw.astype(np.float32), g5["gfs_post_raw"].astype(np.float32), rtol=2e-3, atol=2e-3)
    dec = get_feature_scores(model, tc, tokens, [5, 6, 7], batch_size=8, use_raw_scores=True, use_decoder=True,
                             feature_post=post)
    _close(dec.astype(np.float32), g5["gfs_post_decoder"].astype(np.float32), rtol=2e-3, atol=2e-3)
    z = get_feature_scores(model, zs, tokens, [11, 1000], batch_size=8)
    assert z.dtype == np.float16
    _close(z.astype(np.float32), g5["gfs_zsae"].astype(np.float32), rtol=2e-3, atol=1e-4)
    eot = get_feature_scores(model, tc, tokens[:8], g2["gfs_feats"].tolist(), batch_size=8, ignore_endoftext=True)
    _close(eot.astype(np.float32), g5["gfs_ignore_eot"].astype(np.float32), rtol=2e-3, atol=1e-4)
    with pytest.raises(RuntimeError):
        get_feature_scores(model, tc, tokens, [5], use_raw_scores=True, use_decoder=True)  # the reference fails here too
    # get_top_k_activating_examples on the PRODUCT's own scores: values, (sequence, position) and the score rows
    feats = g2["gfs_feats"].tolist()
    sc = get_feature_scores(model, tc, tokens, feats, batch_size=8)
    for j, f in enumerate(feats[:3]):
        strs, per_seq, pos = get_top_k_activating_examples(sc[:, j, :], tokens, model, k=5)
        seq = np.array([int(s[1]) for s in strs])
        got_scores = sc[seq, j, np.asarray(pos)]
        ref_scores = g2["gfs_scores"][g5[f"gtk_seq_{f}"], j, g5[f"gtk_pos_{f}"]]
        if np.array_equal(sc[:, j, :], g2["gfs_scores"][:, j, :]):  # same fp16 slab: the ranking must be the reference's
            assert gi.same_ranking(np.stack([seq, np.asarray(pos)], 1), got_scores,
                                   np.stack([g5[f"gtk_seq_{f}"], g5[f"gtk_pos_{f}"]], 1), ref_scores)
        else:  # an fp16 rounding boundary moved: the top scores still agree to fp16 precision
            _close(got_scores.astype(np.float32), ref_scores.astype(np.float32), rtol=2e-3, atol=1e-4)
        assert len(per_seq) == 5 and np.array_equal(np.array(per_seq, np.float16), sc[seq, j, :])
        assert all(len(s) == n_pos for s in strs)


def _lens_world(n_layers=3, dead_z=False):
    import load_model_and_encoders as lme
    from standins.synthetic_model import SyntheticHookedTransformer

    g3, g4, _, z_w, t_w = gi.lens_inputs(n_layers)
    if dead_z:
        z_w = [dict(w, b_enc=torch.full_like(w["b_enc"], -100.0)) for w in z_w]
    model = SyntheticHookedTransformer(n_layers=n_layers, d_vocab=1000, n_ctx=32, seed=7, device=DEV)
    lme.install_synthetic(DEV, z_weights=z_w, t_weights=t_w, model=model)
    return g3, g4, model, z_w, t_w


def _ids(runs):
    return [[gi.COMPONENT_IDS.index(r.component), r.run_data["layer"], r.run_data["feature"]] for r, _ in runs]


def _assert_mismatches_are_near_ties(mismatch_ranks, ref_sorted_values, tol=2 * ATOL):
    """A rank at which two rankings name different components is tolerated only where the reference's own values
    make it a near-tie: the value at that rank is within ``tol`` of its neighbour above or below (``ref_sorted_values``
    reaches one rank past the compared ones, so that the boundary of the top-k counts too)."""
    v = [float(x) for x in ref_sorted_values]
    for j in mismatch_ranks:
        gaps = [abs(v[j] - v[i]) for i in (j - 1, j + 1) if 0 <= i < len(v)]
        assert gaps and min(gaps) <= tol, (j, v)


def test_logit_difference_against_the_reference_class():
    from task_evaluation import TaskEvaluation

    g5, *_, prompts = gi.g5_inputs()
    _lens_world()
    te = TaskEvaluation(prompts, None, no_mean_cache=True)
    assert np.array_equal(te.tokens.cpu().numpy(), g5["te_tokens"])
    dirs = te.logit_difference_directions()
    for i in range(len(prompts)):
        d = te.logit_difference(i)
        assert np.array_equal(d.cpu().numpy(), g5[f"te_logit_diff_{i}"])  # a gather and a subtraction: exact
        assert torch.equal(dirs[i], d[-1])
    with pytest.raises(ValueError):
        TaskEvaluation([{"text": "a b", "correct": None, "counter": " x"}], None, no_mean_cache=True).logit_difference(0)
    te2 = TaskEvaluation(prompts[:3], None)  # with the mean cache (:131-139)
    assert te2.mean_logits.shape == (te2.tokens.shape[1], 1000)
    with pytest.raises(NotImplementedError):
        te.get_circuit_discovery_for_prompt(0)


def test_lens_rankings_contribs_and_top_contribs_against_the_reference():
    from circuit_lens import CircuitLens

    g5, *_ = gi.g5_inputs()
    _lens_world()
    cl = CircuitLens(gi.PROMPT)
    s2, tok = g5["unembed2_seq_tok"].tolist()
    runs, _ = cl.get_unembed_lens(tok, s2, visualize=False, k=10)
    assert len(runs) == 10 and _ids(runs) == g5["unembed2_top_ids"].tolist()  # a full top-10, no embed break
    _close(torch.tensor([v for _, v in runs]), g5["unembed2_top_vals"], atol=ATOL)
    layer, s3, feat = g5["mlp_lens2_layer_seq_feature"].tolist()
    runs, _ = cl.get_mlp_feature_lens_at_seq(layer, s3, feat, visualize=False, k=10)
    assert len(runs) == 10 and _ids(runs) == g5["mlp_lens2_top_ids"].tolist()
    _close(torch.tensor([v for _, v in runs]), g5["mlp_lens2_top_vals"], atol=ATOL)
    seq = cl.process_seq_index(-3)
    tokn = int(cl.tokens[0, seq + 1])
    direction = cl.model.W_U[:, tokn] - cl.model.W_U[:, int(cl.tokens[0, 1])]
    for l in range(3):
        c = cl.get_transcoder_contribs(l, seq, direction.clone(), k=5)
        assert [x[3] for x in c] == g5[f"tc_contribs_idx_l{l}"].tolist()
        _close(torch.tensor([x[4] for x in c]), g5[f"tc_contribs_val_l{l}"], atol=1e-5)
    # get_top_contribs: the one case in which the reference returns (no active z feature) -- VALUES
    _lens_world(dead_z=True)
    cl2 = CircuitLens(gi.PROMPT)
    top = cl2.get_top_contribs(direction.clone(), 2, seq, k=5)
    assert [[x[1], x[3]] for x in top] == g5["top_contribs_layer_idx"].tolist()
    _close(torch.tensor([x[4] for x in top]), g5["top_contribs_val"], atol=1e-5)
    _close(torch.stack([x[0] for x in top]) @ gi.proj8().to(DEV), g5["top_contribs_vec_proj"], atol=ATOL)


def test_batched_circuit_lens_equals_single_prompt_lenses():
    """many prompts (of different lengths: right padding) through ONE pass; every per-prompt view
    must reproduce the single-prompt CircuitLens, the reference goldens, and the fused direction lens."""
    from circuit_lens import BatchedCircuitLens, CircuitLens
    from task_evaluation import TaskEvaluation

    g5, *_, prompts = gi.g5_inputs()
    g3, g4, model, z_w, t_w = _lens_world()
    texts = [p["text"] for p in prompts[:5]] + [gi.PROMPT, "Jeff went"]
    batch = BatchedCircuitLens(texts, positions="all")
    assert batch.n_rows == sum(batch.lengths) and len(set(batch.lengths)) > 1
    proj = gi.proj8().to(DEV)
    gold = batch.lens(5)  # the reference's golden prompt, seen through the batch
    for pos in (-3, 5):
        af = gold.get_active_features(pos)
        tag = f"p{pos % gold.n_tokens}"
        assert [[k["attn"], k["mlp"]] for k in af.keys] == g3[f"{tag}_keys"].tolist()
        assert np.array_equal(af.features.cpu().numpy(), g3[f"{tag}_features"].astype(np.int64))
        _close(af.values, g3[f"{tag}_values"])
        _close(af.vectors @ proj, g3[f"{tag}_vec_proj"], atol=ATOL)
    seq = gold.process_seq_index(-3)
    tok = int(gold.tokens[0, seq + 1])
    runs, _ = gold.get_unembed_lens(tok, seq, visualize=False, k=10)
    assert _ids(runs) == g4["unembed_top_ids"].tolist()
    pb, _ = gold.get_transcoder_ixg(gold.mlp_transcoders[1], 1, seq, torch.from_numpy(g4["dirs"][0]).to(DEV))
    _close(pb, g4["ixg_d0_l1"], atol=1e-5)
    dirs = torch.randn(len(texts), 768, generator=torch.Generator().manual_seed(6)).to(DEV)
    for i in (0, 3, 5, 6):
        single = CircuitLens(texts[i])
        view = batch.lens(i)
        assert torch.equal(single.tokens, view.tokens)
        for pos in (1, -1):
            a, b = single.get_active_features(pos), view.get_active_features(pos)
            assert a.keys == b.keys and torch.equal(a.features, b.features)
            _close(b.values, a.values)
            _close(b.vectors, a.vectors, atol=ATOL)
    # fused direction lens over every row == per-prompt `normalize(vectors @ direction)` and its top-k
    res = batch.direction_lens(dirs, k=10)
    runs_all = batch.direction_lens_runs(dirs, k=10)
    row = 0
    for i, n in enumerate(batch.lengths):
        view = batch.lens(i)
        for pos in range(n):
            if i in (0, 5, 6) and pos in (0, n // 2, n - 1):
                af = view.get_active_features(pos)
                ref = view.normalize_activations(af.vectors @ dirs[i])
                got = res["activations"][row][res["mask"][row]]
                _close(got, ref, atol=ATOL)
                ref_runs, _ = view.get_direction_lens(dirs[i], pos, visualize=False, k=10)
                got_runs = runs_all[row]
                assert len(got_runs) == len(ref_runs)
                mism = [j for j, (a, b) in enumerate(zip(_ids(got_runs), _ids(ref_runs))) if a != b]
                assert len(mism) <= 2  # near-equal contributions may swap ...
                _assert_mismatches_are_near_ties(mism, ref.sort(descending=True).values[:len(ref_runs) + 1])  # ... and only those
                _close(torch.tensor([v for _, v in got_runs]), torch.tensor([v for _, v in ref_runs]), atol=ATOL)
            row += 1
    # the evaluation-position form (config 4): one row per prompt, logit-difference directions
    te = TaskEvaluation(prompts, None, no_mean_cache=True)
    out = te.contribution_decomposition(k=10)
    d = te.logit_difference_directions()
    assert out["top_values"].shape == (len(prompts), 10)
    for i in (0, 4, 7):
        single = CircuitLens(prompts[i]["text"])
        af = single.get_active_features(-1)
        ref = single.normalize_activations(af.vectors @ d[i])
        _close(out["activations"][i][out["mask"][i]], ref, atol=ATOL)
        v, idx = ref.topk(10)
        _close(out["top_values"][i], v, atol=ATOL)
        feats = af.features[idx]
        differ = (out["top_feature"][i] != feats).nonzero().flatten().tolist()
        assert len(differ) <= 2
        _assert_mismatches_are_near_ties(differ, ref.topk(11).values)
    with pytest.raises(ValueError):
        te.batched_lens().lens(0).get_active_features(0)  # position not covered by an eval-position pass



def test_direction_lens_with_the_decoders_folded_through_w_o_equals_the_z_space_form():
    """``BatchedCircuitLens.direction_lens`` scores residual directions with the z-SAE decoders folded through W_O once
    (``_ResidualSpaceZ``: no per-pass ``W_O @ d``).  Same contributions as the z-space form (the reference's order:
    project the z vectors, then dot, src/circuit_lens.py:618-627) to the reference's tolerance, same winners up to
    near-ties; a parameter of a z-SAE updated in place is picked up (the folded decoder is rebuilt)."""
    from circuit_lens import BatchedCircuitLens

    g5, *_, prompts = gi.g5_inputs()
    _lens_world()
    texts = [p["text"] for p in prompts[:8]]
    dirs = torch.randn(len(texts), 768, generator=torch.Generator().manual_seed(11)).to(DEV)

    def run(residual, positions):
        batch = BatchedCircuitLens(texts, positions=positions)
        batch.RESIDUAL_DECODER = residual
        return batch, batch.direction_lens(dirs, k=10)

    for positions in (-1, "all"):
        b1, folded = run(True, positions)
        b0, zspace = run(False, positions)
        assert torch.equal(folded["mask"], zspace["mask"])
        _close(folded["activations"], zspace["activations"], atol=ATOL)
        _close(folded["top_values"], zspace["top_values"], atol=ATOL)
        for r in range(folded["top_feature"].shape[0]):
            mism = (folded["top_feature"][r] != zspace["top_feature"][r]).nonzero().flatten().tolist()
            assert len(mism) <= 2
            _assert_mismatches_are_near_ties(mism, zspace["top_values"][r].tolist() + [float("-inf")])
        # the per-prompt views of the SAME batch still see z-space reconstructions (their own pass)
        af = b1.lens(2).get_active_features(-1)
        ref = b0.lens(2).get_active_features(-1)
        _close(af.vectors, ref.vectors, atol=ATOL)
    z = b1.z_saes[1]
    before = b1.direction_lens(dirs, k=10)["activations"].clone()
    with torch.no_grad():
        z.W_dec.mul_(1.5)                                   # in place: the fold must be redone
    after = b1.direction_lens(dirs, k=10)
    b0.RESIDUAL_DECODER = False
    want = b0.direction_lens(dirs, k=10)
    assert not torch.allclose(after["activations"], before, atol=ATOL)
    _close(after["activations"], want["activations"], atol=ATOL)
    with torch.no_grad():
        z.W_dec.div_(1.5)

def test_batched_lens_revalidates_its_cached_handles():
    """``BatchedCircuitLens`` reuses the library handles and call tables of its first pass without checking them on
    the way in (the check runs while the GPU works).  A parameter updated in place, or re-allocated, between two
    passes must still give the results of a fresh batch: the pass that ran on stale handles is redone."""
    from circuit_lens import BatchedCircuitLens

    g5, *_, prompts = gi.g5_inputs()
    _lens_world()
    texts = [p["text"] for p in prompts[:6]]
    dirs = torch.randn(len(texts), 768, generator=torch.Generator().manual_seed(9)).to(DEV)
    batch = BatchedCircuitLens(texts, positions=-1)
    first = batch.direction_lens(dirs, k=10)
    again = batch.direction_lens(dirs, k=10)          # cached handles, nothing changed
    for name in ("top_values", "top_feature", "activations"):
        assert torch.equal(first[name], again[name]), name

    def fresh():
        return BatchedCircuitLens(texts, positions=-1).direction_lens(dirs, k=10)

    batch.mlp_transcoders[1].b_enc.add_(0.05)          # in place: same storage, new version -> refresh + rerun
    got, want = batch.direction_lens(dirs, k=10), fresh()
    assert not torch.equal(got["activations"], first["activations"])
    for name in ("top_values", "top_feature", "activations"):
        assert torch.equal(got[name], want[name]), name
    z = batch.z_saes[0]                                # re-allocated: a new handle -> rerun
    z.W_enc = torch.nn.Parameter((z.W_enc.detach() * 1.01).clone())
    got, want = batch.direction_lens(dirs, k=10), fresh()
    for name in ("top_values", "top_feature", "activations"):
        assert torch.equal(got[name], want[name]), name


def test_fused_lens_pass_matches_the_separate_kernels():
    g, wt, wz, x = gi.g1_inputs()
    w = gi.g1_full_weights()
    d, gw = _dd(w)
    m = 200
    xg = torch.randn(m, 768, generator=torch.Generator().manual_seed(12)).to(DEV)
    dirs = torch.randn(m, 768, generator=torch.Generator().manual_seed(13)).to(DEV)
    feat, val, count = d.encode_sparse(xg, 256)
    recon = d.decode_sparse(feat, val, count)
    f2, v2, c2, r2, s2 = d.sparse_lens_pass(xg, 256, dirs=dirs)
    assert torch.equal(c2, count) and torch.equal(f2, feat) and torch.equal(v2, val)
    _close(r2, recon, atol=1e-5)
    for row in (0, 77, m - 1):  # per-row directions: bit-identical to tsc_contrib_scores with that row's direction
        sc = d.contrib_scores(feat[row:row + 1].contiguous(), val[row:row + 1].contiguous(), count[row:row + 1].contiguous(),
                              dirs[row:row + 1].contiguous())
        assert torch.equal(s2[row], sc[0, :, 0])
    shared = d.sparse_lens_pass(xg, 256, dirs=dirs[0].contiguous(), want_recon=False)
    assert shared[3] is None
    assert torch.equal(shared[4], d.contrib_scores(feat, val, count, dirs[:1].contiguous())[:, :, 0])
    # against the oracle
    acts = E.encode_ref(xg.cpu(), w["W_enc"], w["b_enc"], w["b_dec"])
    for row in (3, 150):
        c = int(count[row])
        ref = L.contribution_scores_ref(val[row, :c].cpu(), feat[row, :c].long().cpu(), w["W_dec"], dirs[row:row + 1].cpu())
        _close(s2[row, :c], ref[:, 0], atol=1e-5)
        assert bool((s2[row, c:] == 0).all())
    _close(r2, E.decode_ref(acts, w["W_dec"], w["b_dec_out"]), atol=1e-4)
    # capacity overflow is reported through count, like tsc_encode_sparse
    small = d.sparse_lens_pass(xg, 8, dirs=dirs)
    assert torch.equal(small[2], count) and int(count.max()) > 8


def _same_active_sets(feat_a, val_a, count_a, feat_b, val_b, count_b, atol=1e-6):
    """Two sparse results agree: every feature whose activation exceeds ``2 atol`` on either side is on both sides
    with the same value to fp32 rounding (``atol`` = the rounding noise of the fp32 sums, which grows with the
    magnitude of the terms that cancel in them), each side sorted by value."""
    worst, tiny = 0.0, 2 * atol
    for m in range(feat_a.shape[0]):
        a = {int(f): float(v) for f, v in zip(feat_a[m, :int(count_a[m])].tolist(), val_a[m, :int(count_a[m])].tolist())}
        b = {int(f): float(v) for f, v in zip(feat_b[m, :int(count_b[m])].tolist(), val_b[m, :int(count_b[m])].tolist())}
        for f in set(a) | set(b):
            va, vb = a.get(f, 0.0), b.get(f, 0.0)
            if max(va, vb) > tiny:
                assert f in a and f in b, (m, f, va, vb)
                assert abs(va - vb) <= 1e-5 * max(va, vb) + atol, (m, f, va, vb)
                worst = max(worst, abs(va - vb))
        vb_sorted = val_b[m, :int(count_b[m])]
        assert bool((vb_sorted[:-1] >= vb_sorted[1:]).all())
    return worst


def test_screened_lens_pass_gives_the_exact_results():
    """TSC_F32_SCREEN: bf16 MFMA screen + exact fp32 recompute of the candidates must find every active feature of
    the plain fp32 pass with the same values -- on ordinary inputs, on inputs that sit close to a large b_dec (the
    folded bias cancels to two digits: the worst case for the screen's error bound), on a ragged dictionary width,
    and with the row bound computed by the library or handed in."""
    import tsc_hip
    from standins.synthetic_dicts import make_dictionary_weights

    cases = []
    w = gi.g1_full_weights()
    x = torch.randn(300, 768, generator=torch.Generator().manual_seed(21))
    cases.append((w, x, 256, 5e-6))  # two summation orders of a 768-term fp32 dot product whose terms are O(1)
    wc = make_dictionary_weights(768, 1536, seed=31, b_enc=0.0)
    gen = torch.Generator().manual_seed(32)
    wc["b_dec"] = 3.0 * torch.randn(768, generator=gen)
    wc["b_enc"] = torch.full_like(wc["b_enc"], -0.02)
    # x @ W_enc and b_dec @ W_enc are ~50x the activations here and cancel: fp32 sums carry ~1e-6 of absolute noise
    cases.append((wc, wc["b_dec"] + 0.05 * torch.randn(200, 768, generator=gen), 1024, 2e-5))
    wr = make_dictionary_weights(768, 1000, seed=77, b_enc=-0.9)
    xr = torch.randn(77, 768, generator=torch.Generator().manual_seed(5)) * torch.linspace(0.2, 3.0, 77)[:, None]  # row norms vary 15x
    cases.append((wr, xr, 1024, 1e-5))  # (one bound per call = the largest row's: the short rows' candidates are many)
    for w, x, cap, atol in cases:
        plain, _ = _dd(w)
        screen, _ = _dd(w, compute_dtype=tsc_hip.F32_SCREEN)
        xg = x.to(DEV)
        dirs = torch.randn(x.shape[0], 768, generator=torch.Generator().manual_seed(1)).to(DEV)
        fa, va, ca, ra, sa = plain.sparse_lens_pass(xg, cap, dirs=dirs)
        fb, vb, cb, rb, sb = screen.sparse_lens_pass(xg, cap, dirs=dirs)                      # library computes the bound
        bound = tsc_hip.screen_row_bound(xg, w["b_dec"].to(DEV))
        fc, vc, cc, rc, sc = screen.sparse_lens_pass(xg, cap, dirs=dirs, row_bound=bound)     # bound handed in
        assert int(ca.max()) <= cap and int(cb.max()) <= cap
        assert int(ca.sum()) > 20 * x.shape[0]  # the case is not vacuous
        for f2, v2, c2, r2, s2 in ((fb, vb, cb, rb, sb), (fc, vc, cc, rc, sc)):
            _same_active_sets(fa.cpu(), va.cpu(), ca.cpu(), f2.cpu(), v2.cpu(), c2.cpu(), atol)
            _close(r2, ra, atol=1e-4)
            # scores follow the (feature-matched) values
            for m in (0, x.shape[0] - 1):
                a = dict(zip(fa[m, :int(ca[m])].tolist(), sa[m, :int(ca[m])].tolist()))
                b = dict(zip(f2[m, :int(c2[m])].tolist(), s2[m, :int(c2[m])].tolist()))
                for f in a.keys() & b.keys():
                    assert abs(a[f] - b[f]) <= 1e-5 * abs(a[f]) + 4 * atol
        # the other entry points of a screening handle are the plain fp32 ones, bit for bit
        assert torch.equal(screen.encode_dense(xg), plain.encode_dense(xg))
        # the candidates may overflow a capacity that the true active set fits: reported through count, as documented
        tight = int(ca.max())
        ft, vt, ct, _, _ = screen.sparse_lens_pass(xg, tight, dirs=dirs)
        over = ct > tight
        ok_rows = (~over).nonzero().flatten().tolist()
        if ok_rows:
            _same_active_sets(fa[ok_rows].cpu(), va[ok_rows].cpu(), ca[ok_rows].cpu(), ft[ok_rows].cpu(), vt[ok_rows].cpu(),
                              ct[ok_rows].cpu(), atol)


def test_hooked_sweep_is_bit_identical_to_the_cache_sweep():
    """forward hooks into a preallocated input ring + encodes on a side stream give the same lists as
    the reference-shaped run_with_cache loop."""
    import load_model_and_encoders as lme
    from standins.synthetic_dicts import make_dictionary_weights
    from standins.synthetic_model import SyntheticHookedTransformer
    from topk_scan import sweep_model

    model = SyntheticHookedTransformer(n_layers=4, d_vocab=300, n_ctx=16, seed=3, device=DEV, d_mlp=256)
    z_w = [make_dictionary_weights(768, 512, seed=50 + l, b_enc=-0.6, transcoder=False) for l in range(4)]
    t_w = [make_dictionary_weights(768, 512, seed=80 + l, b_enc=-2.5, transcoder=True) for l in range(4)]
    _, z_saes, tcs = lme.install_synthetic(DEV, z_weights=z_w, t_weights=t_w, model=model)
    tokens = torch.randint(1, 300, (70, 16), generator=torch.Generator().manual_seed(1)).to(DEV)
    a = sweep_model(model, tokens, z_saes, tcs, k=7, batch_size=8, capture="cache", distributed=False)
    b = sweep_model(model, tokens, z_saes, tcs, k=7, batch_size=8, capture="hooks", distributed=False)
    c = sweep_model(model, tokens, z_saes, tcs, k=7, batch_size=8, distributed=False)  # auto -> hooks
    assert set(a) == set(b) == {f"z_{l}" for l in range(4)} | {f"mlp_{l}" for l in range(4)}
    for name in a:
        assert torch.equal(a[name][0], b[name][0]) and torch.equal(a[name][1], b[name][1]), name
        assert torch.equal(a[name][1], c[name][1])
    assert int((a["mlp_2"][1] >= 0).sum()) > 0


def test_local_file_loaders_round_trip(tmp_path):
    """the reference's checkpoint layouts from LOCAL files -- transcoder ``{"cfg", "state_dict"}`` .pt
    (mlp_transcoder.py:602-659), z-SAE ``<run>_cfg.json`` + ``<run>.pt`` (z_sae.py:79-89)."""
    from mlp_transcoder import SparseTranscoder
    from z_sae import ZSAE

    g, wt, wz, x = gi.g1_inputs()
    cfg = {"d_in": 768, "d_sae": 1536, "l1_coefficient": 2e-3, "hook_point": "blocks.8.ln2.hook_normalized",
           "hook_point_layer": 8}
    torch.save({"cfg": cfg, "state_dict": {k: v.clone() for k, v in wt.items()}}, tmp_path / "tc.pt")
    tc = SparseTranscoder.load_from_pretrained(str(tmp_path / "tc.pt"), device=DEV)
    assert tc.cfg.hook_point_layer == 8 and tc.cfg.hook_point == "blocks.8.ln2.hook_normalized" and tc.cfg.is_transcoder
    assert tc.l1_coefficient == 2e-3 and tc.W_enc.is_cuda
    _close(tc.encode(x), g["tc_encode"])
    _close(tc(x)[0], g["tc_sae_out"], atol=ATOL)
    torch.save({k: v.clone() for k, v in wt.items()}, tmp_path / "bare.pt")  # a bare state dict loads too
    _close(SparseTranscoder.load_from_pretrained(str(tmp_path / "bare.pt"), device=DEV).encode(x), g["tc_encode"])
    with pytest.raises(ValueError):
        SparseTranscoder.load_from_pretrained(str(tmp_path / "tc.pkl.gz"))  # pickles execute code on load: refused
    zcfg = {"dict_size": 1536, "l1_coeff": 1.0, "enc_dtype": "fp32", "seed": 3, "act_size": 768, "layer": 8,
            "device": "cuda:7"}  # the device recorded at training time must not matter
    json.dump(zcfg, open(tmp_path / "run_cfg.json", "w"))
    torch.save({k: v.clone() for k, v in wz.items()}, tmp_path / "run.pt")
    zs = ZSAE.load_from_files(str(tmp_path / "run_cfg.json"), str(tmp_path / "run.pt"), device=DEV)
    assert zs.cfg["layer"] == 8 and zs.W_enc.is_cuda
    _close(zs.encode(x), g["zs_encode"])
    _close(zs(x)[1], g["zs_recon"], atol=ATOL)


def test_state_limits_and_infinite_activations():
    import ctypes

    import tsc_hip
    from standins.synthetic_dicts import make_dictionary_weights

    with pytest.raises(tsc_hip.TscError):
        tsc_hip.TopkState(256, 64, DEV, n_slots=64)  # could be filled but never finalized: refused up front
    lib = tsc_hip.lib()
    st = tsc_hip._TopkStateStruct(16, 16, 256, 64, 33, 0, None)
    assert lib.tsc_topk_state_reset(ctypes.byref(st), None) == -3 and b"n_slots * k" in lib.tsc_last_error()
    ok = tsc_hip.TopkState(256, 64, DEV, n_slots=32)  # exactly at the limit
    assert ok.finalize()[0].shape == (256, 64)
    # an infinite activation in the last real column of a ragged dictionary must not spill into the lists of the
    # columns past d_sae (= the next slot's lists): ADVICE r1
    w = make_dictionary_weights(768, 1000, seed=77, b_enc=-0.9)
    d, _ = _dd(w)
    x = torch.randn(300, 768, generator=torch.Generator().manual_seed(5))
    x[17] = torch.sign(w["W_enc"][:, 999]) * 1e38  # column 999 overflows to +inf, the others stay finite
    xg = x.to(DEV)
    dense = d.encode_dense(xg)
    assert torch.isinf(dense[17, 999])
    exp_v, exp_i = _rank_own_dense(dense, 5)
    state = tsc_hip.TopkState(1000, 5, DEV, n_slots=2)
    d.encode_topk(xg, 0, state)
    v, i = state.finalize()
    assert torch.equal(i.cpu(), exp_i) and torch.equal(v.cpu(), exp_v)
    # padding ids decode to zero rows
    rows = d.decode_rows(torch.tensor([-1, 3], dtype=torch.int32, device=DEV), torch.tensor([2.0, 2.0], device=DEV))
    assert bool((rows[0] == 0).all()) and torch.allclose(rows[1], 2.0 * w["W_dec"][3].to(DEV))


def test_sparse_emit_sorts_long_lists():
    """A dense-ish dictionary: thousands of active features per position (the emit holds up to 8192), which takes
    the bitonic path of the sort; the result must be the exact descending order of the library's own dense row."""
    from standins.synthetic_dicts import make_dictionary_weights

    w = make_dictionary_weights(768, 8192, seed=3, b_enc=0.05)
    d, _ = _dd(w)
    x = torch.randn(9, 768, generator=torch.Generator().manual_seed(4))
    x[3] = x[2]  # equal rows: ties between positions are irrelevant, ties inside a row are broken by feature id
    xg = x.to(DEV)
    dense = d.encode_dense(xg)
    feat, val, count = d.encode_sparse(xg, 8192)
    assert torch.equal(count.long(), (dense > 0).sum(dim=1)) and int(count.min()) > 2000
    for m in range(9):
        c = int(count[m])
        order = torch.sort(dense[m], descending=True, stable=True)
        assert torch.equal(val[m, :c], order.values[:c]) and torch.equal(feat[m, :c].long(), order.indices[:c])
        assert bool((feat[m, c:] == -1).all()) and bool((val[m, c:] == 0).all())
    # a capacity between the rank-sort limit and the count: the overflow is reported, the kept prefix is sorted
    f2, v2, c2 = d.encode_sparse(xg, 1024, check_capacity=False)
    assert torch.equal(c2, count) and bool((v2[:, :-1] >= v2[:, 1:]).all())


def test_module_copies_drop_handles_and_autograd_is_refused():
    import copy
    import pickle

    import tsc_hip
    from load_model_and_encoders import build_transcoder, build_zsae

    g, wt, wz, x = gi.g1_inputs()
    tc, zs = build_transcoder(wt, 8, DEV), build_zsae(wz, 8, DEV)
    _close(tc.encode(x), g["tc_encode"])  # registers the handle
    for clone in (copy.deepcopy(tc), pickle.loads(pickle.dumps(tc))):
        assert clone._hip is None and clone.W_enc.data_ptr() != tc.W_enc.data_ptr()
        _close(clone.encode(x), g["tc_encode"])
    _close(copy.deepcopy(zs).encode(x), g["zs_encode"])
    with torch.enable_grad():
        xr = x.clone().to(DEV).requires_grad_()
        with pytest.raises(tsc_hip.TscError):
            tc.encode(xr)
        with pytest.raises(tsc_hip.TscError):
            zs(xr)
        _close(tc.encode(x.to(DEV)), g["tc_encode"])  # inputs that do not require grad are fine with grad mode on


@pytest.mark.parametrize("screened,m", [(False, 77), (True, 77), (True, 3000)])
def test_grouped_finish_equals_the_per_dictionary_lens_pass(screened, m):
    """``tsc_sparse_lens_finish_many`` after ``TSC_LENS_EMIT_ONLY`` emits: the same kernel body over the rows of
    several dictionaries in one launch -- results bit for bit those of ``tsc_sparse_lens_pass`` per dictionary; and
    ``tsc_sparse_lens_emit_many`` (screening handles: ONE grouped launch of the emits) in front of it."""
    import tsc_hip
    from standins.synthetic_dicts import make_dictionary_weights

    cap, n = 512, 3   # m = 3000: twelve 256-row tiles, every workgroup of the grouped emit walks several
    ws = [make_dictionary_weights(768, d_sae, seed=60 + i, b_enc=-1.0) for i, d_sae in enumerate((1536, 1000, 4100))]
    hs = [_dd(w, compute_dtype=tsc_hip.F32_SCREEN if screened else tsc_hip.F32)[0] for w in ws]
    gen = torch.Generator().manual_seed(8)
    xs = [torch.randn(m, 768, generator=gen).to(DEV) for _ in range(n)]
    dirs = [torch.randn(m, 768, generator=gen).to(DEV) for _ in range(n)]
    ref = [h.sparse_lens_pass(x, cap, dirs=d) for h, x, d in zip(hs, xs, dirs)]
    feat = torch.empty(n, m, cap, dtype=torch.int32, device=DEV)
    val = torch.empty(n, m, cap, device=DEV)
    count = torch.zeros(n, m, dtype=torch.int32, device=DEV)
    recon = torch.empty(n, m, 768, device=DEV)
    scores = torch.empty(n, m, cap, device=DEV)
    for i, (h, x, d) in enumerate(zip(hs, xs, dirs)):
        h.sparse_lens_pass(x, cap, dirs=d, emit_only=True,
                           out={"feat": feat[i], "val": val[i], "count": count[i], "recon": recon[i], "scores": scores[i]})
    tsc_hip.sparse_lens_finish_many(hs, xs, feat, val, count, dirs=dirs, recon=recon, scores=scores)
    for i, (f, v, c, r, s) in enumerate(ref):
        assert int(c.max()) <= cap and int(c.sum()) > 5 * m
        assert torch.equal(count[i], c) and torch.equal(feat[i], f) and torch.equal(val[i], v)
        assert torch.equal(recon[i], r) and torch.equal(scores[i], s)
    # without scores / reconstruction, the emits through the one-call form (two streams), and the argument checks
    count.zero_()
    side = [torch.cuda.Stream(), torch.cuda.Stream()]
    for st in side:
        st.wait_stream(torch.cuda.current_stream())
    tsc_hip.sparse_lens_emit_many(hs, xs, feat, val, count, streams=side,
                                  row_bounds=torch.stack([tsc_hip.screen_row_bound(x, w["b_dec"].to(DEV)) for x, w in zip(xs, ws)]) if screened else None)
    for st in side:
        torch.cuda.current_stream().wait_stream(st)
    tsc_hip.sparse_lens_finish_many(hs, xs, feat, val, count)
    assert all(torch.equal(feat[i], ref[i][0]) and torch.equal(val[i], ref[i][1]) for i in range(n))
    with pytest.raises(tsc_hip.TscError):
        tsc_hip.sparse_lens_finish_many(hs[:2], xs[:2], feat, val, count)
    if screened:
        mixed = [hs[0], _dd(ws[1])[0], hs[2]]
        with pytest.raises(tsc_hip.TscError):
            tsc_hip.sparse_lens_finish_many(mixed, xs, feat, val, count)


def test_fused_per_node_lens_equals_the_vectors_path():
    """The per-node lens calls of the circuit search (unembed, direction, MLP-feature, V, Q, K lenses) through the fused
    path -- W_O pull-back, tsc_contrib_scores_many, tsc_rowdot_many, tsc_lens_assemble with the cut-off, no
    [components, d_model] matrix -- against the reference-shaped path ``normalize(vectors_before @ direction).topk``:
    same components in the same order (a differing rank only at a near-tie of the reference-shaped values), same
    values to the reference's ATOL; single-prompt lens and a per-prompt view of a batch."""
    from circuit_lens import BatchedCircuitLens, CircuitLens

    g3, g4, model, z_w, t_w = _lens_world()
    g5, *_, prompts = gi.g5_inputs()
    single = CircuitLens(gi.PROMPT)
    view = BatchedCircuitLens([p["text"] for p in prompts[:3]] + [gi.PROMPT], positions="all").lens(3)
    gen = torch.Generator().manual_seed(12)
    for lens in (single, view):
        n = lens.n_tokens
        mf, _, _ = lens._active("mlp", 1, n - 3)
        zf, _, _ = lens._active("attn", 2, n - 3)
        calls = [
            lambda l: l.get_unembed_lens(int(l.tokens[0, n - 2]), n - 3, visualize=False, k=10),
            lambda l: l.get_direction_lens(torch.randn(768, generator=gen.manual_seed(3)).to(DEV), 4, visualize=False, k=7),
            lambda l: l.get_mlp_feature_lens_at_seq(1, n - 3, int(mf[0]), visualize=False, k=10),
            lambda l: l.get_mlp_feature_lens_at_seq(0, 2, int(l._active("mlp", 0, 2)[0][0]), visualize=False, k=10),
            lambda l: l.get_v_lens_at_seq(2, 5, n - 3, int(zf[0]), visualize=False, k=10),
            lambda l: l.get_q_lens_on_head_seq(1, 3, 2, n - 3, visualize=False, k=10),
            lambda l: l.get_k_lens_on_head_seq(2, 7, 2, n - 3, visualize=False, k=10),
        ]
        for call in calls:
            lens.FUSED_LENS = True
            got, lazy = call(lens)
            lens.FUSED_LENS = False
            want, af = call(lens)
            assert len(got) == len(want) and len(got) > 0
            gv, wv = torch.tensor([v for _, v in got]), torch.tensor([v for _, v in want])
            _close(gv, wv, atol=ATOL)
            mism = [j for j, (a, b) in enumerate(zip(_ids(got), _ids(want))) if a != b]
            _assert_mismatches_are_near_ties(mism, wv.tolist() + [float("-inf")])
            assert [r.tuple_id[0] for r, _ in got] == [r.tuple_id[0] for r, _ in want] or mism
            # the stand-in for ActiveFeatures builds the real one on demand
            assert lazy.keys == af.keys and torch.equal(lazy.features, af.features)
        lens.FUSED_LENS = True
