"""GPU parity of the drop-in Python surface (same names and call shapes as the reference's modules)
against the golden vectors generated from the reference, plus the reference's own CircuitLens
invariants (``src/test_circuit_lens.py``) restated on the stand-in model."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import encode_ref as E
from replay_model import ReplayModel
from test_hip_kernels import _assert_topk_equal, _close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ATOL = 1e-4  # the reference's own tolerance, test_circuit_lens.py:7


@pytest.fixture(autouse=True)
def _fresh_cache():
    import load_model_and_encoders as lme

    lme.clear_cache()
    prev = torch.is_grad_enabled()
    torch.set_grad_enabled(False)  # as every caller in the reference does (e.g. max_activating.py:98)
    yield
    torch.set_grad_enabled(prev)
    lme.clear_cache()


def test_dictionary_classes_match_reference_outputs():
    from load_model_and_encoders import build_transcoder, build_zsae

    g, wt, wz, x = gi.g1_inputs()
    tc, zs = build_transcoder(wt, 8, DEV), build_zsae(wz, 8, DEV)
    _close(tc.encode(x), g["tc_encode"])            # CPU input is moved, as the reference does (:196-198)
    _close(tc.encode(x.view(4, 16, 768)), g["tc_encode_3d"])
    out = tc(x)
    assert len(out) == 6
    _close(out[0], g["tc_sae_out"], atol=ATOL)
    _close(out[1], g["tc_encode"])
    zout = zs(x)
    assert len(zout) == 5
    _close(zout[1], g["zs_recon"], atol=ATOL)
    _close(zout[2], g["zs_encode"])
    _close(zs.encode(x), g["zs_encode"])
    # a hook on the post-activation still fires; inner hook points refuse
    seen = []
    tc.hook_hidden_post.register_forward_hook(lambda m, i, o: seen.append(o.shape))
    tc.encode(x)
    assert seen == [(64, 1536)]
    tc.hook_hidden_pre.register_forward_hook(lambda m, i, o: None)
    with pytest.raises(NotImplementedError):
        tc.encode(x)


def _g2_setup(kind="mlp"):
    import load_model_and_encoders as lme

    g, w, xs, n_seq, n_pos = gi.g2_inputs()
    model = ReplayModel(xs.to(DEV), n_seq, n_pos)
    tc = lme.build_transcoder(w, 0, DEV)
    lme.model_encoder_cache = (model, [None], [tc])
    return g, w, xs, n_seq, n_pos, model, tc


def test_in_place_weight_updates_are_seen():
    """The handle caches operands derived from the weights (folded bias, transposed W_enc, the bf16
    copy); an optimiser-style in-place update of the parameters must show in the next encode."""
    from load_model_and_encoders import build_transcoder

    g, wt, wz, x = gi.g1_inputs()
    tc = build_transcoder(wt, 8, DEV)
    _close(tc.encode(x), g["tc_encode"])
    w2 = {k: v.clone() for k, v in wt.items()}
    gen = torch.Generator().manual_seed(9)
    w2["b_dec"] += 0.3 * torch.randn(768, generator=gen)
    w2["b_enc"] += 0.1
    w2["W_enc"] *= 1.0 + 0.05 * torch.randn(768, 1, generator=gen)
    tc.b_dec.add_(w2["b_dec"].to(DEV) - tc.b_dec)          # same storage, new values
    tc.b_enc.copy_(w2["b_enc"].to(DEV))
    tc.W_enc.mul_((w2["W_enc"] / wt["W_enc"]).to(DEV))
    ref = E.encode_ref(x, w2["W_enc"], w2["b_enc"], w2["b_dec"])
    assert not torch.allclose(ref, torch.as_tensor(g["tc_encode"]), atol=1e-2)
    _close(tc.encode(x), ref)
    st = tc.new_topk_state(k=4)
    tc.encode_topk(x, 0, st)
    v, _ = st.finalize()
    _close(v[:, 0], ref.max(dim=0).values)


def test_max_act_analysis_active_examples():
    from max_act_analysis import MaxActAnalysis

    g, w, xs, n_seq, n_pos, model, tc = _g2_setup()
    tokens = model.token_matrix()
    for f in (3, 700, 1535):
        ma = MaxActAnalysis("mlp", 0, f, batch_size=8, tokens=tokens)
        seq_pos, vals, scores = ma.active_examples
        assert scores.shape == (n_seq, n_pos) and not scores.is_cuda
        _close(vals, g[f"a5_vals_{f}"])
        ref_sp = g[f"a5_seq_pos_{f}"].astype(np.int64)
        if not np.array_equal(seq_pos.numpy(), ref_sp):  # only near-tie swaps may differ
            diff = (seq_pos.numpy() != ref_sp).any(axis=1)
            assert diff.sum() <= 2
        assert ma.num_examples == ref_sp.shape[0]
        if ref_sp.shape[0]:  # a dead feature has an empty tuple (feature 1535 here)
            s, p, v = ma.get_active_example(0)
            assert (int(s), int(p)) == tuple(ref_sp[0]) and abs(float(v) - float(g[f"a5_vals_{f}"][0])) < 1e-4
        assert ma.active_examples is ma.active_examples  # memoised (:201-202)
    toks = MaxActAnalysis("mlp", 0, 3, tokens=tokens).get_top_k_tokens(k=5)  # values: tests/test_callers_gpu.py (g5)
    assert [int(t) for t in toks] == (tc.W_dec[3] @ model.W_U).topk(5).indices.tolist()
    with pytest.raises(AssertionError):
        MaxActAnalysis("resid", 0, 3, tokens=tokens)
    # all features at once through the fused kernel
    vals, seq, pos = MaxActAnalysis("mlp", 0, 3, batch_size=8, tokens=tokens).top_examples_all_features(k=20)
    idx = torch.where(seq >= 0, seq * n_pos + pos, seq)
    acts = E.encode_ref(xs, w["W_enc"], w["b_enc"], w["b_dec"])
    _assert_topk_equal(vals, idx, torch.from_numpy(g["vals"]), torch.from_numpy(g["idx"].astype(np.int64)), acts)


def test_get_feature_scores_and_top_examples():
    from autointerpretability import get_feature_scores, get_top_k_activating_examples

    g, w, xs, n_seq, n_pos, model, tc = _g2_setup()
    tokens = model.token_matrix()
    feats = g["gfs_feats"].tolist()
    sc = get_feature_scores(model, tc, tokens, feats, batch_size=8)
    assert sc.dtype == np.float16 and sc.shape == (n_seq, len(feats), n_pos)
    _close(sc.astype(np.float32), g["gfs_scores"].astype(np.float32), rtol=2e-3, atol=1e-4)
    raw = get_feature_scores(model, tc, tokens, feats, batch_size=8, use_raw_scores=True)
    _close(raw.astype(np.float32), g["gfs_raw"].astype(np.float32), rtol=2e-3, atol=2e-3)
    post = tc.W_enc[:, feats]
    raw2 = get_feature_scores(model, tc, tokens, feats, batch_size=8, use_raw_scores=True, feature_post=post)
    _close(raw2.astype(np.float32), g["gfs_raw"].astype(np.float32), rtol=2e-3, atol=2e-3)
    strs, per_seq, pos = get_top_k_activating_examples(g["gfs_scores"][:, 2, :], tokens, model, k=5)
    seq = np.array([int(s_[1]) for s_ in strs])  # the replay tokens carry the sequence id in column 1
    assert np.array_equal(g["gfs_scores"][seq, 2, np.asarray(pos)], g["gtk_scores"])  # the reference's top-5 scores
    assert np.array_equal(np.array(per_seq, np.float16), g["gfs_scores"][seq, 2, :])
    with pytest.raises(ValueError):
        get_feature_scores(model, object(), tokens, feats)


def _lens(n_layers=3):
    import load_model_and_encoders as lme
    from circuit_lens import CircuitLens

    g3, g4, model_cpu, z_w, t_w = gi.lens_inputs(n_layers)
    from standins.synthetic_model import SyntheticHookedTransformer

    model = SyntheticHookedTransformer(n_layers=n_layers, d_vocab=1000, n_ctx=32, seed=7, device=DEV)
    lme.install_synthetic(DEV, z_weights=z_w, t_weights=t_w, model=model)
    cl = CircuitLens(gi.PROMPT)
    assert np.array_equal(cl.tokens.cpu().numpy(), g3["tokens"])
    return g3, g4, cl, z_w, t_w


def _ids(runs):
    return [[gi.COMPONENT_IDS.index(r.component), r.run_data["layer"], r.run_data["feature"]] for r, _ in runs]


def test_circuit_lens_active_features_match_reference():
    g3, g4, cl, _, _ = _lens()
    proj = gi.proj8().to(DEV)
    for pos in (-3, 5):
        af = cl.get_active_features(pos)
        tag = f"p{pos % cl.n_tokens}"
        assert [[k["attn"], k["mlp"]] for k in af.keys] == g3[f"{tag}_keys"].tolist()
        assert np.array_equal(af.features.cpu().numpy(), g3[f"{tag}_features"].astype(np.int64))
        _close(af.values, g3[f"{tag}_values"])
        _close(af.vectors @ proj, g3[f"{tag}_vec_proj"], atol=ATOL)
        _close(af.vectors.sum(0), g3[f"{tag}_vec_sum"], atol=ATOL)
        assert cl.get_active_features(pos) is af
    with pytest.raises(ValueError):
        from circuit_lens import CircuitLens

        cl.model.to_tokens = lambda p: torch.zeros(2, 3, dtype=torch.long)
        CircuitLens("x")


def test_circuit_lens_scores_match_reference():
    g3, g4, cl, z_w, t_w = _lens()
    seq = cl.process_seq_index(-3)
    af = cl.get_active_features(seq)
    tok = int(cl.tokens[0, seq + 1])
    _close(af.vectors @ cl.model.W_U[:, tok], g4["unembed_raw"], atol=ATOL)
    runs, af2 = cl.get_unembed_lens(tok, seq, visualize=False, k=10)
    assert af2.vectors is af.vectors and _ids(runs) == g4["unembed_top_ids"].tolist()  # (a stand-in that resolves to the cached object)
    _close(torch.tensor([v for _, v in runs]), g4["unembed_top_vals"], atol=ATOL)
    masked = (af.vectors @ cl.model.W_U[:, tok]).clone()
    masked[:2] = -1e9
    tk = af.get_top_k_features(masked, k=40)
    got = [[gi.COMPONENT_IDS.index(c), l, f] for c, l, f, _ in tk]
    ref = g4["topk_masked_ids"].tolist()
    assert sum(a != b for a, b in zip(got, ref)) <= 2  # near-equal scores may swap
    _close(torch.tensor([v for *_, v in tk]), g4["topk_masked_vals"], atol=ATOL)
    layer, feat = 2, int(g4["mlp_lens_feature"])
    runs, _ = cl.get_mlp_feature_lens_at_seq(layer, seq, feat, visualize=False, k=10)
    assert _ids(runs) == g4["mlp_lens_top_ids"].tolist()
    zfeat = int(g4["z_feature"])
    _close(cl.get_head_seq_activations_for_z_feature(layer, seq, zfeat), g4["head_seq_acts"], atol=ATOL)
    _close(cl.get_v_lens_activations(layer, 4, seq, zfeat), g4["v_lens"], atol=ATOL)
    _close(cl.get_q_lens_activations(layer, 4, 2, seq), g4["q_lens"], atol=ATOL)
    _close(cl.get_k_lens_activations(layer, 4, 2, seq), g4["k_lens"], atol=ATOL)
    dirs = torch.from_numpy(g4["dirs"]).to(DEV)
    proj = gi.proj8().to(DEV)
    for l in range(3):
        for d in range(2):
            pb, acts = cl.get_transcoder_ixg(cl.mlp_transcoders[l], l, seq, dirs[d])
            _close(pb, g4[f"ixg_d{d}_l{l}"], atol=1e-5)
            if d == 0:
                _close(acts, g4[f"ixg_acts_l{l}"])
        _close(cl.get_attn_head_contribs(l, seq, dirs[0]) @ proj, g4[f"attn_contribs_proj_l{l}"], atol=ATOL)
        # batched sparse contribution scores for every position, both directions at once
        feat, scores, count = cl.contribution_scores("mlp", l, dirs)
        c = int(count[seq])
        ref = torch.from_numpy(g4[f"ixg_d1_l{l}"]).to(DEV)
        _close(scores[seq, :c, 1], ref[feat[seq, :c].long()], atol=1e-5)
    tcc = cl.get_transcoder_contribs(1, seq, dirs[0].clone(), k=5)
    assert [c[3] for c in tcc] == g4["tc_contribs_idx"].tolist()
    _close(torch.tensor([c[4] for c in tcc]), g4["tc_contribs_val"], atol=1e-5)
    _close(torch.stack([c[0] for c in tcc]) @ proj, g4["tc_contribs_vec_proj"], atol=ATOL)
    # get_top_contribs = top-k by score over the per-layer contributions (the reference's own result for it, in the
    # one case where its code returns, is pinned in tests/test_callers_gpu.py)
    top = cl.get_top_contribs(dirs[0].clone(), 2, seq, k=5)
    pool = [c for l in range(3) for c in cl.get_transcoder_contribs(l, seq, dirs[0].clone(), k=5)]
    best = sorted(pool, key=lambda c: -c[4])[:5]
    assert [(c[1], c[3]) for c in top] == [(c[1], c[3]) for c in best]
    _close(torch.tensor([c[4] for c in top]), torch.tensor([c[4] for c in best]), atol=1e-6)


def test_reference_invariants_on_standin_model():
    """test_circuit_lens.py:141-219, 222-269, 334-377 restated."""
    g3, g4, cl, _, _ = _lens()
    seq = cl.process_seq_index(-3)
    for layer in range(3):
        af = cl.get_active_features(seq, cache=False)
        _close(af.get_reconstructed_attn_out(layer), cl.cache["attn_out", layer][0, seq], atol=ATOL)
        _close(af.get_reconstructed_mlp_out(layer), cl.cache["mlp_out", layer][0, seq], atol=ATOL)
        _close(af.get_reconstructed_resid_post(layer), cl.cache["resid_post", layer][0, seq], atol=3e-4)
        out = cl.mlp_transcoders[layer](cl.cache["normalized", layer, "ln2"])[0][0, seq]
        _close(af.get_transcoder_reconstruction(layer), out, atol=ATOL)
        z = cl.cache["z", layer][0, seq].reshape(-1)
        _, z_recon, z_acts, _, _ = cl.z_saes[layer](z)
        _close(af.get_sae_out_reconstruction(layer), z_recon @ cl._w_o_flat(layer), atol=ATOL)
        # max-feature rows are W_dec[topk idx] * topk vals (test_compare_max_mlp_features :105-138)
        mlp_acts = cl.mlp_transcoders[layer](cl.cache["normalized", layer, "ln2"][:, seq])[1][0]
        n = int((mlp_acts != 0).sum())
        vals, idx = mlp_acts.topk(k=n)
        _close(af.get_mlp_feature_vectors(layer), cl.mlp_transcoders[layer].W_dec[idx] * vals[:, None], atol=ATOL)
        # encode formula from the decomposition (test_compare_mlp_feature_activations :222-269)
        f = int(mlp_acts.argmax())
        resid_mid = af.get_reconstructed_resid_mid(layer)
        scale = cl.cache["scale", layer, "ln2"][0, seq]
        normed = resid_mid / scale - resid_mid.mean() / scale
        tc = cl.mlp_transcoders[layer]
        value = torch.dot(normed - tc.b_dec, tc.W_enc[:, f]) + tc.b_enc[f]
        _close(value, mlp_acts[f], atol=3e-4)
        # head/seq decomposition of a z feature (:334-377)
        zf = int(z_acts.argmax())
        total = cl.get_head_seq_activations_for_z_feature(layer, seq, zf).sum()
        zs = cl.z_saes[layer]
        target = z_acts[zf] - zs.b_enc[zf] + torch.dot(zs.b_dec, zs.W_enc[:, zf])
        b_v_term = 0  # the stand-in model has value biases, which the reference's identity ignores for GPT-2 (folded)
        _close(total + b_v_term, target, atol=5e-2)


def test_activation_dictionary_and_sweep():
    import load_model_and_encoders as lme
    from max_activating import get_activation_dictionary, get_topk_dictionary
    from standins.synthetic_dicts import make_dictionary_weights
    from standins.synthetic_model import SyntheticHookedTransformer
    from oracle.topk_ref import topk_stream_ref

    model = SyntheticHookedTransformer(n_layers=12, d_vocab=300, n_ctx=16, seed=3, device=DEV, d_mlp=256)
    z_w = [make_dictionary_weights(768, 256, seed=50 + l, b_enc=-0.6, transcoder=False) for l in range(12)]
    t_w = [make_dictionary_weights(768, 256, seed=80 + l, b_enc=-2.5, transcoder=True) for l in range(12)]
    _, z_saes, tcs = lme.install_synthetic(DEV, z_weights=z_w, t_weights=t_w, model=model)
    tokens = torch.randint(1, 300, (24, 16), generator=torch.Generator().manual_seed(1)).to(DEV)
    _, cache = model.run_with_cache(tokens[:8])
    ad = get_activation_dictionary(cache, z_saes, tcs, tokens[:8])
    assert set(ad) == {"tokens"} | {f"z_{l}" for l in range(12)} | {f"mlp_{l}" for l in range(12)}
    assert ad["z_3"].shape == (8, 16, 256)
    x3 = cache["normalized", 3, "ln2"].cpu()
    _close(ad["mlp_3"], E.encode_ref(x3, t_w[3]["W_enc"], t_w[3]["b_enc"], t_w[3]["b_dec"]), atol=2e-4)
    res = get_topk_dictionary(model, tokens, z_saes, tcs, k=5, batch_size=8)
    _, cache_all = model.run_with_cache(tokens)
    for name, w, x in (("mlp_7", t_w[7], cache_all["normalized", 7, "ln2"]), ("z_2", z_w[2], cache_all["z", 2])):
        xf = x.reshape(24 * 16, -1).cpu()
        rv, ri = topk_stream_ref(xf, w, k=5)
        _assert_topk_equal(res[name][0], res[name][1], rv, ri, E.encode_ref(xf, w["W_enc"], w["b_enc"], w["b_dec"]))


def test_sweep_checkpoint_and_resume_on_gpu(tmp_path):
    """MaxActSweep.save / load through the HIP engine: interrupted + resumed == uninterrupted."""
    import topk_scan
    from load_model_and_encoders import build_transcoder

    g, w, xs, n_seq, n_pos = gi.g2_inputs()
    tc = build_transcoder(w, 8, DEV)
    x = xs.to(DEV)
    whole = topk_scan.MaxActSweep({"mlp_8": tc}, k=20, distributed=False)
    whole.feed({"mlp_8": x}, 0)
    ev, ei = whole.result()["mlp_8"]
    a = topk_scan.MaxActSweep({"mlp_8": tc}, k=20, distributed=False)
    a.feed({"mlp_8": x[:1500]}, 0)
    a.save(str(tmp_path / "s.pt"))
    b = topk_scan.MaxActSweep({"mlp_8": tc}, k=20, distributed=False)
    b.load(str(tmp_path / "s.pt"))
    b.feed({"mlp_8": x[1500:]}, b.tokens_seen)
    v, i = b.result()["mlp_8"]
    assert torch.equal(v, ev) and torch.equal(i, ei)


def test_forward_decodes_from_the_sparse_form_on_many_rows():
    """``SparseTranscoder.forward`` / ``ZSAE.forward`` on >= 256 rows: the decoder half runs as tsc_dense_to_sparse +
    tsc_decode_sparse on the activations the caller gets back.  Same reconstruction as the dense product (fp32 sums in
    another order), dense fallback for few rows, for a row with more positives than the lists hold and for activations a
    hook has made negative."""
    import load_model_and_encoders as lme
    import tsc_hip

    w = gi.g1_full_weights()
    tc = lme.build_transcoder(w, 0, DEV)
    x = torch.randn(700, 768, generator=torch.Generator().manual_seed(3)).to(DEV)
    with torch.no_grad():
        sae_out, acts = tc(x)[:2]
        dense = acts @ tc.W_dec + tc.b_dec_out
        assert int((acts > 0).sum(1).max()) <= tsc_hip.DeviceDict.DECODE_SPARSE_CAPACITY
        assert tc.device_dict().decode_dense_acts(acts) is not None           # the sparse route was taken ...
        _close(sae_out, dense, atol=1e-5)                                      # ... and agrees with the dense product
        assert torch.equal(tc(x)[0], sae_out)                                  # fixed summation order: reproducible
        few = tc(x[:100])[0]
        assert tc.device_dict().decode_dense_acts(acts[:100]) is None and torch.equal(few, acts[:100] @ tc.W_dec + tc.b_dec_out)
        # the lists are the positives of each row in a fixed order
        m, cap = 300, 64
        feat = torch.empty(m, cap, dtype=torch.int32, device=DEV)
        val = torch.empty(m, cap, device=DEV)
        count = torch.empty(m, dtype=torch.int32, device=DEV)
        a = acts[:m].contiguous()
        a[7, 5] = -1.0
        a[9, 100] = float("nan")
        tsc_hip._check(tsc_hip.lib().tsc_dense_to_sparse(a.data_ptr(), m, a.shape[1], cap, feat.data_ptr(), val.data_ptr(),
                                                         count.data_ptr(), None))
        torch.cuda.synchronize()
        for r in (0, 1, 150, 299):
            pos = (a[r] > 0).nonzero().flatten()
            c = int(count[r])  # (rows 7 and 9 are not among these)
            assert c == pos.numel()
            got = dict(zip(feat[r, :min(c, cap)].tolist(), val[r, :min(c, cap)].tolist()))
            assert len(got) == min(c, cap) and all(float(a[r, f]) == v for f, v in got.items())
        assert int(count[7]) == -1 - int((a[7] > 0).sum()) and int(count[9]) < 0   # not ReLU outputs: flagged, lists skipped
        assert tc.device_dict().decode_dense_acts(a) is None                   # -> the caller multiplies densely
        # z-SAE forward takes the same route
        wz = gi.g1_inputs()[2]
        zs = lme.build_zsae(wz, 0, DEV)
        xz = torch.randn(400, 768, generator=torch.Generator().manual_seed(4)).to(DEV)
        _, rec, za, _, _ = zs(xz)
        _close(rec, za @ zs.W_dec + zs.b_dec, atol=1e-5)
