"""Pins the CPU oracle (``oracle/``) against the golden vectors produced by running the
reference itself (``tests/golden/make_golden.py``).  CPU only."""
import numpy as np
import torch

import golden_inputs as gi
from oracle import encode_ref as E
from oracle import lens_ref as L
from oracle import topk_ref as T

RTOL, ATOL = 1e-4, 1e-5  # north star: activations within 1e-4 relative (fp32)


def _close(a, b, rtol=RTOL, atol=ATOL):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.allclose(a, b, rtol=rtol, atol=atol), float((a - b).abs().max())


def test_encode_and_forward_match_reference():
    g, wt, wz, x = gi.g1_inputs()
    _close(E.encode_ref(x, wt["W_enc"], wt["b_enc"], wt["b_dec"]).numpy(), g["tc_encode"])
    recon, acts = E.forward_ref(x, wt)
    _close(recon.numpy(), g["tc_sae_out"])
    _close(acts.numpy(), g["tc_encode"])
    zrecon, zacts = E.forward_ref(x, wz)
    _close(zacts.numpy(), g["zs_encode"])
    _close(zrecon.numpy(), g["zs_recon"])
    _close(E.encode_ref(x.view(4, 16, 768), wt["W_enc"], wt["b_enc"], wt["b_dec"]).numpy(), g["tc_encode_3d"])


def test_full_width_row_slice():
    g, _, _, x = gi.g1_inputs()
    w = gi.g1_full_weights()
    _close(E.encode_ref(x[:4], w["W_enc"], w["b_enc"], w["b_dec"]).numpy(), g["full_encode"])


def test_running_topk_matches_reference_argsort():
    g, w, xs, n_seq, n_pos = gi.g2_inputs()
    vals, idx = T.topk_stream_ref(xs, w, k=20, batch_tokens=8 * n_pos)
    _close(vals.numpy(), g["vals"])
    assert np.array_equal(idx.numpy(), g["idx"].astype(np.int64))  # bit-exact indices
    # batch size must not matter
    v2, i2 = T.topk_stream_ref(xs, w, k=20, batch_tokens=1000)
    assert torch.equal(i2, idx) and torch.equal(v2, vals)
    # short lists are padded with (0, -1)
    short = g["n_active"] < 20
    assert short.any()
    assert (idx.numpy()[short][:, -1] == -1).all() and (vals.numpy()[short][:, -1] == 0).all()


def test_topk_merge_of_shards_equals_single_stream():
    g, w, xs, n_seq, n_pos = gi.g2_inputs()
    half = xs.shape[0] // 2
    a = T.topk_stream_ref(xs[:half], w, k=20, token_id_base=0)
    b = T.topk_stream_ref(xs[half:], w, k=20, token_id_base=half)
    v, i = T.topk_merge_ref([b, a], k=20)  # order of shards must not matter
    assert np.array_equal(i.numpy(), g["idx"].astype(np.int64))
    _close(v.numpy(), g["vals"])


def test_active_examples_tuple():
    g, w, xs, n_seq, n_pos = gi.g2_inputs()
    acts = E.encode_ref(xs, w["W_enc"], w["b_enc"], w["b_dec"])
    for f in (3, 700, 1535):
        seq_pos, vals, scores = T.active_examples_ref(acts[:, f].reshape(n_seq, n_pos))
        assert np.array_equal(seq_pos.numpy(), g[f"a5_seq_pos_{f}"].astype(np.int64))
        _close(vals.numpy(), g[f"a5_vals_{f}"])
        assert scores.shape == (n_seq, n_pos)


def test_feature_scores_layout_and_fp16():
    g, w, xs, n_seq, n_pos = gi.g2_inputs()
    feats = g["gfs_feats"].tolist()
    sc = T.feature_scores_ref(xs, w, feats, n_pos, batch_seqs=8)
    assert sc.dtype == np.float16 and sc.shape == (n_seq, len(feats), n_pos)
    _close(sc.astype(np.float32), g["gfs_scores"].astype(np.float32), rtol=2e-3, atol=1e-4)
    raw = T.feature_scores_ref(xs, w, feats, n_pos, batch_seqs=8, use_raw_scores=True)
    _close(raw.astype(np.float32), g["gfs_raw"].astype(np.float32), rtol=2e-3, atol=2e-3)
    top_scores, _, _ = T.top_k_activating_examples_ref(g["gfs_scores"][:, 2, :], k=5)
    assert np.array_equal(top_scores, g["gtk_scores"])


def _lens_setup():
    g3, g4, model, z_w, t_w = gi.lens_inputs()
    tokens = model.to_tokens(gi.PROMPT)
    assert np.array_equal(tokens.numpy(), g3["tokens"])
    logits, cache = model.run_with_cache(tokens)
    assert abs(float(logits.double().sum()) - float(g3["logits_sum"])) < 1e-6 * abs(float(g3["logits_sum"])) + 1e-6
    return g3, g4, model, z_w, t_w, tokens, cache


def test_active_features_decomposition():
    g3, g4, model, z_w, t_w, tokens, cache = _lens_setup()
    proj = gi.proj8()
    for pos in (12, 5):
        vectors, values, features, keys = L.active_features_ref(model, cache, z_w, t_w, pos)
        tag = f"p{pos}"
        assert [[k["attn"], k["mlp"]] for k in keys] == g3[f"{tag}_keys"].tolist()
        assert np.array_equal(features.numpy(), g3[f"{tag}_features"].astype(np.int64))
        _close(values.numpy(), g3[f"{tag}_values"])
        _close((vectors @ proj).numpy(), g3[f"{tag}_vec_proj"], atol=1e-4)
        _close(vectors.sum(0).numpy(), g3[f"{tag}_vec_sum"], atol=1e-4)


def test_lens_scores():
    g3, g4, model, z_w, t_w, tokens, cache = _lens_setup()
    seq = 12
    vectors, values, features, keys = L.active_features_ref(model, cache, z_w, t_w, seq)
    tok = int(tokens[0, seq + 1])
    _close(L.unembed_lens_raw_ref(vectors, model.W_U, tok).numpy(), g4["unembed_raw"], atol=1e-4)
    layer, zfeat = 2, int(g4["z_feature"])
    _close(L.head_seq_acts_ref(model, cache, z_w, layer, seq, zfeat).numpy(), g4["head_seq_acts"], atol=1e-4)
    _close(L.v_lens_raw_ref(vectors, keys, model, z_w, layer, 4, zfeat).numpy(), g4["v_lens"], atol=1e-4)
    _close(L.q_lens_raw_ref(vectors, keys, model, cache, layer, 4, 2).numpy(), g4["q_lens"], atol=1e-4)
    # the K lens decomposes the SOURCE position (circuit_lens.py:1095)
    v2, _, _, keys2 = L.active_features_ref(model, cache, z_w, t_w, 2)
    _close(L.k_lens_raw_ref(v2, keys2, model, cache, layer, 4, seq).numpy(), g4["k_lens"], atol=1e-4)
    dirs = torch.from_numpy(g4["dirs"])
    proj = gi.proj8()
    for l in range(3):
        x_seq = cache["normalized", l, "ln2"]
        for d in range(2):
            pb, acts = L.transcoder_ixg_ref(t_w[l], x_seq, seq, dirs[d])
            _close(pb.numpy(), g4[f"ixg_d{d}_l{l}"])
            if d == 0:
                _close(acts.numpy(), g4[f"ixg_acts_l{l}"])
                # same scores from the sparse form (north-star formula)
                nz = acts.nonzero().flatten()
                sc = L.contribution_scores_ref(acts[nz], nz, t_w[l]["W_dec"], dirs)
                _close(sc[:, 0].numpy(), g4[f"ixg_d0_l{l}"][nz.numpy()])
        _close((L.attn_head_contribs_ref(model, cache, z_w[l], l, seq) @ proj).numpy(),
               g4[f"attn_contribs_proj_l{l}"], atol=1e-4)


# ----------------------------------------------------------------------------- g5: the reference's caller functions
def test_oracle_matches_the_real_max_act_analysis_and_feature_scores():
    from oracle import callers_ref as C
    from replay_model import ReplayModel

    g5, w, wz, xs, n_seq, n_pos, prompts = gi.g5_inputs()
    acts = E.encode_ref(xs, w["W_enc"], w["b_enc"], w["b_dec"])
    for f in (3, 700, 1535):
        seq_pos, vals, _ = T.active_examples_ref(acts[:, f].reshape(n_seq, n_pos))
        assert gi.same_ranking(seq_pos.numpy(), vals.numpy(), g5[f"maa_seq_pos_{f}"], g5[f"maa_vals_{f}"])
        assert seq_pos.shape[0] == int(g5[f"maa_num_examples_{f}"])
    zacts = E.encode_ref(xs, wz["W_enc"], wz["b_enc"], wz["b_dec"])
    seq_pos, vals, _ = T.active_examples_ref(zacts[:, 11].reshape(n_seq, n_pos))
    assert gi.same_ranking(seq_pos.numpy(), vals.numpy(), g5["maa_attn_seq_pos_11"], g5["maa_attn_vals_11"])
    model = ReplayModel(xs, n_seq, n_pos)
    for kind, f, ww in (("mlp", 3, w), ("mlp", 700, w), ("attn", 11, wz), ("attn", 1000, wz)):
        _, ids = C.top_k_tokens_ref(ww, f, model.W_U, k=5, W_O_layer=model.W_O[0] if kind == "attn" else None)
        assert ids.tolist() == g5[f"top_tokens_{kind}_{f}"].tolist()
    post = torch.randn(768, 3, generator=torch.Generator().manual_seed(77)) * 0.05
    for key, dec in (("gfs_post_raw", False), ("gfs_post_decoder", True)):
        got = C.feature_scores_post_ref(xs, w, [5, 6, 7], post, n_pos, use_decoder=dec)
        _close(got.astype(np.float32), g5[key].astype(np.float32), rtol=2e-3, atol=2e-3)
    zs = T.feature_scores_ref(xs, wz, [11, 1000], n_pos, batch_seqs=8)
    _close(zs.astype(np.float32), g5["gfs_zsae"].astype(np.float32), rtol=2e-3, atol=1e-4)
    g2 = gi.load("g2_topk.npz")
    for j, f in enumerate((3, 17, 700)):
        sc, seq, pos = T.top_k_activating_examples_ref(g2["gfs_scores"][:, j, :], k=5)
        ref_sc = g2["gfs_scores"][g5[f"gtk_seq_{f}"], j, g5[f"gtk_pos_{f}"]]
        assert gi.same_ranking(np.stack([seq, pos], 1), sc, np.stack([g5[f"gtk_seq_{f}"], g5[f"gtk_pos_{f}"]], 1), ref_sc)
        assert np.array_equal(g2["gfs_scores"][g5[f"gtk_seq_{f}"], j, :], g5[f"gtk_per_seq_{f}"])


def test_oracle_matches_the_real_logit_difference_and_contribs():
    from oracle import callers_ref as C

    g5, w, wz, xs, n_seq, n_pos, prompts = gi.g5_inputs()
    g3, g4, model, z_w, t_w, tokens, cache = _lens_setup()
    assert np.array_equal(model.to_tokens([p["text"] for p in prompts]).numpy(), g5["te_tokens"])
    for i, p in enumerate(prompts):
        d = C.logit_difference_ref(model, p["correct"], p["counter"])
        assert np.array_equal(d.numpy(), g5[f"te_logit_diff_{i}"])
        assert float(d[:-1].abs().max()) == 0.0 and float(d[-1].abs().max()) > 0  # the BOS rows cancel
    seq = 12
    tok = int(tokens[0, seq + 1])
    direction = model.W_U[:, tok] - model.W_U[:, int(tokens[0, 1])]
    for l in range(3):
        c = C.transcoder_contribs_ref(cache, t_w[l], l, seq, direction, k=5)
        assert [x[3] for x in c] == g5[f"tc_contribs_idx_l{l}"].tolist()
        _close(np.array([x[4] for x in c]), g5[f"tc_contribs_val_l{l}"])
    top = C.top_contribs_ref(cache, t_w, 2, seq, direction, k=5)
    assert [[x[1], x[3]] for x in top] == g5["top_contribs_layer_idx"].tolist()
    _close(np.array([x[4] for x in top]), g5["top_contribs_val"])
    _close((torch.stack([x[0] for x in top]) @ gi.proj8()).numpy(), g5["top_contribs_vec_proj"], atol=1e-4)


def test_oracle_lens_rankings_not_cut_by_the_embed_break():
    from circuit_lens import ActiveFeatures

    g5, *_ = gi.g5_inputs()
    g3, g4, model, z_w, t_w, tokens, cache = _lens_setup()
    s2, tok = g5["unembed2_seq_tok"].tolist()
    vectors, values, features, keys = L.active_features_ref(model, cache, z_w, t_w, s2)
    af = ActiveFeatures(vectors=vectors, values=values, features=features, keys=keys)
    top = af.get_top_k_features(L.normalize_ref(L.unembed_lens_raw_ref(vectors, model.W_U, tok)), k=10)
    assert len(top) == 10
    assert [[gi.COMPONENT_IDS.index(c), l, f] for c, l, f, _ in top] == g5["unembed2_top_ids"].tolist()
    _close(np.array([v for *_, v in top]), g5["unembed2_top_vals"], atol=1e-5)
    layer, s3, feat = g5["mlp_lens2_layer_seq_feature"].tolist()
    vectors, values, features, keys = L.active_features_ref(model, cache, z_w, t_w, s3)
    af = ActiveFeatures(vectors=vectors, values=values, features=features, keys=keys)
    top = af.get_top_k_features(L.normalize_ref(L.mlp_feature_lens_raw_ref(vectors, keys, t_w, layer, feat)), k=10)
    assert [[gi.COMPONENT_IDS.index(c), l, f] for c, l, f, _ in top] == g5["mlp_lens2_top_ids"].tolist()
    _close(np.array([v for *_, v in top]), g5["mlp_lens2_top_vals"], atol=1e-5)
