"""Golden fixture ``g5_callers.npz`` / ``g5_prompts.json``: outputs of the reference's CALLER functions
themselves, run unmodified in the build container (``/root/reference`` must exist):

* ``max_act_analysis.MaxActAnalysis`` -- ``active_examples``, ``get_active_example``, ``num_examples``,
  ``get_top_k_tokens`` (mlp and attn feature)                                     [SURVEY 8a: a5, a6, a7]
* ``autointerpretability.get_feature_scores`` (encode, raw, feature_post modes) and
  ``get_top_k_activating_examples``                                                [a8, a9]
* ``task_evaluation.TaskEvaluation.logit_difference`` on prompts from the reference's
  ``data.ioi_dataset.gen_templated_prompts(template_idex=1)``                      [a15]
* ``circuit_lens.CircuitLens`` lens rankings whose top-10 is NOT cut short by the embed/pos_embed ``break``
  (circuit_lens.py:171-176), ``get_transcoder_contribs`` for every layer, and ``get_top_contribs`` in the one
  situation where the reference's own code returns instead of raising (no active z feature)   [a12, a14]

The reference modules import third-party packages that are absent offline and fetch data at import time;
stub modules stand in for those (``circuitsvis``, ``graphviz``, ``IPython.display``, ``openai_utils``,
``open_web_text`` = the synthetic token matrix, ``transformer_lens``/``jaxtyping`` as in make_golden.py) and the
process-global ``load_model_and_encoders`` caches are pre-seeded with the stand-in models -- the seam the
reference itself provides (load_model_and_encoders.py:10,30-31).  Only inputs-by-seed and OUTPUTS are stored.
"""
from __future__ import annotations

import json
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))  # tests/: replay_model
import make_golden as mg  # noqa: E402  (also puts the package dir on sys.path)

from replay_model import ReplayModel  # noqa: E402
from standins.synthetic_dicts import make_dictionary_weights, weights_fingerprint  # noqa: E402
from standins.synthetic_model import SyntheticHookedTransformer  # noqa: E402


def _mod(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


class _Inert:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, n):
        return lambda *a, **k: None


def main():
    mg._install_shims()
    _mod("circuitsvis")
    _mod("graphviz", Digraph=_Inert)
    ip = _mod("IPython")
    ip.display = _mod("IPython.display", HTML=_Inert, display=lambda *a, **k: None)
    _mod("openai_utils", gen_openai_completion=lambda *a, **k: "")
    sys.modules["transformer_lens"].ActivationCache = dict

    g2 = np.load(os.path.join(HERE, "g2_topk.npz"))
    n_seq, n_pos = 32, 128
    wk = make_dictionary_weights(768, 1536, seed=21, b_enc=(-1.5, -3.6), transcoder=True)
    wz = make_dictionary_weights(768, 1536, seed=22, b_enc=(-1.5, -3.6), transcoder=False)
    xs = torch.randn(n_seq * n_pos, 768, generator=torch.Generator().manual_seed(202))
    xs[n_pos * 5: n_pos * 5 + 8] = xs[0:8]
    assert abs(weights_fingerprint(wk) - float(g2["fp"])) < 1e-6
    replay = ReplayModel(xs, n_seq, n_pos)
    tokens = replay.token_matrix()
    _mod("open_web_text", open_web_text_tokens=tokens)

    sys.path.insert(0, mg.REF)
    os.chdir(mg.REF)
    torch.set_grad_enabled(False)
    import load_model_and_encoders as lme  # reference
    from mlp_transcoder import SparseTranscoder  # reference
    from z_sae import ZSAE  # reference

    lme.model_cache = replay  # data/greater_than_dataset.py:55 calls get_model() at import
    import autointerpretability as ref_ai  # reference
    import circuit_lens as ref_cl  # reference
    import max_act_analysis as ref_maa  # reference
    import task_evaluation as ref_te  # reference
    from data.ioi_dataset import gen_templated_prompts  # reference

    tck = mg._ref_transcoder(SparseTranscoder, wk, 0)
    zsk = mg._ref_zsae(ZSAE, wz, 0)
    lme.model_encoder_cache = (replay, [zsk], [tck])
    g5 = {"fp_tc": weights_fingerprint(wk), "fp_zs": weights_fingerprint(wz)}

    # ---------------------------------------------------------------- a5/a6/a7: the real MaxActAnalysis
    for f in (3, 700, 1535):
        ma = ref_maa.MaxActAnalysis("mlp", 0, f, batch_size=8)
        seq_pos, vals, scores = ma.active_examples
        g5[f"maa_seq_pos_{f}"] = seq_pos.numpy().astype(np.int32)
        g5[f"maa_vals_{f}"] = vals.numpy()
        g5[f"maa_num_examples_{f}"] = ma.num_examples
        # the restated fixture (g2) and the real class agree (ties may be ordered differently by the unstable argsort)
        assert np.array_equal(vals.numpy(), g2[f"a5_vals_{f}"]), f
        if seq_pos.shape[0]:
            s, p, v = ma.get_active_example(0)
            g5[f"maa_example0_{f}"] = np.array([int(s), int(p)], np.int32)
    ma = ref_maa.MaxActAnalysis("attn", 0, 11, batch_size=8)
    seq_pos, vals, _ = ma.active_examples
    g5["maa_attn_seq_pos_11"] = seq_pos.numpy().astype(np.int32)
    g5["maa_attn_vals_11"] = vals.numpy()
    for kind, f in (("mlp", 3), ("mlp", 700), ("attn", 11), ("attn", 1000)):
        toks = ref_maa.MaxActAnalysis(kind, 0, f, batch_size=8).get_top_k_tokens(k=5)
        g5[f"top_tokens_{kind}_{f}"] = np.array([int(t) for t in toks], np.int32)  # ReplayModel.to_str_tokens = str(id)

    # ---------------------------------------------------------------- a8/a9: the real get_feature_scores
    feats = [3, 17, 700, 1535]
    sc = ref_ai.get_feature_scores(replay, tck, tokens, feats, batch_size=8)
    assert sc.dtype == np.float16 and np.array_equal(sc, g2["gfs_scores"])
    raw = ref_ai.get_feature_scores(replay, tck, tokens, feats, batch_size=8, use_raw_scores=True)
    assert np.array_equal(raw, g2["gfs_raw"])
    post = torch.randn(768, 3, generator=torch.Generator().manual_seed(77)) * 0.05
    g5["gfs_post_raw"] = ref_ai.get_feature_scores(replay, tck, tokens, [5, 6, 7], batch_size=8, use_raw_scores=True,
                                                  feature_post=post.clone())
    g5["gfs_post_decoder"] = ref_ai.get_feature_scores(replay, tck, tokens, [5, 6, 7], batch_size=8, use_raw_scores=True,
                                                      use_decoder=True, feature_post=post.clone())
    g5["gfs_zsae"] = ref_ai.get_feature_scores(replay, zsk, tokens, [11, 1000], batch_size=8)
    g5["gfs_ignore_eot"] = ref_ai.get_feature_scores(replay, tck, tokens[:8], feats, batch_size=8, ignore_endoftext=True)
    assert np.array_equal(g5["gfs_ignore_eot"], sc[:8])  # the reference's eos comparison never matches (:244-247)
    for j, f in enumerate(feats[:3]):
        strs, per_seq, pos = ref_ai.get_top_k_activating_examples(sc[:, j, :], tokens, replay, k=5)
        g5[f"gtk_seq_{f}"] = np.array([int(s[1]) for s in strs], np.int32)  # the replay tokens carry the sequence id in column 1
        g5[f"gtk_pos_{f}"] = np.asarray(pos).astype(np.int32)
        g5[f"gtk_per_seq_{f}"] = np.array(per_seq, np.float16)

    # ---------------------------------------------------------------- a15: the real TaskEvaluation.logit_difference
    n_layers = 3
    model = SyntheticHookedTransformer(n_layers=n_layers, d_vocab=1000, n_ctx=32, seed=7)
    z_saes, tcs = [], []
    for l in range(n_layers):
        z_saes.append(mg._ref_zsae(ZSAE, make_dictionary_weights(768, 1536, seed=300 + l, b_enc=-1.3, transcoder=False), l))
        tcs.append(mg._ref_transcoder(SparseTranscoder, make_dictionary_weights(768, 1536, seed=400 + l, b_enc=-2.0, transcoder=True), l))
    lme.model_cache = model
    lme.model_encoder_cache = (model, z_saes, tcs)
    random.seed(0)
    prompts = gen_templated_prompts(template_idex=1, N=8)  # task_evaluation_experiment.py:31
    te = ref_te.TaskEvaluation(prompts, None, no_mean_cache=True)
    g5["te_tokens"] = te.tokens.numpy().astype(np.int32)
    for i in range(len(prompts)):
        g5[f"te_logit_diff_{i}"] = te.logit_difference(i).numpy()
    json.dump(prompts, open(os.path.join(HERE, "g5_prompts.json"), "w"), indent=1)

    # ---------------------------------------------------------------- a12: rankings not cut by the embed break
    cl = ref_cl.CircuitLens(mg_prompt())
    seq = cl.process_seq_index(-3)

    def _full_ranking(runs):  # ten entries, none of them the embed / pos_embed rows, several real features
        comps = [r.component for r, _ in runs]
        return (len(runs) == 10 and "embed" not in comps and "pos_embed" not in comps
                and sum(c in ("z_feature", "mlp_feature") for c in comps) >= 3)

    found = None
    for s2 in [seq] + [p for p in range(2, cl.n_tokens - 1) if p != seq]:
        for tok in range(1, 1000):
            runs, _ = cl.get_unembed_lens(tok, s2, visualize=False, k=10)
            if _full_ranking(runs):
                found = (s2, tok)
                break
        if found:
            break
    assert found is not None
    g5["unembed2_seq_tok"] = np.array(found, np.int32)
    g5["unembed2_top_ids"] = np.array([[mg._c(r.component), r.run_data["layer"], r.run_data["feature"]] for r, _ in runs], np.int32)
    g5["unembed2_top_vals"] = np.array([v for _, v in runs], np.float32)
    found = None
    for s2 in [seq] + [p for p in range(2, cl.n_tokens - 1) if p != seq]:
        for layer in (2, 1):
            acts = tcs[layer](cl.cache["normalized", layer, "ln2"][:, s2])[1][0]
            for feat in torch.nonzero(acts).flatten().tolist():
                runs, _ = cl.get_mlp_feature_lens_at_seq(layer, s2, feat, visualize=False, k=10)
                if _full_ranking(runs):
                    found = (layer, s2, feat)
                    break
            if found:
                break
        if found:
            break
    assert found is not None
    g5["mlp_lens2_layer_seq_feature"] = np.array(found, np.int32)
    g5["mlp_lens2_top_ids"] = np.array([[mg._c(r.component), r.run_data["layer"], r.run_data["feature"]] for r, _ in runs], np.int32)
    g5["mlp_lens2_top_vals"] = np.array([v for _, v in runs], np.float32)

    # ---------------------------------------------------------------- a14: contribs per layer; get_top_contribs
    proj = torch.randn(768, 8, generator=torch.Generator().manual_seed(303))
    tok = int(cl.tokens[0, seq + 1])
    direction = model.W_U[:, tok] - model.W_U[:, int(cl.tokens[0, 1])]
    for l in range(n_layers):
        c = cl.get_transcoder_contribs(l, seq, direction.clone(), k=5)
        g5[f"tc_contribs_idx_l{l}"] = np.array([x[3] for x in c], np.int32)
        g5[f"tc_contribs_val_l{l}"] = np.array([x[4] for x in c], np.float32)
    # the reference's get_top_contribs returns only when no layer <= `layer` has an active z feature (its attention
    # branch unpacks a 2-tuple into three names, circuit_lens.py:1293-1302): silence the z-SAEs and run it for real
    dead_z = []
    for l in range(n_layers):
        w = make_dictionary_weights(768, 1536, seed=300 + l, b_enc=-1.3, transcoder=False)
        w["b_enc"] = torch.full_like(w["b_enc"], -100.0)
        dead_z.append(mg._ref_zsae(ZSAE, w, l))
    lme.model_encoder_cache = (model, dead_z, tcs)
    cl2 = ref_cl.CircuitLens(mg_prompt())
    import contextlib
    import io

    with contextlib.redirect_stdout(io.StringIO()):
        top = cl2.get_top_contribs(direction.clone(), 2, seq, k=5)
    g5["top_contribs_layer_idx"] = np.array([[x[1], x[3]] for x in top], np.int32)
    g5["top_contribs_val"] = np.array([x[4] for x in top], np.float32)
    g5["top_contribs_vec_proj"] = torch.stack([x[0] for x in top]).matmul(proj).numpy()
    np.savez_compressed(os.path.join(HERE, "g5_callers.npz"), **g5)
    print("g5_callers.npz", os.path.getsize(os.path.join(HERE, "g5_callers.npz")), "keys", len(g5))


def mg_prompt():
    return "Mary and Jeff went to the store, and Mary gave an apple to Jeff"  # test_circuit_lens.py:13


if __name__ == "__main__":
    main()
