"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE ITSELF.

Run only in the build container (``/root/reference`` must exist; it does not on the GPU
box):  ``python tests/golden/make_golden.py``.

The unmodified reference modules ``mlp_transcoder``, ``z_sae`` and ``circuit_lens`` are
imported from ``/root/reference/src`` with stub modules standing in for the absent
third-party packages ``jaxtyping`` / ``transformer_lens`` / ``plotly`` helpers (SURVEY.md
section 8c).  They are driven with seeded synthetic weights (``synthetic_dicts.py``) and the
seeded stand-in model (``synthetic_model.py``); only inputs-by-seed and OUTPUTS are stored.
``max_act_analysis.py`` / ``autointerpretability.py`` fetch datasets at import time, so
their few arithmetic lines are applied verbatim-in-meaning to the reference encoders'
outputs (each line cited below).
"""
from __future__ import annotations

import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
PKG = os.path.join(REPO, "transcoder-sae-circuits_amd")
REF = "/root/reference/src"

sys.dont_write_bytecode = True
sys.path.insert(0, PKG)

from standins.synthetic_dicts import make_dictionary_weights, weights_fingerprint  # noqa: E402
from standins.synthetic_model import SyntheticHookedTransformer  # noqa: E402


# ----------------------------------------------------------------------------- shims
def _install_shims():
    class _Sub:
        def __class_getitem__(cls, item):
            return cls

    jt = types.ModuleType("jaxtyping")
    jt.Float = _Sub
    jt.Int = _Sub
    sys.modules["jaxtyping"] = jt

    class HookPoint(nn.Module):
        def forward(self, x):
            return x

    class HookedRootModule(nn.Module):
        def setup(self):
            pass

    tl = types.ModuleType("transformer_lens")
    hp = types.ModuleType("transformer_lens.hook_points")
    hp.HookPoint = HookPoint
    hp.HookedRootModule = HookedRootModule
    ut = types.ModuleType("transformer_lens.utils")
    ut.to_numpy = lambda t: t.detach().cpu().numpy()
    tl.hook_points = hp
    tl.utils = ut
    tl.HookedTransformer = object
    sys.modules["transformer_lens"] = tl
    sys.modules["transformer_lens.hook_points"] = hp
    sys.modules["transformer_lens.utils"] = ut
    pu = types.ModuleType("plotly_utils")
    pu.imshow = lambda *a, **k: None
    sys.modules["plotly_utils"] = pu


def _ref_transcoder(SparseTranscoder, w, layer):
    cfg = SimpleNamespace(
        d_in=w["W_enc"].shape[0], d_sae=w["W_enc"].shape[1], l1_coefficient=1e-3,
        dtype=torch.float32, device="cpu", is_transcoder=True, d_out=w["W_dec"].shape[1],
        is_sparse_connection=False, use_ghost_grads=False,
        hook_point=f"blocks.{layer}.ln2.hook_normalized", hook_point_layer=layer,
    )
    m = SparseTranscoder(cfg)
    with torch.no_grad():
        for k, v in w.items():
            getattr(m, k).data.copy_(v)
    return m.eval()


def _ref_zsae(ZSAE, w, layer):
    m = ZSAE({"dict_size": w["W_enc"].shape[1], "l1_coeff": 1.0, "enc_dtype": "fp32", "seed": 0,
              "act_size": w["W_enc"].shape[0], "device": "cpu", "layer": layer})
    with torch.no_grad():
        for k, v in w.items():
            getattr(m, k).data.copy_(v)
    return m.eval()


def main():
    _install_shims()
    sys.path.insert(0, REF)
    os.chdir(REF)
    from mlp_transcoder import SparseTranscoder  # reference
    from z_sae import ZSAE  # reference
    import load_model_and_encoders as lme  # reference
    import circuit_lens as ref_cl  # reference

    torch.set_grad_enabled(False)
    torch.manual_seed(0)

    # ------------------------------------------------------------------ G1 encode/forward
    g1 = {}
    wt = make_dictionary_weights(768, 1536, seed=11, b_enc=-0.30, transcoder=True)
    wz = make_dictionary_weights(768, 1536, seed=12, b_enc=-0.30, transcoder=False)
    tc = _ref_transcoder(SparseTranscoder, wt, 8)
    zs = _ref_zsae(ZSAE, wz, 8)
    x = torch.randn(64, 768, generator=torch.Generator().manual_seed(101))
    g1["fp_tc"] = weights_fingerprint(wt)
    g1["fp_zs"] = weights_fingerprint(wz)
    g1["x_sum"] = float(x.double().sum())
    g1["tc_encode"] = tc.encode(x).numpy()
    tcf = tc(x)
    g1["tc_sae_out"] = tcf[0].numpy()
    assert torch.equal(tcf[1], torch.from_numpy(g1["tc_encode"]))
    g1["zs_encode"] = zs.encode(x).numpy()
    zf = zs(x)
    g1["zs_recon"] = zf[1].numpy()
    assert torch.equal(zf[2], torch.from_numpy(g1["zs_encode"]))
    # 3-D batch shape ([b, pos, d]) as max_activating.py:50,58 feeds it
    g1["tc_encode_3d"] = tc.encode(x.view(4, 16, 768)).numpy()
    # one full-width row slice
    wfull = make_dictionary_weights(768, 24576, seed=13, b_enc=-0.70, transcoder=True)
    tcfull = _ref_transcoder(SparseTranscoder, wfull, 8)
    g1["fp_full"] = weights_fingerprint(wfull)
    g1["full_encode"] = tcfull.encode(x[:4]).numpy()
    np.savez_compressed(os.path.join(HERE, "g1_encode.npz"), **g1)

    # ------------------------------------------------------------------ G2 per-feature top-k
    n_seq, n_pos, k = 32, 128, 20
    wk = make_dictionary_weights(768, 1536, seed=21, b_enc=(-1.5, -3.6), transcoder=True)
    tck = _ref_transcoder(SparseTranscoder, wk, 8)
    xs = torch.randn(n_seq * n_pos, 768, generator=torch.Generator().manual_seed(202))
    # duplicate a block of rows so exact ties occur (BOS-like repeated tokens)
    xs[n_pos * 5: n_pos * 5 + 8] = xs[0:8]
    batch = 8 * n_pos
    cols = []
    for i in range(0, xs.shape[0], batch):  # max_act_analysis.py:215-239 (batched encode)
        cols.append(tck.encode(xs[i:i + batch]))
    acts = torch.cat(cols, 0)  # [tokens, d_sae]
    vals = np.zeros((1536, k), np.float32)
    idx = np.full((1536, k), -1, np.int64)
    n_active = np.zeros((1536,), np.int64)
    for f in range(1536):
        score_flat = acts[:, f]  # max_act_analysis.py:232,243-245
        num_active = int((score_flat > 0).sum().item())  # :246
        # :248 uses argsort(descending=True) (unstable); a stable sort gives the
        # tie order "smaller token id first" which the build defines as canonical.
        order = torch.sort(score_flat, descending=True, stable=True).indices[:num_active]
        m = min(k, num_active)
        vals[f, :m] = score_flat[order[:m]].numpy()
        idx[f, :m] = order[:m].numpy()
        n_active[f] = num_active
    g2 = {"fp": weights_fingerprint(wk), "x_sum": float(xs.double().sum()),
          "vals": vals, "idx": idx.astype(np.int32), "n_active": n_active.astype(np.int32)}
    for f in (3, 700, 1535):  # a5 tuples, max_act_analysis.py:243-253
        scores = acts[:, f].reshape(n_seq, n_pos)
        score_flat = scores.flatten()
        num_active = int((score_flat > 0).sum().item())
        indices = torch.sort(score_flat, descending=True, stable=True).indices[:num_active]
        seq_pos = torch.stack(torch.unravel_index(indices, scores.shape)).T
        g2[f"a5_seq_pos_{f}"] = seq_pos.numpy().astype(np.int32)
        g2[f"a5_vals_{f}"] = score_flat[indices].numpy()
    # get_feature_scores layout + fp16 rounding (autointerpretability.py:240-257) for 4 features
    feats = [3, 17, 700, 1535]
    sc = acts[:, feats].reshape(n_seq, n_pos, len(feats)).permute(0, 2, 1).numpy().astype(np.float16)
    g2["gfs_feats"] = np.array(feats)
    g2["gfs_scores"] = sc
    # raw-score mode (autointerpretability.py:226-238)
    fp = tck.W_enc[:, feats]
    bias = tck.b_enc[feats] - (tck.b_dec @ fp)
    raw = (xs @ fp) + bias
    g2["gfs_raw"] = raw.reshape(n_seq, n_pos, len(feats)).permute(0, 2, 1).numpy().astype(np.float16)
    # get_top_k_activating_examples index part (autointerpretability.py:261-266), k=5, feature 700
    fs = sc[:, 2, :]
    flat = fs.flatten()
    top = np.argsort(flat, kind="stable")[-5:][::-1]
    g2["gtk_scores"] = flat[top]
    np.savez_compressed(os.path.join(HERE, "g2_topk.npz"), **g2)

    # ------------------------------------------------------------------ G3/G4 CircuitLens
    n_layers = 3
    model = SyntheticHookedTransformer(n_layers=n_layers, d_vocab=1000, n_ctx=32, seed=7)
    z_saes, tcs, fps = [], [], []
    for l in range(n_layers):
        w_z = make_dictionary_weights(768, 1536, seed=300 + l, b_enc=-1.3, transcoder=False)
        w_t = make_dictionary_weights(768, 1536, seed=400 + l, b_enc=-2.0, transcoder=True)
        fps += [weights_fingerprint(w_z), weights_fingerprint(w_t)]
        z_saes.append(_ref_zsae(ZSAE, w_z, l))
        tcs.append(_ref_transcoder(SparseTranscoder, w_t, l))
    lme.model_encoder_cache = (model, z_saes, tcs)  # the seam, load_model_and_encoders.py:10,30-31
    prompt = "Mary and Jeff went to the store, and Mary gave an apple to Jeff"  # test_circuit_lens.py:13
    cl = ref_cl.CircuitLens(prompt)
    g3 = {"fp": np.array(fps), "tokens": cl.tokens.numpy(), "logits_sum": float(cl.logits.double().sum())}
    proj = torch.randn(768, 8, generator=torch.Generator().manual_seed(303))
    for pos in (-3, 5):
        af = cl.get_active_features(pos)
        tag = f"p{pos % cl.n_tokens}"
        g3[f"{tag}_values"] = af.values.numpy()
        g3[f"{tag}_features"] = af.features.numpy().astype(np.int32)
        g3[f"{tag}_keys"] = np.array([[kk["attn"], kk["mlp"]] for kk in af.keys], np.int32)
        g3[f"{tag}_vec_proj"] = (af.vectors @ proj).numpy()
        g3[f"{tag}_vec_sum"] = af.vectors.sum(0).numpy()
    np.savez_compressed(os.path.join(HERE, "g3_active_features.npz"), **g3)

    g4 = {}
    seq = cl.process_seq_index(-3)
    af = cl.get_active_features(seq)
    tok = int(cl.tokens[0, seq + 1])
    import einops
    raw_unembed = einops.einsum(af.vectors, model.W_U[:, tok], "comp d_model, d_model -> comp")
    g4["unembed_raw"] = raw_unembed.numpy()
    runs, _ = cl.get_unembed_lens(tok, seq, visualize=False, k=10)
    g4["unembed_top_ids"] = np.array([[_c(r.component), r.run_data["layer"], r.run_data["feature"]] for r, _ in runs], np.int32)
    g4["unembed_top_vals"] = np.array([v for _, v in runs], np.float32)
    # get_top_k_features on its own (circuit_lens.py:161-238): with embed/pos_embed masked the
    # early "break" at rows 0/1 is not hit and every component branch appears
    masked = raw_unembed.clone()
    masked[:2] = -1e9
    tk = af.get_top_k_features(masked, k=40)
    g4["topk_masked_ids"] = np.array([[_c(c), l, f] for c, l, f, _ in tk], np.int32)
    g4["topk_masked_vals"] = np.array([v for *_, v in tk], np.float32)
    layer = 2
    mlp_acts = tcs[layer](cl.cache["normalized", layer, "ln2"][:, seq])[1][0]
    feat = int(mlp_acts.argmax())
    g4["mlp_lens_feature"] = feat
    runs, _ = cl.get_mlp_feature_lens_at_seq(layer, seq, feat, visualize=False, k=10)
    g4["mlp_lens_top_ids"] = np.array([[_c(r.component), r.run_data["layer"], r.run_data["feature"]] for r, _ in runs], np.int32)
    g4["mlp_lens_top_vals"] = np.array([v for _, v in runs], np.float32)
    zfeat = int(z_saes[layer](cl.cache["z", layer][0, seq].reshape(-1))[2].argmax())
    g4["z_feature"] = zfeat
    g4["head_seq_acts"] = cl.get_head_seq_activations_for_z_feature(layer, seq, zfeat).numpy()
    g4["v_lens"] = cl.get_v_lens_activations(layer, 4, seq, zfeat).numpy()
    g4["q_lens"] = cl.get_q_lens_activations(layer, 4, 2, seq).numpy()
    g4["k_lens"] = cl.get_k_lens_activations(layer, 4, 2, seq).numpy()
    dirs = torch.stack([model.W_U[:, tok] - model.W_U[:, int(cl.tokens[0, 1])],  # logit-diff, task_evaluation.py:141-151
                        torch.randn(768, generator=torch.Generator().manual_seed(404))])
    g4["dirs"] = dirs.numpy()
    for d in range(2):
        for l in range(n_layers):
            pb, fa = cl.get_transcoder_ixg(tcs[l], l, seq, dirs[d].clone())
            g4[f"ixg_d{d}_l{l}"] = pb.numpy()
            if d == 0:
                g4[f"ixg_acts_l{l}"] = fa.numpy()
    for l in range(n_layers):
        zr = cl.get_attn_head_contribs(l, seq, dirs[0])
        g4[f"attn_contribs_proj_l{l}"] = (zr @ proj).numpy()
    tc_contribs = cl.get_transcoder_contribs(1, seq, dirs[0].clone(), k=5)
    g4["tc_contribs_idx"] = np.array([c[3] for c in tc_contribs], np.int32)
    g4["tc_contribs_val"] = np.array([c[4] for c in tc_contribs], np.float32)
    g4["tc_contribs_vec_proj"] = torch.stack([c[0] for c in tc_contribs]).matmul(proj).numpy()
    np.savez_compressed(os.path.join(HERE, "g4_lens.npz"), **g4)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


_COMPONENT_IDS = ["z_feature", "mlp_feature", "attn_head", "unembed", "unembed_at_token", "embed",
                  "pos_embed", "b_O", "z_sae_error", "z_sae_bias", "transcoder_error", "transcoder_bias"]


def _c(name):
    return _COMPONENT_IDS.index(name)


if __name__ == "__main__":
    main()
