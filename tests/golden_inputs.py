"""Rebuilds, from seeds, the inputs the golden fixtures were generated on
(``tests/golden/make_golden.py``) and checks the weight fingerprints stored with them."""
import os

import numpy as np
import torch

from standins.synthetic_dicts import make_dictionary_weights, weights_fingerprint
from standins.synthetic_model import SyntheticHookedTransformer

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PROMPT = "Mary and Jeff went to the store, and Mary gave an apple to Jeff"
COMPONENT_IDS = ["z_feature", "mlp_feature", "attn_head", "unembed", "unembed_at_token", "embed",
                 "pos_embed", "b_O", "z_sae_error", "z_sae_bias", "transcoder_error", "transcoder_bias"]


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def _check(fp_expected, w):
    got = weights_fingerprint(w)
    assert abs(got - float(fp_expected)) <= 1e-9 * max(1.0, abs(got)), "seeded weights drifted from the fixture's"


def g1_inputs():
    g = load("g1_encode.npz")
    wt = make_dictionary_weights(768, 1536, seed=11, b_enc=-0.30, transcoder=True)
    wz = make_dictionary_weights(768, 1536, seed=12, b_enc=-0.30, transcoder=False)
    x = torch.randn(64, 768, generator=torch.Generator().manual_seed(101))
    _check(g["fp_tc"], wt)
    _check(g["fp_zs"], wz)
    assert abs(float(x.double().sum()) - float(g["x_sum"])) < 1e-9
    return g, wt, wz, x


def g1_full_weights():
    g = load("g1_encode.npz")
    w = make_dictionary_weights(768, 24576, seed=13, b_enc=-0.70, transcoder=True)
    _check(g["fp_full"], w)
    return w


def g2_inputs():
    g = load("g2_topk.npz")
    w = make_dictionary_weights(768, 1536, seed=21, b_enc=(-1.5, -3.6), transcoder=True)
    n_seq, n_pos = 32, 128
    xs = torch.randn(n_seq * n_pos, 768, generator=torch.Generator().manual_seed(202))
    xs[n_pos * 5: n_pos * 5 + 8] = xs[0:8]
    _check(g["fp"], w)
    assert abs(float(xs.double().sum()) - float(g["x_sum"])) < 1e-9
    return g, w, xs, n_seq, n_pos


def lens_inputs(n_layers=3):
    g3 = load("g3_active_features.npz")
    model = SyntheticHookedTransformer(n_layers=n_layers, d_vocab=1000, n_ctx=32, seed=7)
    z_w, t_w, fps = [], [], []
    for l in range(n_layers):
        wz = make_dictionary_weights(768, 1536, seed=300 + l, b_enc=-1.3, transcoder=False)
        wt = make_dictionary_weights(768, 1536, seed=400 + l, b_enc=-2.0, transcoder=True)
        fps += [weights_fingerprint(wz), weights_fingerprint(wt)]
        z_w.append(wz)
        t_w.append(wt)
    assert np.allclose(np.array(fps), g3["fp"], rtol=1e-9, atol=1e-9)
    return g3, load("g4_lens.npz"), model, z_w, t_w


def proj8():
    return torch.randn(768, 8, generator=torch.Generator().manual_seed(303))


def g5_inputs():
    """Inputs of ``g5_callers.npz`` (tests/golden/make_golden_callers.py): the g2 transcoder and activations, a
    z-SAE on the same activations, and the IOI prompts the reference generated."""
    import json

    g5 = load("g5_callers.npz")
    g, w, xs, n_seq, n_pos = g2_inputs()
    wz = make_dictionary_weights(768, 1536, seed=22, b_enc=(-1.5, -3.6), transcoder=False)
    _check(g5["fp_tc"], w)
    _check(g5["fp_zs"], wz)
    prompts = json.load(open(os.path.join(GOLDEN, "g5_prompts.json")))
    return g5, w, wz, xs, n_seq, n_pos, prompts


def same_ranking(got_pairs, got_vals, ref_pairs, ref_vals):
    """Rankings agree when the value sequences are equal and, within every run of exactly equal values, the same
    SET of (row) pairs appears (the reference's argsort is unstable, so the order inside a tie is unspecified)."""
    import numpy as np

    got_pairs, ref_pairs = np.asarray(got_pairs), np.asarray(ref_pairs)
    got_vals, ref_vals = np.asarray(got_vals), np.asarray(ref_vals)
    if got_pairs.shape != ref_pairs.shape or not np.array_equal(got_vals, ref_vals):
        return False
    start = 0
    n = len(ref_vals)
    while start < n:
        end = start + 1
        while end < n and ref_vals[end] == ref_vals[start]:
            end += 1
        a = sorted(map(tuple, got_pairs[start:end].reshape(end - start, -1).tolist()))
        b = sorted(map(tuple, ref_pairs[start:end].reshape(end - start, -1).tolist()))
        if a != b:
            return False
        start = end
    return True
