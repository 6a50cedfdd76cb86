"""Process-global model + dictionary cache: the seam every caller goes through.

Same names and semantics as reference ``src/load_model_and_encoders.py``: ``model_encoder_cache`` is a
``(model, z_saes, mlp_transcoders)`` tuple, ``get_model_encoders(device)`` returns it (building it on
first use).  Pre-seeding the cache swaps the backend with no caller changes;
``install_synthetic`` does that with the stand-in model and seeded random dictionaries, which is what
runs offline.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Any, Optional, Sequence, Tuple

import torch

from mlp_transcoder import SparseTranscoder
from z_sae import ZSAE

model_encoder_cache: Optional[Tuple[Any, Any, Any]] = None
model_cache: Optional[Any] = None


def get_model(device):
    global model_cache
    if model_cache is not None:
        return model_cache
    try:
        from transformer_lens import HookedTransformer  # type: ignore
    except ImportError as e:  # offline image
        raise RuntimeError("transformer_lens is not installed: seed the cache with install_synthetic(...) "
                           "or assign load_model_and_encoders.model_cache") from e
    model_cache = HookedTransformer.from_pretrained("gpt2-small", device=device)
    return model_cache


def get_model_encoders(device):
    global model_encoder_cache
    if model_encoder_cache is not None:
        return model_encoder_cache
    model = get_model(device)
    z_saes = [ZSAE.load_zsae_for_layer(i) for i in range(model.cfg.n_layers)]
    mlp_transcoders = [SparseTranscoder.load_from_hugging_face(i) for i in range(model.cfg.n_layers)]
    model_encoder_cache = (model, z_saes, mlp_transcoders)
    return model_encoder_cache


def build_transcoder(w, layer: int, device) -> SparseTranscoder:
    """A SparseTranscoder holding the given weight dict (``synthetic_dicts.make_dictionary_weights``)."""
    cfg = SimpleNamespace(
        d_in=int(w["W_enc"].shape[0]), d_sae=int(w["W_enc"].shape[1]), l1_coefficient=1e-3, dtype=torch.float32,
        device="cpu", is_transcoder=True, d_out=int(w["W_dec"].shape[1]), is_sparse_connection=False,
        use_ghost_grads=False, hook_point=f"blocks.{layer}.ln2.hook_normalized", hook_point_layer=layer)
    m = SparseTranscoder(cfg)
    with torch.no_grad():
        for k, v in w.items():
            getattr(m, k).data.copy_(v)
    m = m.to(device).eval()
    m.cfg.device = str(device)
    return m


def build_zsae(w, layer: int, device) -> ZSAE:
    m = ZSAE({"dict_size": int(w["W_enc"].shape[1]), "l1_coeff": 1.0, "enc_dtype": "fp32", "seed": 0,
              "act_size": int(w["W_enc"].shape[0]), "device": "cpu", "layer": layer})
    with torch.no_grad():
        for k, v in w.items():
            getattr(m, k).data.copy_(v)
    m = m.to(device).eval()
    m.cfg["device"] = str(device)
    return m


def install_synthetic(device, n_layers: int = 12, d_sae: int = 24576, d_vocab: int = 50257, n_ctx: int = 128,
                      seed: int = 0, z_b_enc: float = -0.3, t_b_enc: float = -0.7,
                      z_weights: Optional[Sequence] = None, t_weights: Optional[Sequence] = None, model=None):
    """Seed the cache with the stand-in GPT-2-shaped model and random-init dictionaries."""
    global model_encoder_cache, model_cache
    from standins.synthetic_dicts import make_dictionary_weights
    from standins.synthetic_model import SyntheticHookedTransformer

    if model is None:
        model = SyntheticHookedTransformer(n_layers=n_layers, d_vocab=d_vocab, n_ctx=n_ctx, seed=seed, device=device)
    z_saes, tcs = [], []
    for l in range(model.cfg.n_layers):
        wz = z_weights[l] if z_weights is not None else make_dictionary_weights(768, d_sae, seed=300 + l, b_enc=z_b_enc, transcoder=False)
        wt = t_weights[l] if t_weights is not None else make_dictionary_weights(768, d_sae, seed=400 + l, b_enc=t_b_enc, transcoder=True)
        z_saes.append(build_zsae(wz, l, device))
        tcs.append(build_transcoder(wt, l, device))
    model_cache = model
    model_encoder_cache = (model, z_saes, tcs)
    return model_encoder_cache


def clear_cache():
    global model_encoder_cache, model_cache
    model_encoder_cache = None
    model_cache = None
