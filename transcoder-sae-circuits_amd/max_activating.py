"""All-layer activation helpers with the reference's names (``src/max_activating.py:22-65``) plus the
fused replacement of its dense dump.

``get_activation_dictionary`` returns the same dict of DENSE activations as the reference (24 HIP
encode launches on one cache).  ``get_topk_dictionary`` / ``main`` are the MI355X-first form of
``main`` (:97-122): instead of appending ~2.5 MB of dense activations per token to a Python list and
pickling it, every batch is folded into per-feature top-k lists (never materialising activations).
"""
from __future__ import annotations

import torch

from load_model_and_encoders import get_model_encoders

device = "cuda" if torch.cuda.is_available() else "cpu"


def get_z_activations(model, batch, layer):
    with torch.no_grad():
        _, cache = model.run_with_cache(batch)
        z = cache["z", layer]
    return z.reshape(z.shape[0] * z.shape[1], -1)  # "b s n d -> (b s) (n d)"


def get_zsae_acts(z_acts, sae):
    with torch.no_grad():
        return sae.encode(z_acts)


def get_transcoder_acts(mlp_acts, transcoder):
    with torch.no_grad():
        return transcoder.encode(mlp_acts)


def get_activation_dictionary(cache, z_saes, transcoders, tokens):
    activation_dict = {"tokens": tokens}
    for layer in range(12):
        z_acts = cache["z", layer]
        z_acts = z_acts.reshape(z_acts.shape[0], z_acts.shape[1], -1)  # "b s n d -> b s (n d)"
        activation_dict[f"z_{layer}"] = get_zsae_acts(z_acts, z_saes[layer])
        activation_dict[f"mlp_{layer}"] = get_transcoder_acts(cache["normalized", layer, "ln2"], transcoders[layer])
    return activation_dict


def get_topk_dictionary(model, tokens, z_saes, transcoders, k: int = 20, batch_size: int = 64, **kw):
    """``{"tokens", "z_{L}": (vals, idx), "mlp_{L}": (vals, idx)}`` over all of ``tokens``."""
    from topk_scan import sweep_model

    out = sweep_model(model, tokens, z_saes=z_saes, transcoders=transcoders, k=k, batch_size=batch_size, **kw)
    out["tokens"] = tokens
    return out


def main(total_tokens: int = 1_000_000, batch_size: int = 64, seq_len: int = 128, k: int = 20,
         out_path: str = "maxact_topk.pt"):
    import open_web_text

    torch.set_grad_enabled(False)
    tl_model, z_saes, transcoders = get_model_encoders(device=device)
    tokens = open_web_text.get_tokens()[: total_tokens // seq_len, :seq_len]
    result = get_topk_dictionary(tl_model, tokens, z_saes, transcoders, k=k, batch_size=batch_size)
    torch.save({n: (tuple(t.cpu() for t in v) if isinstance(v, tuple) else v.cpu()) for n, v in result.items()}, out_path)
    print(f"Per-feature top-{k} lists saved to {out_path}")
    return result
