"""Oracle: CircuitLens residual decomposition and contribution scores (torch CPU).

Test infrastructure only.  Restates reference ``src/circuit_lens.py:556-675``
(``get_active_features``), ``:753-778`` / ``:940-968`` / ``:974-1004`` / ``:1031-1054`` /
``:1085-1107`` (lens dot products), ``:780-801`` (head/seq decomposition), ``:1138-1170``
(``get_transcoder_ixg``), ``:1189-1233`` (``get_attn_head_contribs``) and ``:551-554``
(``normalize_activations``).  Weights are plain dicts of CPU tensors.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch

from .encode_ref import forward_ref

AUX_ATTN = 3  # b_O, z error, z bias        (circuit_lens.py:35-40)
AUX_MLP = 2   # transcoder error, b_dec_out


def sparse_desc(acts: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Active features of one position, value-descending: ``nonzero().numel()`` then
    ``topk(count)`` (circuit_lens.py:612-614, :647-649)."""
    count = int((acts != 0).sum())
    if count == 0:
        return acts.new_zeros(0), torch.zeros(0, dtype=torch.int64)
    vals, idx = torch.topk(acts, k=count)
    return vals, idx


def project_z(vecs: torch.Tensor, W_O_layer: torch.Tensor) -> torch.Tensor:
    """``[n, n_heads*d_head]`` in z space -> ``[n, d_model]`` through ``W_O[layer]
    [n_heads, d_head, d_model]`` (circuit_lens.py:595-604, :618-627)."""
    n_heads, d_head, d_model = W_O_layer.shape
    return vecs.reshape(-1, n_heads * d_head) @ W_O_layer.reshape(n_heads * d_head, d_model)


def active_features_ref(model, cache, z_w: Sequence[Dict], t_w: Sequence[Dict], seq_index: int):
    """Returns ``(vectors [comp,d_model], values [comp], features [comp] int64, keys)`` with
    the reference's row order: embed, pos_embed, then per layer b_O, z-SAE error, z-SAE
    bias, attention features (desc), transcoder error, transcoder output bias, MLP features
    (desc) (circuit_lens.py:570-659)."""
    vec_rows: List[torch.Tensor] = [cache["embed"][0, seq_index][None], cache["pos_embed"][0, seq_index][None]]
    val_rows: List[torch.Tensor] = [torch.ones(2)]
    feat_rows: List[torch.Tensor] = [torch.full((2,), -1, dtype=torch.int64)]
    keys = []
    for layer in range(model.cfg.n_layers):
        zw, tw = z_w[layer], t_w[layer]
        z_in = cache["z", layer][0, seq_index].reshape(-1)
        z_recon, z_acts = forward_ref(z_in, zw)
        aux = torch.stack([z_in - z_recon, zw["b_dec"]])
        zv, zi = sparse_desc(z_acts)
        vec_rows += [model.b_O[layer][None], project_z(aux, model.W_O[layer]),
                     project_z(zw["W_dec"][zi] * zv[:, None], model.W_O[layer])]
        val_rows += [torch.ones(AUX_ATTN), zv]
        feat_rows += [torch.full((AUX_ATTN,), -1, dtype=torch.int64), zi]

        m_in = cache["normalized", layer, "ln2"][0, seq_index]
        m_recon, m_acts = forward_ref(m_in, tw)
        mv, mi = sparse_desc(m_acts)
        vec_rows += [(cache["mlp_out", layer][0, seq_index] - m_recon)[None], tw["b_dec_out"][None],
                     tw["W_dec"][mi] * mv[:, None]]
        val_rows += [torch.ones(AUX_MLP), mv]
        feat_rows += [torch.full((AUX_MLP,), -1, dtype=torch.int64), mi]
        keys.append({"mlp": int(mv.numel()), "attn": int(zv.numel())})
    return torch.cat(vec_rows), torch.cat(val_rows), torch.cat(feat_rows), keys


def rows_before(keys, kind: str, layer: int) -> int:
    """Number of leading rows that feed component ``kind`` of ``layer``
    (circuit_lens.py:63-74)."""
    n = 2
    for l in range(layer):
        n += keys[l]["mlp"] + keys[l]["attn"] + AUX_ATTN + AUX_MLP
    if kind == "mlp":
        n += keys[layer]["attn"] + AUX_ATTN
    return n


def normalize_ref(a: torch.Tensor) -> torch.Tensor:
    return a / a.sum()  # circuit_lens.py:551-554


def unembed_lens_raw_ref(vectors, W_U, token: int):
    return vectors @ W_U[:, token]  # circuit_lens.py:762-766


def mlp_feature_lens_raw_ref(vectors, keys, t_w, layer: int, feature: int):
    v = vectors[: rows_before(keys, "mlp", layer)]
    return v @ t_w[layer]["W_enc"][:, feature]  # circuit_lens.py:953-957


def v_lens_raw_ref(vectors, keys, model, z_w, layer: int, head: int, feature: int):
    v = vectors[: rows_before(keys, "attn", layer)]
    d_head = model.cfg.d_head
    eff_feature = z_w[layer]["W_enc"][head * d_head:(head + 1) * d_head, feature]
    return (v @ model.W_V[layer, head]) @ eff_feature  # circuit_lens.py:986-1002


def q_lens_raw_ref(vectors, keys, model, cache, layer: int, head: int, source: int):
    v = vectors[: rows_before(keys, "attn", layer)]
    return (v @ model.W_Q[layer, head]) @ cache["k", layer][0, source, head]  # :1043-1052


def k_lens_raw_ref(vectors, keys, model, cache, layer: int, head: int, destination: int):
    v = vectors[: rows_before(keys, "attn", layer)]
    return (v @ model.W_K[layer, head]) @ cache["q", layer][0, destination, head]  # :1097-1105


def head_seq_acts_ref(model, cache, z_w, layer: int, seq_index: int, feature: int):
    """Per (head, source position) contribution to one z feature's pre-activation
    (circuit_lens.py:780-812): ``pattern[h, seq, src] * v[src, h, :] . W_enc[h-slice, f]``,
    flattened head-major."""
    n_heads, d_head = model.cfg.n_heads, model.cfg.d_head
    w = z_w[layer]["W_enc"][:, feature].reshape(n_heads, d_head)
    v = cache["v", layer][0]                       # [src, head, d_head]
    pat = cache["pattern", layer][0, :, seq_index]  # [head, src]
    per = torch.einsum("shd,hd->hs", v, w) * pat
    return per.reshape(-1)


def transcoder_ixg_ref(tw: Dict, x_seq: torch.Tensor, seq_index: int, direction: torch.Tensor):
    """``(W_dec @ direction) * feature_acts`` over ALL features (circuit_lens.py:1151-1164);
    ``x_seq [1, n_pos, d_in]`` is the transcoder's input for the whole prompt."""
    _, acts = forward_ref(x_seq, tw)
    acts = acts[0, seq_index]
    return (tw["W_dec"] @ direction) * acts, acts


def attn_head_contribs_ref(model, cache, zw: Dict, layer: int, seq_index: int):
    """Active z features as residual vectors ``[n_active, d_model]`` (circuit_lens.py:1192-1233)."""
    z_in = cache["z", layer][0, seq_index].reshape(-1)
    _, acts = forward_ref(z_in, zw)
    zv, zi = sparse_desc(acts)
    return project_z(zw["W_dec"][zi] * zv[:, None], model.W_O[layer])


def contribution_scores_ref(vals: torch.Tensor, feats: torch.Tensor, W_dec: torch.Tensor, directions: torch.Tensor):
    """North-star formula ``feature_act x W_dec row . direction`` for sparse ``(feats, vals)``
    of one position against ``directions [n_dir, d_out]`` -> ``[n_active, n_dir]``."""
    return (W_dec[feats] * vals[:, None]) @ directions.t()
