"""Oracle: max-activating-token selection (torch CPU / numpy).  Test infrastructure only.

Restates reference ``src/max_act_analysis.py:197-255`` (``MaxActAnalysis.active_examples``),
``src/autointerpretability.py:240-257`` (``get_feature_scores`` storage layout and fp16
rounding) and ``:260-266`` (``get_top_k_activating_examples`` index selection), and defines
the all-feature running top-k the HIP path computes instead of one scan per feature.

Ranking rule.  The reference ranks with ``argsort(descending=True)`` (max_act_analysis.py:248)
and numpy's default ``argsort`` (autointerpretability.py:262); neither is stable, so the order
among *exactly equal* activations is unspecified there.  The build fixes it: value descending,
then token id ascending.  Only strictly positive activations count (max_act_analysis.py:246).
Lists shorter than k are padded with ``(0.0, -1)``.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .encode_ref import encode_ref


def empty_state(d_sae: int, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    return torch.zeros(d_sae, k, dtype=torch.float32), torch.full((d_sae, k), -1, dtype=torch.int64)


def topk_update_ref(
    state: Tuple[torch.Tensor, torch.Tensor],
    acts: torch.Tensor,
    token_id_base: int,
    exact_ties: bool = True,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Fold one batch of dense activations ``acts [B, d_sae]`` (tokens ``base .. base+B-1``)
    into the per-feature lists ``(vals [d_sae,k] desc, idx [d_sae,k])``.

    ``exact_ties=True`` uses a stable descending sort over ``[state ; batch]`` so that equal
    values keep the smaller token id first (state entries precede the batch and batches
    arrive in ascending id order).  ``exact_ties=False`` uses ``torch.topk`` (what a torch
    user would write; used by the CPU-baseline timing, tie order unspecified).
    """
    vals, idx = state
    d_sae, k = vals.shape
    b = acts.shape[0]
    ids = torch.arange(token_id_base, token_id_base + b, dtype=torch.int64)
    cat_v = torch.cat([vals.t(), acts.to(torch.float32)], dim=0)  # [k+B, d_sae]
    cat_i = torch.cat([idx.t(), ids.unsqueeze(1).expand(b, d_sae)], dim=0)
    if exact_ties:
        order = torch.sort(cat_v, dim=0, descending=True, stable=True).indices[:k]
    else:
        order = torch.topk(cat_v, k=min(k, cat_v.shape[0]), dim=0).indices
    new_v = torch.gather(cat_v, 0, order)
    new_i = torch.gather(cat_i, 0, order)
    dead = new_v <= 0
    new_v = torch.where(dead, torch.zeros_like(new_v), new_v)
    new_i = torch.where(dead, torch.full_like(new_i, -1), new_i)
    return new_v.t().contiguous(), new_i.t().contiguous()


def topk_stream_ref(
    x: torch.Tensor,
    w: Dict[str, torch.Tensor],
    k: int = 20,
    batch_tokens: int = 16384,
    token_id_base: int = 0,
    exact_ties: bool = True,
    state: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Encode ``x [T, d_in]`` batch by batch (max_act_analysis.py:215-231) and keep the
    per-feature top-k for ALL features."""
    d_sae = w["W_enc"].shape[1]
    if state is None:
        state = empty_state(d_sae, k)
    for s in range(0, x.shape[0], batch_tokens):
        acts = encode_ref(x[s:s + batch_tokens], w["W_enc"], w["b_enc"], w["b_dec"])
        state = topk_update_ref(state, acts, token_id_base + s, exact_ties=exact_ties)
    return state


def topk_merge_ref(lists: Sequence[Tuple[torch.Tensor, torch.Tensor]], k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merge several ``(vals, idx)`` list sets into one (value desc, id asc); used for the
    per-rank lists after the all-gather."""
    v = torch.cat([l[0] for l in lists], dim=1)
    i = torch.cat([l[1] for l in lists], dim=1)
    # canonical order: sort by id ascending first (pads with id -1 carry value 0 and are
    # masked below), then stable by value descending
    big = torch.iinfo(torch.int64).max
    key_i = torch.where(i < 0, torch.full_like(i, big), i)
    o1 = torch.sort(key_i, dim=1, stable=True).indices
    v = torch.gather(v, 1, o1)
    i = torch.gather(i, 1, o1)
    o2 = torch.sort(v, dim=1, descending=True, stable=True).indices[:, :k]
    v = torch.gather(v, 1, o2)
    i = torch.gather(i, 1, o2)
    dead = (v <= 0) | (i < 0)
    v = torch.where(dead, torch.zeros_like(v), v)
    i = torch.where(dead, torch.full_like(i, -1), i)
    return v.contiguous(), i.contiguous()


def active_examples_ref(scores: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """``scores [n_seq, n_pos]`` of ONE feature -> ``(seq_pos [n_active,2], vals [n_active],
    scores)`` as max_act_analysis.py:243-253: flatten, count the positives, take that many
    from the descending order, unravel to (sequence, position)."""
    flat = scores.flatten()
    n_active = int((flat > 0).sum())
    order = torch.sort(flat, descending=True, stable=True).indices[:n_active]
    n_pos = scores.shape[1]
    seq_pos = torch.stack([order // n_pos, order % n_pos], dim=1)
    return seq_pos, flat[order], scores


def feature_scores_ref(
    x: torch.Tensor, w: Dict[str, torch.Tensor], features: Sequence[int], n_pos: int,
    batch_seqs: int = 64, use_raw_scores: bool = False,
) -> np.ndarray:
    """``get_feature_scores`` on pre-computed activations ``x [n_seq*n_pos, d_in]``:
    float16 ``[n_seq, n_feat, n_pos]`` (autointerpretability.py:216-257)."""
    from .encode_ref import raw_feature_scores_ref

    out: List[np.ndarray] = []
    step = batch_seqs * n_pos
    feats = list(features)
    for s in range(0, x.shape[0], step):
        xb = x[s:s + step]
        if use_raw_scores:
            cur = raw_feature_scores_ref(xb, w, feats)
        else:
            cur = encode_ref(xb, w["W_enc"], w["b_enc"], w["b_dec"])[:, feats]
        cur = cur.reshape(-1, n_pos, len(feats)).permute(0, 2, 1)
        out.append(cur.numpy().astype(np.float16))
    return np.concatenate(out, axis=0)


def top_k_activating_examples_ref(feature_scores: np.ndarray, k: int = 5):
    """Index part of ``get_top_k_activating_examples`` (autointerpretability.py:261-266) for
    one feature's ``[n_seq, n_pos]`` slab: returns ``(scores, seq_idx, pos_idx)``.  The
    reference's ``argsort()[-k:][::-1]`` (unstable) is restated with a stable sort, so equal
    scores come out larger-flat-index first."""
    flat = feature_scores.flatten()
    top = np.argsort(flat, kind="stable")[-k:][::-1]
    seq_idx, pos_idx = np.unravel_index(top, feature_scores.shape)
    return flat[top], seq_idx, pos_idx
