"""Oracle: the caller-level arithmetic around the dictionaries (torch CPU / numpy).  Test infrastructure only.

Restates reference ``src/max_act_analysis.py:156-170`` (``get_top_k_tokens``), ``src/autointerpretability.py:
226-238`` (``get_feature_scores`` with ``feature_post`` / ``use_decoder``), ``src/task_evaluation.py:141-151``
(``logit_difference``), ``src/circuit_lens.py:1235-1274`` (``get_transcoder_contribs``) and ``:1276-1317``
(``get_top_contribs``, MLP branch: its attention branch raises whenever a z feature is active).  Pinned by
``tests/golden/g5_callers.npz``, which holds the outputs of those reference functions themselves.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch

from .lens_ref import transcoder_ixg_ref


def top_k_tokens_ref(w: Dict[str, torch.Tensor], feature: int, W_U: torch.Tensor, k: int = 5, W_O_layer=None):
    """Feature's output direction through the unembedding; attention features go through ``W_O[layer]
    [n_heads, d_head, d_model]`` first (max_act_analysis.py:157-168).  Returns ``(logits [k], ids [k])``."""
    vec = w["W_dec"][feature]
    if W_O_layer is not None:
        n_heads, d_head, d_model = W_O_layer.shape
        vec = torch.einsum("nd,ndo->o", vec.reshape(n_heads, d_head), W_O_layer)
    top = (vec @ W_U).topk(k)
    return top.values, top.indices


def feature_scores_post_ref(x: torch.Tensor, w: Dict[str, torch.Tensor], features: Sequence[int], feature_post: torch.Tensor,
                            n_pos: int, use_decoder: bool = False) -> np.ndarray:
    """Raw-score mode with an explicit ``feature_post [d_in, n]`` (autointerpretability.py:226-238): bias is
    ``-(b_dec @ post)`` for a transcoder with ``use_decoder`` else ``b_enc[F] - b_dec @ post``; fp16
    ``[n_seq, n, n_pos]``."""
    bias = -(w["b_dec"] @ feature_post) if use_decoder else w["b_enc"][list(features)] - (w["b_dec"] @ feature_post)
    cur = x.to(torch.float32) @ feature_post + bias
    return cur.reshape(-1, n_pos, feature_post.shape[1]).permute(0, 2, 1).numpy().astype(np.float16)


def logit_difference_ref(model, correct: str, counter: str) -> torch.Tensor:
    """task_evaluation.py:148-151: ``to_tokens([correct, counter]).squeeze()`` (BOS prepended), unembedding
    directions of every token, first minus second."""
    toks = model.to_tokens([correct, counter]).squeeze()
    unembed = model.tokens_to_residual_directions(toks)
    return unembed[0] - unembed[1]


def ln_constant_ref(cache, vector: torch.Tensor, layer: int, token: int) -> torch.Tensor:
    x = cache["resid_pre", layer][0, token]  # circuit_lens.py:1172-1187 with is_ln2=False
    y = cache["normalized", layer, "ln1"][0, token]
    if torch.dot(vector, x) == 0:
        return torch.tensor(0.0)
    return torch.dot(vector, y) / torch.dot(vector, x)


def transcoder_contribs_ref(cache, tw: Dict, layer: int, seq_index: int, direction: torch.Tensor, k: int = 5):
    """``[(vector, layer, seq, feature, score)]`` of the k largest pull-backs (circuit_lens.py:1259-1274)."""
    pulled, _ = transcoder_ixg_ref(tw, cache["normalized", layer, "ln2"], seq_index, direction)
    top_v, top_i = torch.topk(pulled, k=k)
    out = []
    for contrib, index in zip(top_v, top_i):
        vector = tw["W_enc"][:, index] * (tw["W_dec"] @ direction)[index]
        vector = vector * ln_constant_ref(cache, vector, layer, seq_index)
        out.append((vector, layer, seq_index, int(index), float(contrib)))
    return out


def top_contribs_ref(cache, t_w: Sequence[Dict], layer: int, seq_index: int, direction: torch.Tensor, k: int = 5) -> List:
    """MLP branch of ``get_top_contribs`` (circuit_lens.py:1279-1283, 1312-1317)."""
    allc = []
    for cur in range(layer + 1):
        allc += transcoder_contribs_ref(cache, t_w[cur], cur, seq_index, direction, k=k)
    scores = torch.tensor([c[4] for c in allc])
    _, top = torch.topk(scores, k=min(k, len(scores)))
    return [allc[int(i)] for i in top]
