"""Oracle: dictionary encode / forward (torch CPU fp32).  Test infrastructure only.

Restates reference ``src/mlp_transcoder.py:194-213`` (``SparseTranscoder.encode``),
``:108-147`` (the ``sae_out`` / ``feature_acts`` part of ``forward``), ``src/z_sae.py:73-77``
(``ZSAE.encode``) and ``:57-62`` (``x_reconstruct`` / ``acts`` of ``forward``).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch


def encode_ref(x: torch.Tensor, W_enc: torch.Tensor, b_enc: torch.Tensor, b_dec: torch.Tensor) -> torch.Tensor:
    """``relu((x - b_dec) @ W_enc + b_enc)`` over the last axis of ``x``.

    mlp_transcoder.py:200-212 subtracts ``b_dec`` first, contracts ``d_in`` and adds
    ``b_enc`` before the ReLU; z_sae.py:75-76 is the same expression.
    """
    x = x.to(torch.float32)
    centred = x - b_dec
    pre = torch.matmul(centred, W_enc) + b_enc
    return torch.clamp_min(pre, 0.0)


def encode_bf16_ref(x: torch.Tensor, W_enc: torch.Tensor, b_enc: torch.Tensor, b_dec: torch.Tensor) -> torch.Tensor:
    """The ``TSC_BF16`` contract (BASELINE config 5; no counterpart in the reference, whose shipped
    dictionaries are fp32): the same expression as :func:`encode_ref` on bf16-rounded operands
    (round to nearest even, ``(x - b_dec)`` rounded after the subtraction) with fp32 accumulation.
    Parity unpinned by reference fixtures -- this function *defines* the mode; it is tied to the
    reference only through :func:`encode_ref` (the two agree to bf16 precision, tested)."""
    centred = (x.to(torch.float32) - b_dec).to(torch.bfloat16).to(torch.float32)
    W = W_enc.to(torch.bfloat16).to(torch.float32)
    return torch.clamp_min(torch.matmul(centred, W) + b_enc, 0.0)


def decode_ref(acts: torch.Tensor, W_dec: torch.Tensor, out_bias: torch.Tensor) -> torch.Tensor:
    """``acts @ W_dec + bias``: mlp_transcoder.py:131-138 (bias = ``b_dec_out``),
    z_sae.py:62 (bias = ``b_dec``)."""
    return torch.matmul(acts, W_dec) + out_bias


def forward_ref(x: torch.Tensor, w: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns ``(reconstruction, feature_acts)``; a transcoder (has ``b_dec_out``) adds
    ``b_dec_out``, a z-SAE adds ``b_dec``."""
    acts = encode_ref(x, w["W_enc"], w["b_enc"], w["b_dec"])
    bias: Optional[torch.Tensor] = w.get("b_dec_out")
    if bias is None:
        bias = w["b_dec"]
    return decode_ref(acts, w["W_dec"], bias), acts


def raw_feature_scores_ref(x: torch.Tensor, w: Dict[str, torch.Tensor], features) -> torch.Tensor:
    """Pre-ReLU scores of selected features with the bias folded,
    ``x @ W_enc[:,F] + (b_enc[F] - b_dec @ W_enc[:,F])`` (autointerpretability.py:226-238)."""
    sel = w["W_enc"][:, features]
    folded = w["b_enc"][features] - torch.matmul(w["b_dec"], sel)
    return torch.matmul(x.to(torch.float32), sel) + folded
