"""Seeded random-init dictionary weights with the reference's initial statistics.

The trained checkpoints (``pchlenski/gpt2-transcoders``, ``ckkissane/attn-saes-gpt2-small-
all-layers``; reference ``src/mlp_transcoder.py:590-600``, ``src/z_sae.py:91-96``) cannot be
fetched offline.  Goldens, parity tests and the benchmark therefore use weights drawn the
way the reference's constructors draw them -- kaiming-uniform ``W_enc``/``W_dec`` with
``W_dec`` rows normalised to unit length (reference ``src/mlp_transcoder.py:70-89``,
``src/z_sae.py:34-47``) -- plus a constant negative ``b_enc`` that sets the fraction of
active features per token (``b_enc = -0.70`` gives ~0.26 % active
for unit-variance inputs, i.e. L0 ~ 64 at ``d_sae = 24576``).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch


def _kaiming_uniform(rows: int, cols: int, gen: torch.Generator) -> torch.Tensor:
    # torch.nn.init.kaiming_uniform_ on a [rows, cols] tensor: fan_in = cols, a = 0.
    bound = math.sqrt(6.0 / cols)
    return (torch.rand(rows, cols, generator=gen, dtype=torch.float32) * 2.0 - 1.0) * bound


def make_dictionary_weights(
    d_in: int = 768,
    d_sae: int = 24576,
    d_out: Optional[int] = None,
    seed: int = 0,
    b_enc=-0.70,
    bias_scale: float = 0.05,
    transcoder: bool = True,
) -> Dict[str, torch.Tensor]:
    """CPU fp32 tensors ``W_enc [d_in,d_sae]``, ``b_enc``, ``W_dec [d_sae,d_out]``, ``b_dec``
    and (transcoders only) ``b_dec_out``.  ``b_dec``/``b_dec_out`` are small random vectors so
    that the ``x - b_dec`` prologue and the output bias are exercised (the reference
    initialises them to zero, trained checkpoints do not keep them there).  ``b_enc`` may be
    a ``(first, last)`` pair: the bias then ramps linearly over the features, giving a spread
    of per-feature firing rates.  Pre-activation std for unit-variance inputs is
    ``sqrt(2 * d_in / d_sae)`` (0.25 at 24576, 1.0 at 1536)."""
    d_out = d_out or d_in
    gen = torch.Generator().manual_seed(1_000_003 * seed + d_sae)
    W_enc = _kaiming_uniform(d_in, d_sae, gen)
    W_dec = _kaiming_uniform(d_sae, d_out, gen)
    W_dec = W_dec / W_dec.norm(dim=1, keepdim=True)
    out = {
        "W_enc": W_enc.contiguous(),
        "b_enc": (torch.linspace(float(b_enc[0]), float(b_enc[1]), d_sae, dtype=torch.float32)
                  if isinstance(b_enc, (tuple, list))
                  else torch.full((d_sae,), float(b_enc), dtype=torch.float32)),
        "W_dec": W_dec.contiguous(),
        "b_dec": torch.randn(d_in, generator=gen, dtype=torch.float32) * bias_scale,
    }
    if transcoder:
        out["b_dec_out"] = torch.randn(d_out, generator=gen, dtype=torch.float32) * bias_scale
    return out


def weights_fingerprint(w: Dict[str, torch.Tensor]) -> float:
    """Order-sensitive checksum used by the golden fixtures to detect RNG drift."""
    acc = 0.0
    for i, key in enumerate(sorted(w)):
        t = w[key].double().flatten()
        idx = torch.arange(t.numel(), dtype=torch.float64)
        acc += float((t * torch.cos(idx * (0.37 + i))).sum())
    return acc
