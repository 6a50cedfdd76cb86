"""Seeded random-init GPT-2-small-shaped stand-in for TransformerLens' ``HookedTransformer``.

The reference obtains its model with ``HookedTransformer.from_pretrained("gpt2-small")``
(reference ``src/load_model_and_encoders.py:21``).  Neither ``transformer_lens`` nor the
GPT-2 checkpoint exists offline, so every parity test, golden fixture and benchmark in
this repository drives the hot path with this stand-in instead.  It reproduces the
*interface* the hot path's callers touch -- ``to_tokens``, ``run_with_cache`` (with
``names_filter`` / ``stop_at_layer``), the ``cache[...]`` key algebra, ``W_O/b_O/W_U/
W_Q/W_K/W_V/OV``, ``blocks[l].mlp``, ``tokenizer.decode``, ``to_str_tokens`` -- and the
*arithmetic identities* the reference's own tests rely on
(reference ``src/test_circuit_lens.py:141-219``): ``attn_out = z . W_O + b_O``,
``resid_mid = resid_pre + attn_out``, ``resid_post = resid_mid + mlp_out``, LayerNormPre
``normalized = (x - mean) / scale``.

It is upstream of the accelerated path (the model forward stays
PyTorch); nothing in here is a HIP kernel.
"""
from __future__ import annotations

import math
import zlib
from types import SimpleNamespace
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Union

import torch
from torch import nn


def get_act_name(name: str, layer: Optional[int] = None, layer_type: Optional[str] = None) -> str:
    """Short-hand -> full hook name, the subset of TransformerLens' rule the callers use.

    ``("z", 3)`` -> ``blocks.3.attn.hook_z``; ``("normalized", 3, "ln2")`` ->
    ``blocks.3.ln2.hook_normalized``; ``"embed"`` -> ``hook_embed``.
    """
    if layer is None:
        if name in ("embed", "pos_embed"):
            return f"hook_{name}"
        return name
    if name in ("q", "k", "v", "z", "pattern", "attn_scores", "result"):
        layer_type = "attn"
    elif name in ("pre", "post", "mid"):
        layer_type = "mlp"
    if layer_type in ("attn", "mlp", "ln1", "ln2"):
        return f"blocks.{layer}.{layer_type}.hook_{name}"
    return f"blocks.{layer}.hook_{name}"


class SyntheticCache(dict):
    """``dict`` of activations that also accepts TransformerLens tuple keys."""

    def __getitem__(self, key):
        if isinstance(key, tuple):
            key = get_act_name(*key)
        elif not dict.__contains__(self, key):
            key = get_act_name(key)
        return dict.__getitem__(self, key)


class _ByteWordTokenizer:
    """Whitespace tokenizer with CRC-hashed ids; deterministic across processes."""

    def __init__(self, d_vocab: int):
        self.d_vocab = d_vocab
        self.bos_token_id = 0
        self.eos_token_id = 0
        self.eos_token = "<|endoftext|>"
        self._id_to_str: Dict[int, str] = {0: self.eos_token}

    def encode(self, text: str) -> List[int]:
        ids = []
        for i, word in enumerate(text.split(" ")):
            piece = word if i == 0 else " " + word
            tid = zlib.crc32(piece.encode("utf-8")) % (self.d_vocab - 1) + 1
            self._id_to_str.setdefault(tid, piece)
            ids.append(tid)
        return ids

    def decode(self, ids: Iterable[int]) -> str:
        out = []
        for t in ids:
            t = int(t)
            out.append(self._id_to_str.get(t, f"<{t}>"))
        return "".join(out)


class _MLP(nn.Module):
    def __init__(self, W_in, b_in, W_out, b_out):
        super().__init__()
        self.W_in, self.b_in, self.W_out, self.b_out = W_in, b_in, W_out, b_out

    def forward(self, x):
        pre = x @ self.W_in + self.b_in
        post = 0.5 * pre * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (pre + 0.044715 * pre**3)))
        return post @ self.W_out + self.b_out


class SyntheticHookedTransformer(nn.Module):
    """GPT-2-small geometry (d_model 768, 12 heads x 64) with folded LayerNorm.

    ``n_layers``, ``d_vocab`` and ``n_ctx`` are configurable so fixtures stay small; the
    dictionaries fix ``d_model = 768`` (reference ``src/mlp_transcoder.py:595``).
    """

    def __init__(
        self,
        n_layers: int = 12,
        d_model: int = 768,
        n_heads: int = 12,
        d_vocab: int = 50257,
        n_ctx: int = 128,
        d_mlp: Optional[int] = None,
        seed: int = 0,
        device: Union[str, torch.device] = "cpu",
        dtype: torch.dtype = torch.float32,
    ):
        super().__init__()
        d_head = d_model // n_heads
        d_mlp = d_mlp or 4 * d_model
        self.cfg = SimpleNamespace(
            n_layers=n_layers, d_model=d_model, n_heads=n_heads, d_head=d_head,
            d_vocab=d_vocab, n_ctx=n_ctx, d_mlp=d_mlp, eps=1e-5, device=str(device),
        )
        g = torch.Generator().manual_seed(seed)

        def rnd(*shape, scale):
            return (torch.randn(*shape, generator=g, dtype=torch.float32) * scale).to(device=device, dtype=dtype)

        s = 1.0 / math.sqrt(d_model)
        self.W_E = rnd(d_vocab, d_model, scale=1.0)
        self.W_pos = rnd(n_ctx, d_model, scale=0.3)
        self.W_Q = rnd(n_layers, n_heads, d_model, d_head, scale=s)
        self.W_K = rnd(n_layers, n_heads, d_model, d_head, scale=s)
        self.W_V = rnd(n_layers, n_heads, d_model, d_head, scale=s)
        self.b_Q = rnd(n_layers, n_heads, d_head, scale=0.02)
        self.b_K = rnd(n_layers, n_heads, d_head, scale=0.02)
        self.b_V = rnd(n_layers, n_heads, d_head, scale=0.02)
        self.W_O = rnd(n_layers, n_heads, d_head, d_model, scale=s)
        self.b_O = rnd(n_layers, d_model, scale=0.02)
        self.W_in = rnd(n_layers, d_model, d_mlp, scale=s)
        self.b_in = rnd(n_layers, d_mlp, scale=0.02)
        self.W_out = rnd(n_layers, d_mlp, d_model, scale=1.0 / math.sqrt(d_mlp))
        self.b_out = rnd(n_layers, d_model, scale=0.02)
        self.W_U = rnd(d_model, d_vocab, scale=s)
        self.b_U = rnd(d_vocab, scale=0.02)
        self.tokenizer = _ByteWordTokenizer(d_vocab)
        self.blocks = [
            SimpleNamespace(mlp=_MLP(self.W_in[l], self.b_in[l], self.W_out[l], self.b_out[l]))
            for l in range(n_layers)
        ]
        self.device = torch.device(device)

    # ------------------------------------------------------------------ tokens
    def to_tokens(self, text: Union[str, Sequence[str]], prepend_bos: bool = True) -> torch.Tensor:
        texts = [text] if isinstance(text, str) else list(text)
        rows = []
        for t in texts:
            ids = self.tokenizer.encode(t)
            rows.append(([self.tokenizer.bos_token_id] if prepend_bos else []) + ids)
        width = max(len(r) for r in rows)
        rows = [r + [self.tokenizer.bos_token_id] * (width - len(r)) for r in rows]
        return torch.tensor(rows, dtype=torch.long, device=self.device)

    def to_str_tokens(self, inp) -> List[str]:
        if isinstance(inp, str):
            inp = self.to_tokens(inp)[0]
        inp = torch.as_tensor(inp).flatten().tolist()
        return [self.tokenizer.decode([t]) for t in inp]

    def to_string(self, tok) -> str:
        tok = torch.as_tensor(tok).flatten().tolist()
        return self.tokenizer.decode(tok)

    def to_single_token(self, s: str) -> int:
        ids = self.tokenizer.encode(s)
        assert len(ids) == 1
        return ids[0]

    def tokens_to_residual_directions(self, tokens) -> torch.Tensor:
        tokens = torch.as_tensor(tokens, device=self.device)
        return self.W_U[:, tokens].movedim(0, -1)

    @property
    def OV(self) -> torch.Tensor:
        # [layer, head, d_model, d_model]
        return torch.einsum("lhmd,lhdo->lhmo", self.W_V, self.W_O)

    # ----------------------------------------------------------------- forward
    @staticmethod
    def _ln_pre(x: torch.Tensor, eps: float):
        x = x - x.mean(dim=-1, keepdim=True)
        scale = (x.pow(2).mean(dim=-1, keepdim=True) + eps).sqrt()
        return x / scale, scale

    @torch.no_grad()
    def run_with_cache(
        self,
        tokens: torch.Tensor,
        return_type: Optional[str] = "logits",
        names_filter: Union[None, str, Sequence[str], Callable[[str], bool]] = None,
        stop_at_layer: Optional[int] = None,
    ):
        if names_filter is None:
            keep = lambda n: True  # noqa: E731
        elif isinstance(names_filter, str):
            keep = lambda n: n == names_filter  # noqa: E731
        elif callable(names_filter):
            keep = names_filter
        else:
            wanted = set(names_filter)
            keep = lambda n: n in wanted  # noqa: E731

        cache = SyntheticCache()

        def put(name, value):
            if keep(name):
                cache[name] = value
            return value

        out = self._forward(tokens, put, return_type, stop_at_layer)
        return out, cache

    @torch.no_grad()
    def run_with_hooks(self, tokens: torch.Tensor, return_type: Optional[str] = "logits", fwd_hooks=(),
                       stop_at_layer: Optional[int] = None, **_ignored):
        """TransformerLens' ``run_with_hooks`` for forward hooks: ``fwd_hooks = [(name or name-filter, fn)]``,
        ``fn(activation, hook=<object with .name>)``; a tensor returned by ``fn`` replaces the activation."""
        table = []
        for sel, fn in fwd_hooks:
            table.append(((lambda n, s=sel: n == s) if isinstance(sel, str) else sel, fn))

        def put(name, value):
            for match, fn in table:
                if match(name):
                    res = fn(value, hook=SimpleNamespace(name=name))
                    if res is not None:
                        value = res
            return value

        return self._forward(tokens, put, return_type, stop_at_layer)

    def _forward(self, tokens, put, return_type, stop_at_layer):
        cfg = self.cfg
        tokens = torch.as_tensor(tokens).to(self.device)
        if tokens.dim() == 1:
            tokens = tokens.unsqueeze(0)
        b, n = tokens.shape
        embed = put("hook_embed", self.W_E[tokens])
        pos_embed = put("hook_pos_embed", self.W_pos[:n].unsqueeze(0).expand(b, -1, -1))
        resid = embed + pos_embed
        n_run = cfg.n_layers if stop_at_layer is None else min(stop_at_layer, cfg.n_layers)
        causal = torch.ones(n, n, dtype=torch.bool, device=self.device).tril()
        for l in range(n_run):
            resid = put(f"blocks.{l}.hook_resid_pre", resid)
            normed, scale = self._ln_pre(resid, cfg.eps)
            put(f"blocks.{l}.ln1.hook_scale", scale)
            normed = put(f"blocks.{l}.ln1.hook_normalized", normed)
            q = put(f"blocks.{l}.attn.hook_q", torch.einsum("bpm,hmd->bphd", normed, self.W_Q[l]) + self.b_Q[l])
            k = put(f"blocks.{l}.attn.hook_k", torch.einsum("bpm,hmd->bphd", normed, self.W_K[l]) + self.b_K[l])
            v = put(f"blocks.{l}.attn.hook_v", torch.einsum("bpm,hmd->bphd", normed, self.W_V[l]) + self.b_V[l])
            scores = torch.einsum("bqhd,bkhd->bhqk", q, k) / math.sqrt(cfg.d_head)
            scores = put(f"blocks.{l}.attn.hook_attn_scores", scores.masked_fill(~causal, float("-inf")))
            pattern = put(f"blocks.{l}.attn.hook_pattern", scores.softmax(dim=-1))
            z = put(f"blocks.{l}.attn.hook_z", torch.einsum("bhqk,bkhd->bqhd", pattern, v))
            attn_out = put(f"blocks.{l}.hook_attn_out", torch.einsum("bqhd,hdm->bqm", z, self.W_O[l]) + self.b_O[l])
            resid = put(f"blocks.{l}.hook_resid_mid", resid + attn_out)
            normed2, scale2 = self._ln_pre(resid, cfg.eps)
            put(f"blocks.{l}.ln2.hook_scale", scale2)
            normed2 = put(f"blocks.{l}.ln2.hook_normalized", normed2)
            mlp_out = put(f"blocks.{l}.hook_mlp_out", self.blocks[l].mlp(normed2))
            resid = put(f"blocks.{l}.hook_resid_post", resid + mlp_out)
        if stop_at_layer is not None:
            return resid
        final, fscale = self._ln_pre(resid, cfg.eps)
        put("ln_final.hook_scale", fscale)
        final = put("ln_final.hook_normalized", final)
        if return_type is None:
            return None
        return final @ self.W_U + self.b_U

    def forward(self, tokens, return_type="logits"):
        return self.run_with_cache(tokens, return_type=return_type, names_filter=lambda n: False)[0]
