"""Max-activating-example analysis of ONE feature, with the reference's surface.

Drop-in for the hot-path part of ``MaxActAnalysis`` (reference ``src/max_act_analysis.py``):
constructor (:19-53), ``active_examples`` (:197-255), ``get_active_example`` (:190-195),
``num_examples`` (:302-304), ``get_top_k_tokens`` (:156-170).  The reference encodes all ``d_sae``
features of every batch to keep one column; here only that column is computed
(``tsc_project_features``).  Scanning ALL features at once is ``top_examples_all_features`` /
``topk_scan.sweep_model`` (the fused kernel).  Circuit discovery, LLM auto-interpretation and
notebook display methods are out of scope and raise.
"""
from __future__ import annotations

from typing import List, Optional

import torch

import open_web_text
from load_model_and_encoders import get_model_encoders

BASIC_FILTER = None    # discovery_strategies.BASIC_FILTER / BASIC_STRATEGY live in the graph-search layer,
BASIC_STRATEGY = None  # which is not rebuilt; the arguments are accepted and stored for signature parity


def top_k_token_ids(model, encoder, feature_type: str, layer: int, features, k: int = 5):
    """``get_top_k_tokens`` for MANY features at once: the features' output directions
    ``W_dec[F]`` (attention dictionaries: through ``W_O[layer]``, reference :157-163) as ONE ``[F, 768] x W_U``
    product instead of one GEMV over the 154-MB unembedding per feature.  Returns ``(logits [F, k], ids [F, k])``."""
    feats = torch.as_tensor(features, device=encoder.W_dec.device, dtype=torch.long).reshape(-1)
    with torch.no_grad():
        vecs = encoder.W_dec.detach().index_select(0, feats)                       # [F, d]
        if feature_type == "attn":
            w_o = model.W_O[layer]
            vecs = vecs @ w_o.reshape(w_o.shape[0] * w_o.shape[1], w_o.shape[2])   # "f (n d), (n d) o -> f o"
        top = (vecs @ model.W_U).topk(k, dim=1)
    return top.values, top.indices


def get_top_k_tokens_for_features(model, encoder, feature_type: str, layer: int, features, k: int = 5) -> List[List[str]]:
    """The token strings of :func:`top_k_token_ids`, one list per feature."""
    return [model.to_str_tokens(row) for row in top_k_token_ids(model, encoder, feature_type, layer, features, k)[1]]


class MaxActAnalysis:
    num_sequences: int

    def __init__(self, feature_type: str, layer: int, feature: int, num_sequences=None, batch_size=128,
                 component_filter=BASIC_FILTER, strategy=BASIC_STRATEGY, tokens: Optional[torch.Tensor] = None):
        assert feature_type in ("attn", "mlp")
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.feature_type = feature_type
        model, z_saes, transcoders = get_model_encoders(self.device)
        self.model = model
        self.feature = feature
        self.layer = layer
        self.batch_size = batch_size
        self.component_filter = component_filter
        self.strategy = strategy
        self._tokens = tokens if tokens is not None else open_web_text.get_tokens()
        self.num_sequences = self._tokens.size(0) if num_sequences is None else num_sequences
        self.encoder = z_saes[layer] if feature_type == "attn" else transcoders[layer]
        self._z_saes, self._transcoders = z_saes, transcoders
        self._context_referenced_prompt_cache = dict()

    _active_examples = None

    def update_filter_and_strategy(self, component_filter=None, strategy=None):
        if component_filter is not None:
            self.component_filter = component_filter
        if strategy is not None:
            self.strategy = strategy

    # ------------------------------------------------------------------ hot path
    def _hook_name(self) -> str:
        if self.feature_type == "attn":
            return f"blocks.{self.layer}.attn.hook_z"
        return self.encoder.cfg.hook_point

    @property
    def active_examples(self):
        """``(seq_pos [n_active,2] int64, vals [n_active] desc, scores [n_seq, n_pos])`` on the CPU."""
        if self._active_examples is not None:
            return self._active_examples
        tokens = self._tokens[: self.num_sequences]
        name_filter = self._hook_name()
        chunks = []
        for i in range(0, tokens.shape[0], self.batch_size):
            with torch.no_grad():
                cur = tokens[i: i + self.batch_size]
                _, cache = self.model.run_with_cache(cur, stop_at_layer=self.layer + 1, names_filter=name_filter)
                acts = cache[name_filter]
                flat = acts.reshape(acts.shape[0] * acts.shape[1], -1)
                col = self.encoder.encode_features(flat, [self.feature], relu=True)[:, 0]
            chunks.append(col.reshape(cur.size(0), -1))
        scores = torch.cat(chunks, dim=0)  # stays on the GPU until the end: one transfer, not one per batch
        flat = scores.flatten()
        num_active = int((flat > 0).sum().item())
        order = torch.sort(flat, descending=True, stable=True).indices[:num_active]
        vals = flat[order].cpu()
        order = order.cpu()
        n_pos = scores.shape[1]
        seq_pos = torch.stack([torch.div(order, n_pos, rounding_mode="floor"), order % n_pos]).T
        self._active_examples = (seq_pos, vals, scores.cpu())
        return self._active_examples

    def get_active_example(self, i):
        seq_pos, vals, _ = self.active_examples
        seq, pos = seq_pos[i]
        return seq, pos, vals[i]

    @property
    def num_examples(self):
        return self.active_examples[0].size(0)

    def get_top_k_tokens(self, k=5) -> List[str]:
        """Tokens the feature's output direction promotes most (through ``W_U``; reference :156-170)."""
        ids = top_k_token_ids(self.model, self.encoder, self.feature_type, self.layer, [self.feature], k)[1][0]
        return self.model.to_str_tokens(ids)

    def top_examples_all_features(self, k: int = 20):
        """New capability: the k max-activating tokens of EVERY feature of this dictionary in one fused
        pass -> ``(vals [d_sae,k], seq [d_sae,k], pos [d_sae,k])``."""
        from topk_scan import sweep_model, unravel_token_ids

        tokens = self._tokens[: self.num_sequences]
        kw = {"z_saes": self._z_saes} if self.feature_type == "attn" else {"transcoders": self._transcoders}
        res = sweep_model(self.model, tokens, layers=[self.layer], k=k, batch_size=self.batch_size, distributed=False, **kw)
        vals, idx = res[("z_" if self.feature_type == "attn" else "mlp_") + str(self.layer)]
        seq, pos = unravel_token_ids(idx, tokens.shape[1])
        return vals, seq, pos

    # ------------------------------------------------------------------ out of scope
    def _out_of_scope(self, *a, **k):
        raise NotImplementedError("circuit discovery / LLM auto-interpretation / notebook display are outside the "
                                  "accelerated hot path (SURVEY.md section 2.1); use the reference for them")

    get_circuit_discovery_for_max_activating_example = _out_of_scope
    get_context_referenced_prompts_for_range = _out_of_scope
    get_feature_auto_interp = _out_of_scope
    get_context_referenced_prompt = _out_of_scope
    show_top_active_examples = _out_of_scope
    show_example = _out_of_scope
