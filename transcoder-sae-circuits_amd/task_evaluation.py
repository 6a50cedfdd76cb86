"""Task-level entry to the batched CircuitLens pass (BASELINE config 4: "CircuitLens feature-contribution
decomposition for batched IOI prompts, all layers, logit-diff direction").

Drop-in for the hot-path part of ``TaskEvaluation`` (reference ``src/task_evaluation.py``): the constructor
(:111-139: prompts -> token matrix, optional mean cache), ``str_prompts`` (:153-155) and ``logit_difference``
(:141-151), which defines the direction the decomposition is scored against.  The circuit-discovery driver, the
mean-ablation faithfulness loop and the plots of that class are the graph-search layer (out
of scope) and raise.  New here: ``logit_difference_directions`` and ``contribution_decomposition`` -- all prompts
in one ``BatchedCircuitLens`` pass instead of one ``CircuitLens`` per prompt.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch

from circuit_lens import BatchedCircuitLens
from load_model_and_encoders import get_model_encoders


def all_allowed(_component: str) -> bool:  # circuit_discovery.all_allowed, the reference's default filter
    return True


class TaskEvaluation:
    def __init__(self, prompts: List[dict], circuit_discovery_strategy: Optional[Callable] = None,
                 allowed_components_filter: Callable[[str], bool] = all_allowed, eval_index: int = -1,
                 no_mean_cache=False):
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.model = get_model_encoders(self.device)[0]
        self.circuit_discovery_strategy = circuit_discovery_strategy
        self.prompts = prompts
        self.tokens = self.model.to_tokens(self.str_prompts)
        self.eval_index = eval_index
        self.allowed_components_filter = allowed_components_filter
        self._batched: Optional[BatchedCircuitLens] = None
        if no_mean_cache:
            return
        with torch.no_grad():
            logits, cache = self.model.run_with_cache(self.tokens, return_type="logits")
        self.mean_cache = {key: value.mean(dim=0) for key, value in cache.items()}  # :133-138
        self.mean_logits = logits.mean(dim=0)

    @property
    def str_prompts(self):
        return [p["text"] for p in self.prompts]

    def logit_difference(self, prompt_idx: int):
        """Unembedding direction ``correct - counter`` (:141-151), exactly as the reference computes it:
        ``to_tokens`` prepends BOS (and may split a name into several tokens), so the result has one row per
        token of the tokenised answers, all but the last being ``W_U[:, bos] - W_U[:, bos] = 0``."""
        correct_token = self.prompts[prompt_idx]["correct"]
        counter_token = self.prompts[prompt_idx]["counter"]
        if correct_token is None or counter_token is None:
            raise ValueError("Prompt does not have correct or counter token")
        toks = self.model.to_tokens([correct_token, counter_token]).squeeze()
        unembed = self.model.tokens_to_residual_directions(toks)
        return unembed[0] - unembed[1]

    # ------------------------------------------------------------------ batched form (new)
    def logit_difference_directions(self) -> torch.Tensor:
        """``[n_prompts, d_model]``: for every prompt the last row of ``logit_difference`` -- the answer tokens'
        ``W_U[:, correct] - W_U[:, counter]`` -- gathered in one indexing operation."""
        correct = self.model.to_tokens([p["correct"] for p in self.prompts])[:, -1]
        counter = self.model.to_tokens([p["counter"] for p in self.prompts])[:, -1]
        w_u = self.model.W_U
        return (w_u[:, correct] - w_u[:, counter]).t().contiguous()

    def batched_lens(self, positions="eval") -> BatchedCircuitLens:
        """All prompts through ONE dictionary pass; ``positions="eval"`` covers ``eval_index`` of every prompt,
        ``"all"`` every position (what per-prompt circuit discovery walks back through)."""
        pos = self.eval_index if positions == "eval" else positions
        if self._batched is None or self._batched.positions != pos:
            self._batched = BatchedCircuitLens(self.str_prompts, positions=pos)
        return self._batched

    def contribution_decomposition(self, k: int = 10, positions="eval"):
        """Per prompt, at ``eval_index``: every component's normalised contribution to the prompt's logit-difference
        direction and the top-``k`` of them (``BatchedCircuitLens.direction_lens``)."""
        return self.batched_lens(positions).direction_lens(self.logit_difference_directions(), k=k)

    # ------------------------------------------------------------------ out of scope
    def _out_of_scope(self, *a, **k):
        raise NotImplementedError("circuit discovery and mean-ablation evaluation are the reference's graph-search "
                                  "layer (SURVEY.md section 2.1); use BatchedCircuitLens.lens(i) to feed it")

    get_circuit_discovery_for_prompt = _out_of_scope
    get_faithfulness_curve_over_data = _out_of_scope
    get_attn_head_freqs_over_dataset = _out_of_scope
    get_features_at_heads_over_dataset = _out_of_scope
