"""A stand-in "model" that replays pre-computed activations: row ``s`` of the token matrix carries
its own sequence index in column 1, and ``run_with_cache`` returns the stored activations of those
sequences under the requested hook name.  Lets the golden top-k fixtures (generated on raw synthetic
activations) drive the model-facing drop-in functions."""
from types import SimpleNamespace

import torch


class ReplayModel:
    def __init__(self, acts: torch.Tensor, n_seq: int, n_pos: int, n_heads: int = 12, d_vocab: int = 64):
        self.acts = acts.reshape(n_seq, n_pos, -1)
        self.n_heads = n_heads
        self.cfg = SimpleNamespace(n_layers=1, n_heads=n_heads, d_head=self.acts.shape[-1] // n_heads, d_vocab=d_vocab)
        self.tokenizer = SimpleNamespace(eos_token="<|endoftext|>", decode=lambda ids: " ".join(str(int(i)) for i in ids))
        g = torch.Generator().manual_seed(5)
        d = self.acts.shape[-1]
        self.W_U = torch.randn(d, d_vocab, generator=g).to(acts.device)
        self.W_O = torch.randn(1, n_heads, d // n_heads, d, generator=g).to(acts.device) / d**0.5

    def token_matrix(self) -> torch.Tensor:
        n_seq, n_pos, _ = self.acts.shape
        t = torch.zeros(n_seq, n_pos, dtype=torch.int64)
        t[:, 1] = torch.arange(n_seq)
        return t

    def run_with_cache(self, tokens, stop_at_layer=None, names_filter=None, return_type="logits"):
        rows = tokens[:, 1].to(self.acts.device)
        a = self.acts[rows]
        names = [names_filter] if isinstance(names_filter, str) else list(names_filter)
        cache = {}
        for n in names:
            cache[n] = a.reshape(a.shape[0], a.shape[1], self.n_heads, -1) if n.endswith("hook_z") else a
        return None, cache

    def to_str_tokens(self, ids):
        return [str(int(i)) for i in torch.as_tensor(ids).flatten().tolist()]

    def to_string(self, tok):
        return str(int(tok))
