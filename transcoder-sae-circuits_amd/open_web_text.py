"""Token source of the max-activation scans.

The reference streams ``Skylion007/openwebtext`` and tokenises 25 600 rows of 128 tokens AT IMPORT
(``src/open_web_text.py:8-19``).  There is no network here, so importing this module does nothing;
``open_web_text_tokens`` is filled by ``set_tokens`` (tests and the benchmark use seeded synthetic
ids) or by ``load_tokens`` when a dataset is reachable.
"""
from __future__ import annotations

from typing import Optional

import torch

SEQ_LEN = 128       # open_web_text.py:13
NUM_ROWS = 25_600   # open_web_text.py:16
open_web_text_tokens: Optional[torch.Tensor] = None


def set_tokens(tokens: torch.Tensor) -> torch.Tensor:
    global open_web_text_tokens
    if tokens.dim() != 2:
        raise ValueError("tokens must be [n_seq, seq_len]")
    open_web_text_tokens = tokens.to(torch.int64)
    return open_web_text_tokens


def synthetic_tokens(n_seq: int, d_vocab: int, seq_len: int = SEQ_LEN, seed: int = 0, bos: int = 0) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    t = torch.randint(1, d_vocab, (n_seq, seq_len), generator=g, dtype=torch.int64)
    t[:, 0] = bos  # every row starts with BOS, as tokenize_and_concatenate produces
    return set_tokens(t)


def load_tokens(num_rows: int = NUM_ROWS):  # pragma: no cover - needs network
    from datasets import load_dataset
    from transformers import AutoTokenizer

    ds = load_dataset("Skylion007/openwebtext", split="train", streaming=True)
    tok = AutoTokenizer.from_pretrained("gpt2")
    rows, buf = [], []
    for ex in ds:
        buf.extend(tok(ex["text"] + tok.eos_token)["input_ids"])
        while len(buf) >= SEQ_LEN - 1 and len(rows) < num_rows:
            rows.append([tok.bos_token_id] + buf[: SEQ_LEN - 1])
            buf = buf[SEQ_LEN - 1:]
        if len(rows) >= num_rows:
            break
    return set_tokens(torch.tensor(rows))


def get_tokens() -> torch.Tensor:
    if open_web_text_tokens is None:
        raise RuntimeError("open_web_text_tokens is empty: call open_web_text.set_tokens / synthetic_tokens / load_tokens")
    return open_web_text_tokens
