#!/usr/bin/env python3
"""Upstream-producer measurement: model forward + fused sweep over all 24 full-width
dictionaries, with the reference-shaped ``run_with_cache`` dict per batch against forward hooks into a preallocated
input ring + encodes on a side stream.  Stand-in GPT-2-small-shaped model, synthetic tokens.  Prints one JSON line."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "transcoder-sae-circuits_amd"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
import load_model_and_encoders as lme  # noqa: E402
from mlp_transcoder import SparseTranscoder  # noqa: E402
from standins.synthetic_model import SyntheticHookedTransformer  # noqa: E402
from topk_scan import HipTopkEngine, sweep_model  # noqa: E402
from types import SimpleNamespace  # noqa: E402
from z_sae import ZSAE  # noqa: E402

dev = torch.device("cuda", 0)
n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.set_grad_enabled(False)
model = SyntheticHookedTransformer(n_layers=12, d_vocab=8192, n_ctx=128, seed=4, device=dev)
z_saes, tcs = [], []
for _, layer, d_sae in bench.dictionary_widths()[12:]:
    m = ZSAE({"dict_size": d_sae, "l1_coeff": 1.0, "enc_dtype": "fp32", "seed": 40 + layer, "act_size": 768,
              "device": str(dev), "layer": layer})
    m.b_enc.fill_(-0.2)
    z_saes.append(m)
for _, layer, d_sae in bench.dictionary_widths()[:12]:
    cfg = SimpleNamespace(d_in=768, d_sae=d_sae, l1_coefficient=1e-3, dtype=torch.float32, device=str(dev),
                          is_transcoder=True, d_out=768, is_sparse_connection=False, use_ghost_grads=False,
                          hook_point=f"blocks.{layer}.ln2.hook_normalized", hook_point_layer=layer)
    m = SparseTranscoder(cfg)
    m.b_enc.fill_(-0.7)
    tcs.append(m)
tokens = torch.randint(1, 8192, (n_seq, 128), generator=torch.Generator().manual_seed(0)).to(dev)
out = {"tokens": n_seq * 128, "dictionaries": 24, "batch": "128 sequences x 128 positions (max_act_analysis.py:25)"}
res = {}
for mode in ("cache", "hooks", "cache", "hooks"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res[mode] = sweep_model(model, tokens, z_saes, tcs, k=20, batch_size=128, capture=mode, distributed=False)
    torch.cuda.synchronize()
    out[mode + "_tokens_per_s"] = n_seq * 128 / (time.perf_counter() - t0)  # the second round of each mode is kept
# the same with the fp32 kernel only (round-1 behaviour); the engine factory disables the hook capture's "auto", so say it
for mode in ("cache", "hooks", "cache", "hooks"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = sweep_model(model, tokens, z_saes, tcs, k=20, batch_size=128, capture=mode, distributed=False,
                    engine_factory=lambda d: HipTopkEngine(d, 20, screened=False))
    torch.cuda.synchronize()
    out[mode + "_fp32_only_tokens_per_s"] = n_seq * 128 / (time.perf_counter() - t0)
out["fp32_only_lists_identical_to_screened"] = all(torch.equal(r[n][0], res["hooks"][n][0]) and torch.equal(r[n][1], res["hooks"][n][1]) for n in r)
t0 = time.perf_counter()
for s in range(0, n_seq, 128):
    model.run_with_hooks(tokens[s:s + 128], return_type=None, stop_at_layer=12, fwd_hooks=[])
torch.cuda.synchronize()
out["model_forward_only_tokens_per_s"] = n_seq * 128 / (time.perf_counter() - t0)
out["lists_identical"] = all(torch.equal(res["cache"][n][1], res["hooks"][n][1]) and
                             torch.equal(res["cache"][n][0], res["hooks"][n][0]) for n in res["cache"])
print(json.dumps(out))
