#!/usr/bin/env python3
"""Latency of the per-node lens calls the circuit search makes (reference src/circuit_discovery.py:448,558), in the shape of
the reference's own stopwatch loop (src/circuit_lens_test_notebook.py:199-208: 4000 x get_unembed_lens_for_prompt_token
at random positions 1..8): GPT-2-small-shaped stand-in model, 24 dictionaries at their real widths, one prompt.
Reports microseconds per call for the fused path and for the reference-shaped `vectors @ direction` path, steady state
(positions seen before) and the cost of the first call at a position.  Not a test."""
import json
import os
import random
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "transcoder-sae-circuits_amd"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
import load_model_and_encoders as lme  # noqa: E402
from circuit_lens import CircuitLens  # noqa: E402
from mlp_transcoder import SparseTranscoder  # noqa: E402
from standins.synthetic_model import SyntheticHookedTransformer  # noqa: E402
from types import SimpleNamespace  # noqa: E402
from z_sae import ZSAE  # noqa: E402

torch.manual_seed(4)
model = SyntheticHookedTransformer(n_layers=12, d_vocab=8192, n_ctx=32, seed=4, device=dev)
z_saes, tcs = [], []
for _, layer, d_sae in bench.dictionary_widths()[12:]:
    z_saes.append(ZSAE({"dict_size": d_sae, "l1_coeff": 1.0, "enc_dtype": "fp32", "seed": 40 + layer, "act_size": 768,
                        "device": str(dev), "layer": layer}))
for _, layer, d_sae in bench.dictionary_widths()[:12]:
    cfg = SimpleNamespace(d_in=768, d_sae=d_sae, l1_coefficient=1e-3, dtype=torch.float32, device=str(dev), is_transcoder=True,
                          d_out=768, is_sparse_connection=False, use_ghost_grads=False,
                          hook_point=f"blocks.{layer}.ln2.hook_normalized", hook_point_layer=layer)
    tcs.append(SparseTranscoder(cfg))
lme.model_cache = model
lme.model_encoder_cache = (model, z_saes, tcs)
prompt = "Mary and Jeff went to the store, and Mary gave an apple to Jeff"
probe = CircuitLens(prompt)
for enc, (kind, layer) in [(e, ("attn", l)) for l, e in enumerate(z_saes)] + [(e, ("mlp", l)) for l, e in enumerate(tcs)]:
    x = probe._dictionary_input(kind, layer)
    enc.b_enc.fill_(-2.8 * float((x @ enc.W_enc).std()))  # ~0.26 % of the features active, as in bench.py's config 4
out = {"prompt_tokens": probe.n_tokens, "calls": n_calls}
_pass = CircuitLens(prompt)._ensure_pass()
out["list_capacity"] = int(_pass[0].shape[2])
out["max_active_per_position"] = torch.tensor(_pass[4]).amax(dim=0).tolist()
for fused in (True, False):
    lens = CircuitLens(prompt)
    lens.FUSED_LENS = fused
    lens._ensure_pass()
    torch.cuda.synchronize()
    first = []
    for pos in range(1, 9):  # first call at each position (the vectors path builds [components, 768] here)
        t0 = time.perf_counter()
        lens.get_unembed_lens_for_prompt_token(pos, visualize=False)
        first.append((time.perf_counter() - t0) * 1e6)
    rnd = random.Random(0)
    t0 = time.perf_counter()
    for _ in range(n_calls):  # the reference's loop
        lens.get_unembed_lens_for_prompt_token(rnd.randint(1, 8), visualize=False)
    unembed = (time.perf_counter() - t0) / n_calls * 1e6
    mf = [int(lens._active("mlp", 6, p)[0][0]) for p in range(1, 9)]
    t0 = time.perf_counter()
    for _ in range(n_calls):
        p = rnd.randint(1, 8)
        lens.get_mlp_feature_lens_at_seq(6, p, mf[p - 1], visualize=False)
    mlp = (time.perf_counter() - t0) / n_calls * 1e6
    t0 = time.perf_counter()
    for _ in range(n_calls):
        lens.get_q_lens_on_head_seq(7, rnd.randint(0, 11), rnd.randint(1, 4), rnd.randint(5, 8), visualize=False)
    q = (time.perf_counter() - t0) / n_calls * 1e6
    out["fused" if fused else "vectors_path"] = {
        "first_call_at_a_position_us_median": sorted(first)[len(first) // 2], "first_call_at_a_position_us_max": max(first),
        "unembed_lens_us_per_call": unembed, "mlp_feature_lens_us_per_call": mlp, "q_lens_us_per_call": q}
print(json.dumps(out))
