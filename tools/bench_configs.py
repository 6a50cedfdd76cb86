#!/usr/bin/env python3
"""Side measurements for BASELINE.json configs 4 and 5 (bench.py stays on config 2, the metric's config).
Prints one JSON line per config; the outputs kept under profiles/ were produced by this script.

config 4: CircuitLens contribution pass -- 512 positions x 24 dictionaries (26 units of 24576 features),
          sparse emit + sparse reconstruction + contribution scores against one direction; positions/s.
config 5: wide-dictionary stress -- d_sae = 131072, bf16 operands, fused encode + top-20; tokens/s.
Synthetic weights/activations with the reference's init statistics (see bench.py)."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "transcoder-sae-circuits_amd"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402
import tsc_hip  # noqa: E402
from bench import dictionary_widths, make_device_dictionary  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def config4(positions=512, reps=10):
    g = torch.Generator(device=dev).manual_seed(3)
    dicts = []
    for i, (_, _, d_sae) in enumerate(dictionary_widths()):
        W, be, bd = make_device_dictionary(d_sae, 2000 + i, dev)
        Wd = torch.randn(d_sae, 768, generator=g, device=dev)
        Wd /= Wd.norm(dim=1, keepdim=True)
        dicts.append(tsc_hip.DeviceDict(W, be, bd, Wd, bd))
    xs = torch.randn(len(dicts), positions, 768, generator=g, device=dev)
    direction = torch.randn(1, 768, generator=g, device=dev)
    active = [0]

    def one_pass():
        counts = []
        for i, d in enumerate(dicts):
            feat, val, count = d.encode_sparse(xs[i], 512, check_capacity=False)
            d.decode_sparse(feat, val, count)
            d.contrib_scores(feat, val, count, direction)
            counts.append(count)
        counts = torch.stack(counts)
        assert int(counts.max()) <= 512  # the capacity check, once per pass
        active[0] = int(counts.sum())

    t = timed(one_pass, reps)
    n_active = active[0]
    by = n_active * (768 * 4 + 8) * 2 + positions * len(dicts) * 768 * 8 + sum(2 * 768 * w * 4 for _, _, w in dictionary_widths())
    return {"config": "configs[3]: CircuitLens contribution pass, 512 positions x 24 dictionaries, 1 direction",
            "value": positions / t, "unit": "positions/s", "ms_per_pass": t * 1e3,
            "active_features_per_position_and_dictionary": n_active / positions / len(dicts),
            "algorithmic_GB_per_s": by / t / 1e9,
            "note": "one capacity check (host sync) per pass"}


def config5(tokens=16384, launches=40):
    d_sae = 131072
    W, be, bd = make_device_dictionary(d_sae, 5, dev)
    d = tsc_hip.DeviceDict(W, be, bd, compute_dtype=tsc_hip.BF16)
    st = tsc_hip.TopkState(d_sae, 20, dev)
    g = torch.Generator(device=dev).manual_seed(4)
    xs = [torch.randn(tokens, 768, generator=g, device=dev) for _ in range(launches)]
    for i in range(launches // 2):
        d.encode_topk(xs[i], i * tokens, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(launches // 2, launches):
        d.encode_topk(xs[i], i * tokens, st)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / (launches - launches // 2)
    tf = 2.0 * 768 * d_sae * tokens / t / 1e12
    return {"config": "configs[4]: d_sae=131072 random-init transcoder, bf16 MFMA encode with fused top-20",
            "value": tokens / t, "unit": "tokens/s", "ms_per_launch": t * 1e3, "tokens_per_launch": tokens,
            "roofline": {"bound": "mfma", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0},
            "dtype": "bf16 operands, f32 accumulate", "note": "x conversion kernel included"}


if __name__ == "__main__":
    print(json.dumps(config4()), flush=True)
    print(json.dumps(config5()), flush=True)
