#!/usr/bin/env python3
"""Experiment: the 24 fused encode + top-k launches of a bench step on one stream vs alternating over S streams
(tails of one launch overlapping the head of the next).  Prints ms per step for each S."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "transcoder-sae-circuits_amd"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
import tsc_hip  # noqa: E402

dev = torch.device("cuda", 0)
widths = bench.dictionary_widths()
dicts, states = [], []
for i, (_, _, d_sae) in enumerate(widths):
    W, be, bd = bench.make_device_dictionary(d_sae, 1000 + i, dev)
    dicts.append(tsc_hip.DeviceDict(W, be, bd))
    states.append(tsc_hip.TopkState(d_sae, 20, dev))
gen = torch.Generator(device=dev).manual_seed(1)
B, steps, warm = 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 8, 5
xs = [torch.randn(len(widths), B, 768, generator=gen, device=dev) for _ in range(4)]
for n_streams in (1, 2, 3, 1, 2):
    for st in states:
        st.reset()
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    main = torch.cuda.current_stream()

    def step(s):
        if n_streams == 1:
            for i, d in enumerate(dicts):
                d.encode_topk(xs[s % 4][i], s * B, states[i])
            return
        ev = torch.cuda.Event()
        ev.record(main)
        for st_ in streams:
            st_.wait_event(ev)
        for i, d in enumerate(dicts):
            with torch.cuda.stream(streams[i % n_streams]):
                d.encode_topk(xs[s % 4][i], s * B, states[i])
        for st_ in streams:
            main.wait_stream(st_)

    for s in range(warm):
        step(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        step(warm + s)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    flop = sum(2.0 * 768 * w * B for _, _, w in widths)
    print(f"streams={n_streams}: {ms:.2f} ms/step, {flop / ms / 1e9:.1f} TFLOP/s", flush=True)
