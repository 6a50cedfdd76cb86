#!/usr/bin/env python3
"""Short profiling target: N launches of the fused encode+top-k on one dictionary.
Used under rocprofv3 (--kernel-trace --stats, or one --pmc pass at a time); not a test."""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "transcoder-sae-circuits_amd"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--d-sae", type=int, default=24576)
ap.add_argument("--tokens", type=int, default=16384)
ap.add_argument("--launches", type=int, default=12)
ap.add_argument("--mode", default="topk", choices=["topk", "dense", "sparse"])
ap.add_argument("--slots", type=int, default=0)
ap.add_argument("--bf16", action="store_true")
ap.add_argument("--screened", action="store_true", help="tsc_encode_topk_screened after --warm fp32 launches")
ap.add_argument("--warm", type=int, default=2)
ap.add_argument("--no-floor", action="store_true", help="without the cross-slot k-th-best floor")
ap.add_argument("--b-enc", type=float, default=None, help="override b_enc (e.g. -100: nothing is ever active)")
ap.add_argument("--zero-operands", action="store_true",
                help="x = 0 and W_enc = 0: the same instruction stream on operands that toggle no bits (power / clock check)")
args = ap.parse_args()

import tsc_hip  # noqa: E402
from bench import make_device_dictionary  # noqa: E402

dev = torch.device("cuda", 0)
W, be, bd = make_device_dictionary(args.d_sae, 1, dev)
if args.b_enc is not None:
    be.fill_(args.b_enc)
if args.zero_operands:
    W.zero_()
    bd.zero_()
d = tsc_hip.DeviceDict(W, be, bd, compute_dtype=tsc_hip.BF16 if args.bf16 else tsc_hip.F32_SCREEN if args.screened else tsc_hip.F32)
st = tsc_hip.TopkState(args.d_sae, 20, dev, n_slots=args.slots or None, cross_slot_floor=not args.no_floor)
g = torch.Generator(device=dev).manual_seed(3)
xs = [torch.randn(args.tokens, 768, generator=g, device=dev) for _ in range(args.launches)]
if args.zero_operands:
    for x in xs:
        x.zero_()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
_lib = tsc_hip.lib()
for i, x in enumerate(xs):
    if i == args.launches // 2:
        if hasattr(_lib, "tsc_debug_read_stamps"):  # count the timed half only
            import ctypes
            torch.cuda.synchronize()
            _lib.tsc_debug_read_stamps((ctypes.c_ulonglong * 16)(), 1)
        e0.record()
    if args.mode == "topk" and args.screened and i >= args.warm:
        d.encode_topk_screened(x, i * args.tokens, st)
    elif args.mode == "topk":
        d.encode_topk(x, i * args.tokens, st)
    elif args.mode == "dense":
        out = d.encode_dense(x)
    else:
        out = d.encode_sparse(x, 512)
e1.record()
torch.cuda.synchronize()
n = args.launches - args.launches // 2
ms = e0.elapsed_time(e1) / n
print(f"{args.mode} d_sae={args.d_sae} tokens={args.tokens}: {ms:.3f} ms/launch, "
      f"{2.0 * 768 * args.d_sae * args.tokens / ms / 1e9:.1f} TFLOP/s")
lib = tsc_hip.lib()
if hasattr(lib, "tsc_debug_read_stamps"):  # diagnosis builds (-DTSC_STAMPS) only
    import ctypes
    buf = (ctypes.c_ulonglong * 16)()
    lib.tsc_debug_read_stamps(buf, 0)
    names = ["k-loop", "pre-check", "push scan", "barrier 1", "owner", "barrier 2", "-", "-"]
    for w, o in (("wave 0", 0), ("wave 3", 8)):
        tot = sum(buf[o + i] for i in range(8))
        print(w, " ".join(f"{n}={100.0 * buf[o + i] / max(tot, 1):.2f}%" for i, n in enumerate(names) if n != "-"))
