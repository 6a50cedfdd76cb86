#!/usr/bin/env python3
"""Benchmark of the hot path: fused dictionary encode + running per-feature top-k.

Workload (BASELINE.json configs[1], "All 12 transcoders + 12 z-SAEs, fp32, 16M synthetic
activation tokens on one MI355X"): every token position carries 24 activation vectors [768]
(one per dictionary: ``ln2.hook_normalized`` for the 12 MLP transcoders, flattened ``hook_z`` for
the 12 attention z-SAEs); each is encoded by its dictionary (d_sae 24576, 49152 for z-SAE layers 5
and 7 as in reference src/z_sae.py:16,18) and folded into that dictionary's per-feature top-20.
One "step" = one batch of ``--tokens-per-step`` positions (default 65536) through all 24 dictionaries; the 16M
positions of the config are ``16M / tokens-per-step`` such steps, so tokens/s is size-independent.
Weights are seeded random-init with the reference's init statistics, activations are seeded
N(0,1) (no checkpoints or datasets offline): ``"data": "synthetic"``.

Contract: ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on rank 0.
N > 1 is launched by torch.distributed.run; ranks own disjoint token-id ranges (weak scaling), the
only collective is the final all-gather of the finished lists followed by a local merge.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(REPO, "transcoder-sae-circuits_amd"), REPO):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

D_IN = 768
PEAK_F32_MATRIX_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 FLOP/clk/CU x 256 CU x 2.4 GHz
PMC_FABRIC_BYTES_PER_LAUNCH = 1.870e9 + 3.9e6  # FETCH_SIZE x2 + WRITE_SIZE, see roofline.traffic below
ZSAE_WIDE_LAYERS = (5, 7)


def dictionary_widths():
    widths = [("mlp", l, 24576) for l in range(12)]
    widths += [("attn", l, 49152 if l in ZSAE_WIDE_LAYERS else 24576) for l in range(12)]
    return widths


def make_device_dictionary(d_sae, seed, device):
    """Reference init statistics (kaiming-uniform W_enc; mlp_transcoder.py:70-74) generated on the
    device; b_enc = -0.70 gives ~0.26 % active features per token at d_sae = 24576."""
    import math

    g = torch.Generator(device=device).manual_seed(seed)
    bound = math.sqrt(6.0 / d_sae)
    W_enc = (torch.rand(D_IN, d_sae, generator=g, device=device) * 2 - 1) * bound
    # keep the pre-activation scale of the 24576-wide dictionaries for the wide ones too
    W_enc *= math.sqrt(d_sae / 24576.0)
    b_enc = torch.full((d_sae,), -0.70, device=device)
    b_dec = torch.randn(D_IN, generator=g, device=device) * 0.05
    return W_enc, b_enc, b_dec


def cpu_baseline(tokens, threads):
    """The oracle (CPU port of the reference path: torch sgemm encode + torch.topk running top-k)
    on ONE 24576-wide dictionary, converted to positions/s of the 24-dictionary workload."""
    from oracle.topk_ref import topk_stream_ref
    from synthetic_dicts import make_dictionary_weights

    torch.set_num_threads(threads)
    w = make_dictionary_weights(D_IN, 24576, seed=8, b_enc=-0.70)
    x = torch.randn(tokens + 4096, D_IN, generator=torch.Generator().manual_seed(0))
    topk_stream_ref(x[:4096], w, k=20, batch_tokens=4096, exact_ties=False)  # warm-up
    t0 = time.perf_counter()
    topk_stream_ref(x[4096:], w, k=20, batch_tokens=16384, exact_ties=False)
    dt = time.perf_counter() - t0
    units = sum(wd for _, _, wd in dictionary_widths()) / 24576.0
    return {
        "value": tokens / dt / units,
        "unit": "positions/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{tokens} tokens x one 24576-wide fp32 dictionary, batch 16384, k=20 in {dt:.1f} s "
                  f"({tokens / dt:.0f} dictionary-tokens/s), divided by the {units:.0f} such units per position",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    # 512 sequences x 128 positions per step.  The reference batches 128 x 128 (max_act_analysis.py:25) because it
    # materialises the [tokens, d_sae] activations; the fused path does not, so the batch is a free parameter and
    # a larger one amortises the per-launch head and tail (16384: 146.3, 32768: 147.4, 65536: 148.4 TFLOP/s).
    ap.add_argument("--tokens-per-step", type=int, default=65536)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--cpu-tokens", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # rehearsal of the N > 1 path on a one-GPU box: every rank on cuda:0, collectives over gloo
    ap.add_argument("--rehearse-on-one-gpu", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    device = torch.device("cuda", 0 if args.rehearse_on_one_gpu else local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            import datetime

            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=60))
        else:
            dist.init_process_group("nccl", device_id=device)

    import tsc_hip

    B, K, W = args.tokens_per_step, args.steps, args.warmup
    widths = dictionary_widths()
    dicts, states = [], []
    for i, (_, _, d_sae) in enumerate(widths):
        W_enc, b_enc, b_dec = make_device_dictionary(d_sae, 1000 + i, device)
        dicts.append(tsc_hip.DeviceDict(W_enc, b_enc, b_dec))
        states.append(tsc_hip.TopkState(d_sae, args.k, device))
    # inputs resident in HBM before the timed region: one distinct batch per step and dictionary
    gen = torch.Generator(device=device).manual_seed(12345 + rank)
    # one distinct batch per step while they fit in two thirds of the free HBM (196 GB = 160 steps of the default
    # size); a longer run cycles through the pool (repeated tokens tie with their first copy, which only makes
    # the top-k epilogue busier)
    per_batch = len(widths) * B * D_IN * 4
    n_batches = max(1, min(K + W, int(torch.cuda.mem_get_info(device)[0] * 0.66) // per_batch))
    xs = [torch.randn(len(widths), B, D_IN, generator=gen, device=device) for _ in range(n_batches)]
    rank_base = rank * (K + W) * B  # contiguous, disjoint token-id range per rank

    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in widths]
          for _ in range(K)]

    def step(s, timed_index=None):
        base = rank_base + s * B
        for i, d in enumerate(dicts):
            if timed_index is not None:
                ev[timed_index][i][0].record()
            d.encode_topk(xs[s % n_batches][i], base, states[i])
            if timed_index is not None:
                ev[timed_index][i][1].record()

    def finish():
        outs = [st.finalize() for st in states]
        if world > 1:
            from topk_scan import allgather_merge_many

            if args.rehearse_on_one_gpu:  # gloo moves host tensors; the merge still runs on the device
                outs = allgather_merge_many([(v.cpu(), i.cpu()) for v, i in outs],
                                            lambda v, i: tsc_hip.topk_merge(v.to(device), i.to(device)))
            else:
                outs = allgather_merge_many(outs, tsc_hip.topk_merge)
        return outs

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()

    for s in range(W):
        step(s)
    finish()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(K):
        step(W + s, timed_index=s)
    outs = finish()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist

        t = torch.tensor([elapsed], device="cpu" if args.rehearse_on_one_gpu else device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-launch kernel time from the HIP events recorded on the launch stream
    kern_ms = sum(a.elapsed_time(b) for row in ev for a, b in row)
    n_launch = K * len(widths)
    flops = sum(2.0 * D_IN * d_sae * B for _, _, d_sae in widths) * K
    achieved = flops / (kern_ms * 1e-3) / 1e12
    filled = sum(int((idx[:, -1] >= 0).sum()) for _, idx in outs)
    total_feats = sum(d for _, _, d in widths)

    if rank == 0:
        out = {
            "metric": "activation tokens/sec encoded + per-feature top-k (1/8 GPU); MFMA util %",
            "value": world * K * B / elapsed,
            "unit": "positions/s (each position = 24 activation vectors through 12 transcoders + 12 z-SAEs, top-20 per feature)",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: 12 transcoders + 12 z-SAEs (10 x 24576, 2 x 49152), fp32, fused encode + running top-20",
                "positions_per_step_per_gpu": B,
                "dictionaries": len(widths),
                "features_total": total_feats,
                "k": args.k,
                "parallelism": f"token-sharded x{world}, final all-gather + merge" if world > 1 else "single GPU",
                "lists_full_at_end": f"{filled}/{total_feats}",
            },
            "dictionary_tokens_per_s": world * K * B * len(widths) / elapsed,
            "roofline": {
                "bound": "mfma",
                "kernel": "encode_f32_kernel<TOPK>",
                "achieved": achieved,
                "peak": PEAK_F32_MATRIX_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_F32_MATRIX_TFLOPS,
                "avg_launch_ms": kern_ms / n_launch,
                "launches": n_launch,
                # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/r01_encode_topk_pmc_v9.json), per launch on a
                # 24576-wide dictionary, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950.  These are
                # the L2's fabric-side requests: x + W_enc^T (125 MB) stay in the 256 MiB Infinity Cache, so almost
                # all of it is served on die; the algorithmic bytes are 126 MB per launch.
                "traffic": PMC_FABRIC_BYTES_PER_LAUNCH,
                "traffic_note": "bytes per launch at d_sae=24576, L2 fabric side incl. Infinity Cache hits (offline PMC pass); "
                                "algorithmic 1.26e8",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(args.cpu_tokens, threads)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
