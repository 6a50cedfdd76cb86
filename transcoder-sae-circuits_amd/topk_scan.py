"""Streaming per-feature top-k of max-activating tokens over many dictionaries, single- or multi-GPU.

This is the hot path itself (BASELINE.json north star): activations stream through
``encode_topk`` (fused MFMA encode + top-k insert, no activation matrix in HBM), each rank owns a
contiguous range of token rows, and the per-rank lists are combined with ONE all-gather + a local
k-way merge at the end.  It replaces, for every feature at once, what the
reference does one feature at a time in ``MaxActAnalysis.active_examples``
(``src/max_act_analysis.py:197-255``) and what ``max_activating.main`` (``src/max_activating.py:97-122``)
approximates by dumping dense activations.

The numerical engine is pluggable only so that the multi-process plumbing can be exercised on CPU
with ``gloo`` in the tests (which inject the oracle); the default and only product engine is HIP.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, Optional, Sequence, Tuple

import torch

import tsc_hip


class HipTopkEngine:
    """Running top-k of one dictionary on the GPU."""

    # The first rows of a sweep run on the fp32 kernel: on empty lists every positive activation is a candidate of the
    # screen.  16384 rows put ~40 activations into a list of a feature that fires on 0.26 % of the tokens, so its k-th
    # best already dismisses half of what follows; rarer features stay "cold" for as long as they have fewer than k
    # activations, which is exactly when their candidates are few.  (Whatever the lists hold, the result is the
    # fp32 kernel's bit for bit: a queue that overflows makes the library redo the chunk exactly.)
    SCREEN_AFTER_TOKENS = 16384

    def __init__(self, dictionary, k: int = 20, bf16: bool = False, screened: bool = True):
        self.dictionary = dictionary
        self.k = k
        self.bf16 = bf16
        self.screened = screened  # bit-identical lists either way (tsc_encode_topk_screened); False forces the fp32 kernel
        self.state = dictionary.new_topk_state(k)
        self.tokens_seen = 0

    # While the lists are young a batch is fed in pieces no longer than what the lists have already seen: the screen's
    # floor is the k-th best BEFORE a call, so a piece as long as the history has ~k candidates per feature above it, one
    # three times as long ~3k -- each an exact recompute.  (65536-row batches: [16384 fp32][16384][32768], then whole.)
    YOUNG_UNTIL_TOKENS = 131072

    def update(self, x: torch.Tensor, token_id_base: int) -> None:
        can_screen = self.screened and not self.bf16
        n, done = x.shape[0], 0
        if not can_screen:
            self.dictionary.encode_topk(x, token_id_base, self.state, bf16=self.bf16, screened=False)
            done = n
        while done < n:
            seen = self.tokens_seen + done
            if seen < self.SCREEN_AFTER_TOKENS:  # the head of a sweep on the fp32 kernel (see above)
                step, screened = min(n - done, self.SCREEN_AFTER_TOKENS - seen), False
            elif seen < self.YOUNG_UNTIL_TOKENS:
                step, screened = min(n - done, max(self.SCREEN_AFTER_TOKENS, seen)), True
            else:
                step, screened = n - done, True
            piece = x if step == n else x[done:done + step]
            self.dictionary.encode_topk(piece, token_id_base + done, self.state, bf16=False, screened=screened)
            done += step
        self.tokens_seen += n

    def finalize(self) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.state.finalize()

    def load(self, vals: torch.Tensor, idx: torch.Tensor) -> None:
        """Resume from finalized lists (they become the running state)."""
        dev = self.state.device
        self.state.load(vals.to(dev, torch.float32), idx.to(dev, torch.int64))
        self.tokens_seen = max(self.SCREEN_AFTER_TOKENS, self.YOUNG_UNTIL_TOKENS)  # resumed lists are warm

    @staticmethod
    def merge(vals: torch.Tensor, idx: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return tsc_hip.topk_merge(vals.contiguous(), idx.contiguous())


def shard_rows(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous row range ``[start, stop)`` of rank ``rank``; the first ``n_rows % world`` ranks get
    one extra row so the shards differ by at most one."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    base, extra = divmod(n_rows, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def allgather_merge(vals: torch.Tensor, idx: torch.Tensor, merge: Callable, group=None):
    """All ranks end with the identical merged lists.  The message is tiny (d_sae * k * 12 bytes per
    rank), so one flat all-gather over xGMI is latency-bound; no ring tuning is needed."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    d_sae, k = vals.shape
    gv = torch.empty(world * d_sae, k, dtype=vals.dtype, device=vals.device)  # concatenated along dim 0:
    gi = torch.empty(world * d_sae, k, dtype=idx.dtype, device=idx.device)    # the layout both RCCL and gloo accept
    dist.all_gather_into_tensor(gv, vals.contiguous(), group=group)
    dist.all_gather_into_tensor(gi, idx.contiguous(), group=group)
    return merge(gv.view(world, d_sae, k), gi.view(world, d_sae, k))


def allgather_merge_many(lists, merge: Callable, group=None):
    """:func:`allgather_merge` for several dictionaries at once: their finalized lists (same ``k``) are
    concatenated along the feature axis, so the whole exchange is two all-gathers and one merge launch
    whatever the number of dictionaries (the merge treats the concatenation as one wide dictionary)."""
    widths = [v.shape[0] for v, _ in lists]
    vals = torch.cat([v for v, _ in lists], dim=0)
    idx = torch.cat([i for _, i in lists], dim=0)
    mv, mi = allgather_merge(vals, idx, merge, group)
    return list(zip(torch.split(mv, widths, dim=0), torch.split(mi, widths, dim=0)))


class MaxActSweep:
    """Top-k lists for a set of named dictionaries fed with activation batches.

    ``feed({name: x [tokens, d_in]}, token_id_base)`` may be called any number of times with
    increasing, non-overlapping token-id ranges; ``result()`` returns ``{name: (vals [d_sae,k],
    idx [d_sae,k])}``, merged across ranks when a process group is active."""

    def __init__(self, dictionaries: Dict[str, object], k: int = 20, engine_factory: Optional[Callable] = None,
                 distributed: Optional[bool] = None, group=None):
        factory = engine_factory or (lambda d: HipTopkEngine(d, k))
        self.engines = {name: factory(d) for name, d in dictionaries.items()}
        self.k = k
        self.group = group
        if distributed is None:
            import torch.distributed as dist

            distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.distributed = distributed
        self.tokens_seen = 0

    def feed(self, acts: Dict[str, torch.Tensor], token_id_base: int) -> None:
        n = None
        for name, x in acts.items():
            x2 = x.reshape(-1, x.shape[-1])
            n = x2.shape[0] if n is None else n
            if x2.shape[0] != n:
                raise ValueError("all dictionaries must be fed the same number of token rows per call")
            self.engines[name].update(x2, token_id_base)
        self.tokens_seen += n or 0

    # ---- resume point (a compact top-k file instead of the reference's dump of
    # dense activations, src/max_activating.py:117-120).  Plain tensors only: torch.load(weights_only=True).
    def state_dict(self) -> Dict[str, torch.Tensor]:
        """This rank's finalized lists (NOT merged across ranks) and how many token rows went in."""
        out = {"tokens_seen": torch.tensor(self.tokens_seen, dtype=torch.int64), "k": torch.tensor(self.k)}
        for name, eng in self.engines.items():
            vals, idx = eng.finalize()
            out[f"{name}.vals"], out[f"{name}.idx"] = vals.cpu(), idx.cpu()
        return out

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        if int(sd["k"]) != self.k:
            raise ValueError(f"checkpoint holds top-{int(sd['k'])} lists, this sweep keeps top-{self.k}")
        missing = [n for n in self.engines if f"{n}.vals" not in sd]
        if missing:
            raise KeyError(f"checkpoint has no lists for {missing}")
        for name, eng in self.engines.items():
            eng.load(sd[f"{name}.vals"], sd[f"{name}.idx"])
        self.tokens_seen = int(sd["tokens_seen"])

    def save(self, path: str) -> None:
        torch.save(self.state_dict(), path)

    def load(self, path: str) -> None:
        self.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))

    def result(self) -> Dict[str, Tuple[torch.Tensor, torch.Tensor]]:
        names = list(self.engines)
        lists = [self.engines[n].finalize() for n in names]
        if self.distributed and names:
            lists = allgather_merge_many(lists, self.engines[names[0]].merge, self.group)
        return dict(zip(names, lists))


def hook_names(kind: str, layer: int, encoder=None) -> str:
    if kind == "attn":
        return f"blocks.{layer}.attn.hook_z"  # max_act_analysis.py:207
    return getattr(getattr(encoder, "cfg", None), "hook_point", None) or f"blocks.{layer}.ln2.hook_normalized"


class _HookedCapture:
    """Upstream producer of the sweep: forward hooks write the wanted activations of a
    batch straight into a preallocated ``[tokens, d_in]`` ring slot per dictionary -- no ``run_with_cache`` dict,
    no per-batch allocation -- and the fused encode + top-k of that batch runs on a SIDE stream while the next
    batch's model forward runs on the main one.  Two ring slots: slot ``b & 1`` is rewritten by batch ``b + 2``
    only after the side stream has finished batch ``b`` (event ``done``)."""

    def __init__(self, hooks: Dict[str, str], d_in: Dict[str, int], max_rows: int, device):
        self.hooks = hooks
        self.ring = {name: torch.empty(2, max_rows, d_in[name], dtype=torch.float32, device=device) for name in hooks}
        self.side = torch.cuda.Stream(device=device)
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]
        self.done = [None, None]
        self.batch = 0

    def forward_and_feed(self, model, rows: torch.Tensor, stop_at: int, sweep: "MaxActSweep", base: int) -> None:
        slot = self.batch & 1
        main = torch.cuda.current_stream()
        if self.done[slot] is not None:
            main.wait_event(self.done[slot])  # the encodes that read this ring slot two batches ago
        n_rows = rows.shape[0] * rows.shape[1]
        by_hook: Dict[str, list] = {}
        for name, hook in self.hooks.items():
            by_hook.setdefault(hook, []).append(name)

        def grab(act, hook):
            flat = act.reshape(n_rows, -1)  # "b pos n d -> (b pos) (n d)", max_act_analysis.py:227
            for name in by_hook[hook.name]:
                self.ring[name][slot, :n_rows].copy_(flat)

        with torch.no_grad():
            model.run_with_hooks(rows, return_type=None, stop_at_layer=stop_at,
                                 fwd_hooks=[(h, grab) for h in by_hook])
        self.ready[slot].record(main)
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.ready[slot])
            sweep.feed({name: self.ring[name][slot, :n_rows] for name in self.hooks}, base)
            ev = torch.cuda.Event()
            ev.record(self.side)
            self.done[slot] = ev
        self.batch += 1

    def finish(self) -> None:
        torch.cuda.current_stream().wait_stream(self.side)


def sweep_model(model, tokens: torch.Tensor, z_saes: Optional[Sequence] = None, transcoders: Optional[Sequence] = None,
                layers: Optional[Iterable[int]] = None, k: int = 20, batch_size: int = 128,
                rank: int = 0, world: int = 1, engine_factory: Optional[Callable] = None, distributed: Optional[bool] = None,
                capture: str = "auto"):
    """Run ``model`` over ``tokens [n_seq, n_pos]`` and keep, for every feature of every given
    dictionary, the ``k`` max-activating tokens.  Token ids are ``seq * n_pos + pos`` (global), rank
    ``rank`` of ``world`` processes rows ``shard_rows(n_seq, rank, world)``.  Returns
    ``{"z_{L}" / "mlp_{L}": (vals, idx)}`` -- the keys of ``get_activation_dictionary``
    (reference src/max_activating.py:55,63).

    ``capture``: ``"hooks"`` = forward hooks into a preallocated input ring, encodes on a side stream
    (:class:`_HookedCapture`); ``"cache"`` = the reference's ``run_with_cache`` + dict per batch
    (max_act_analysis.py:219-239); ``"auto"`` = hooks when the model offers ``run_with_hooks``, the engine is the
    HIP one and the model lives on a GPU."""
    n_seq, n_pos = tokens.shape
    dicts, hooks = {}, {}
    n_layers = len(z_saes) if z_saes is not None else len(transcoders)
    layers = list(range(n_layers)) if layers is None else list(layers)
    for l in layers:
        if z_saes is not None:
            dicts[f"z_{l}"] = z_saes[l]
            hooks[f"z_{l}"] = hook_names("attn", l)
        if transcoders is not None:
            dicts[f"mlp_{l}"] = transcoders[l]
            hooks[f"mlp_{l}"] = hook_names("mlp", l, transcoders[l])
    sweep = MaxActSweep(dicts, k=k, engine_factory=engine_factory, distributed=distributed)
    start, stop = shard_rows(n_seq, rank, world)
    wanted = sorted(set(hooks.values()))
    stop_at = max(layers) + 1
    dev = getattr(next(iter(dicts.values())), "W_enc", None)
    on_gpu = dev is not None and dev.is_cuda
    if capture == "auto":
        capture = "hooks" if (hasattr(model, "run_with_hooks") and engine_factory is None and on_gpu) else "cache"
    if capture not in ("hooks", "cache"):
        raise ValueError(f"capture must be 'auto', 'hooks' or 'cache' (got {capture!r})")
    producer = None
    if capture == "hooks":
        d_in = {name: int(d.W_enc.shape[0]) for name, d in dicts.items()}
        producer = _HookedCapture(hooks, d_in, batch_size * n_pos, next(iter(dicts.values())).W_enc.device)
    for s in range(start, stop, batch_size):
        rows = tokens[s:min(s + batch_size, stop)]
        if producer is not None:
            producer.forward_and_feed(model, rows, stop_at, sweep, s * n_pos)
            continue
        with torch.no_grad():
            _, cache = model.run_with_cache(rows, stop_at_layer=stop_at, names_filter=wanted)
        acts = {}
        for name, hook in hooks.items():
            a = cache[hook]
            acts[name] = a.reshape(a.shape[0] * a.shape[1], -1)  # "b pos n d -> (b pos) (n d)", max_act_analysis.py:227
        sweep.feed(acts, s * n_pos)
    if producer is not None:
        producer.finish()
    return sweep.result()


def unravel_token_ids(idx: torch.Tensor, n_pos: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Global token id -> ``(sequence, position)``; padding ids (-1) map to ``(-1, -1)``."""
    seq = torch.where(idx >= 0, torch.div(idx, n_pos, rounding_mode="floor"), idx)
    pos = torch.where(idx >= 0, idx % n_pos, idx)
    return seq, pos
