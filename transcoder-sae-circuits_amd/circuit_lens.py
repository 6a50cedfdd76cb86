"""CircuitLens: residual-stream decomposition into dictionary features and "lens" contribution
scores, with the reference's Python surface and batched HIP kernels underneath.

Drop-in for the hot-path part of reference ``src/circuit_lens.py``:
``ActiveFeatures`` (:45-323, row layout and index algebra are API), ``ComponentLens`` (:329-523) and
``CircuitLens`` (:526-1317).  What changed underneath:

* the reference runs 24 weight-bound M=1 dictionary forwards per *position* (:581-659); here each
  dictionary is run ONCE over all positions of the prompt (``tsc_encode_sparse`` emits the active
  (feature, value) lists value-descending, ``tsc_decode_sparse`` the reconstructions), and
  ``get_active_features(pos)`` only gathers ``W_dec`` rows (``tsc_decode_rows``) for that position;
* contribution scores of the form ``feature_act x W_dec row . direction`` go through
  ``tsc_contrib_scores`` on the sparse active set (``contribution_scores``, ``get_transcoder_ixg``).

* the reference refuses more than one prompt (:538-539).  ``BatchedCircuitLens`` lifts that: ONE model forward
  over all prompts, ONE ``tsc_sparse_lens_pass`` per dictionary over all prompts x positions (or one chosen
  position per prompt), one host sync for the whole pass; ``.lens(i)`` hands out per-prompt ``CircuitLens`` views
  whose results equal those of a single-prompt ``CircuitLens`` (what ``CircuitDiscovery`` consumes), and
  ``.direction_lens`` is the batched form of the unembed / logit-difference lens (SURVEY.md section 8f-1,
  BASELINE config 4).

Plotly heat maps are out of scope: ``visualize=True`` prints the ranked labels only.
"""
from __future__ import annotations

import bisect
import os
from dataclasses import dataclass
from pprint import pprint
from typing import Any, Dict, List, Optional, Tuple, TypedDict, Union

import torch

from load_model_and_encoders import get_model_encoders


class CircuitComponent:
    Z_FEATURE = "z_feature"
    MLP_FEATURE = "mlp_feature"
    ATTN_HEAD = "attn_head"
    UNEMBED = "unembed"
    UNEMBED_AT_TOKEN = "unembed_at_token"
    EMBED = "embed"
    POS_EMBED = "pos_embed"
    BIAS_O = "b_O"
    Z_SAE_ERROR = "z_sae_error"
    Z_SAE_BIAS = "z_sae_bias"
    TRANSCODER_ERROR = "transcoder_error"
    TRANSCODER_BIAS = "transcoder_bias"


class LayerKey(TypedDict):
    mlp: int
    attn: int


BIAS_OFFSET = 0
Z_ERROR_OFFSET = 1
Z_BIAS_OFFSET = 2
NUM_ATTN_AUXILARY_FEATURES = 3
NUM_MLP_AUXILARY_FEATURES = 2
NUM_AUXILARY_FEATURES = NUM_ATTN_AUXILARY_FEATURES + NUM_MLP_AUXILARY_FEATURES
_ATTN_AUX = (CircuitComponent.BIAS_O, CircuitComponent.Z_SAE_ERROR, CircuitComponent.Z_SAE_BIAS)
_MLP_AUX = (CircuitComponent.TRANSCODER_ERROR, CircuitComponent.TRANSCODER_BIAS)


@dataclass
class ActiveFeatures:
    """Rows: ``[embed, pos_embed]`` then per layer ``[b_O, z error, z bias, attn features (desc),
    transcoder error, transcoder bias, mlp features (desc)]``."""

    vectors: torch.Tensor   # [comp, d_model]
    values: torch.Tensor    # [comp]
    features: torch.Tensor  # [comp] int, -1 on auxiliary rows
    keys: List[LayerKey]

    # ---- offsets: attn block of layer l starts at _starts()[l], its mlp block at _mlp_starts()[l]
    def _starts(self) -> List[int]:
        out, pos = [], 2
        for key in self.keys:
            out.append(pos)
            pos += key["attn"] + key["mlp"] + NUM_AUXILARY_FEATURES
        out.append(pos)
        return out

    def get_total_active_features(self):
        return sum(key["mlp"] + key["attn"] for key in self.keys)

    def get_total_auxilary_features(self):
        return 2 + NUM_AUXILARY_FEATURES * len(self.keys)

    def get_attn_start_index(self, layer: int):
        return self._starts()[min(layer, len(self.keys))]

    def get_mlp_start_index(self, layer: int):
        return self._starts()[layer] + self.keys[layer]["attn"] + NUM_ATTN_AUXILARY_FEATURES

    def get_vectors_before_comp(self, kind: str, layer: int):
        end = self._starts()[layer]  # layer == n_layers is allowed: everything
        if kind == "mlp":
            end += self.keys[layer]["attn"] + NUM_ATTN_AUXILARY_FEATURES
        return self.vectors[:end]

    def get_z_error(self, layer: int):
        return self.vectors[self.get_attn_start_index(layer) + Z_ERROR_OFFSET]

    @property
    def max_active_features(self):
        return max(max(key["mlp"], key["attn"]) for key in self.keys)

    def get_reconstructed_resid_pre(self, layer: int):
        return self.get_vectors_before_comp("attn", layer).sum(dim=0)

    def get_reconstructed_resid_post(self, layer: int):
        return self.get_vectors_before_comp("attn", layer + 1).sum(dim=0)

    def get_reconstructed_resid_mid(self, layer: int):
        return self.get_vectors_before_comp("mlp", layer).sum(dim=0)

    def get_reconstructed_attn_out(self, layer: int):
        a = self.get_attn_start_index(layer)
        return self.vectors[a: a + self.keys[layer]["attn"] + NUM_ATTN_AUXILARY_FEATURES].sum(dim=0)

    def get_sae_out_reconstruction(self, layer: int):
        a = self.get_attn_start_index(layer) + Z_BIAS_OFFSET  # z bias + attention features
        return self.vectors[a: a + self.keys[layer]["attn"] + 1].sum(dim=0)

    def get_transcoder_reconstruction(self, layer: int):
        m = self.get_mlp_start_index(layer) + 1  # transcoder bias + mlp features
        return self.vectors[m: m + self.keys[layer]["mlp"] + 1].sum(dim=0)

    def get_attn_feature_vectors(self, layer: int):
        a = self.get_attn_start_index(layer) + NUM_ATTN_AUXILARY_FEATURES
        return self.vectors[a: a + self.keys[layer]["attn"]]

    def get_mlp_feature_vectors(self, layer: int):
        m = self.get_mlp_start_index(layer) + NUM_MLP_AUXILARY_FEATURES
        return self.vectors[m: m + self.keys[layer]["mlp"]]

    def get_reconstructed_mlp_out(self, layer: int):
        m = self.get_mlp_start_index(layer)
        return self.vectors[m: m + self.keys[layer]["mlp"] + NUM_MLP_AUXILARY_FEATURES].sum(dim=0)

    def classify_row(self, i: int) -> Tuple[str, int]:
        """Row index -> ``(component kind, layer)``."""
        if i == 0:
            return CircuitComponent.EMBED, 0
        if i == 1:
            return CircuitComponent.POS_EMBED, 0
        starts = self._starts()
        layer = bisect.bisect_right(starts, i) - 1
        off = i - starts[layer]
        n_attn = self.keys[layer]["attn"]
        if off < NUM_ATTN_AUXILARY_FEATURES:
            return _ATTN_AUX[off], layer
        off -= NUM_ATTN_AUXILARY_FEATURES
        if off < n_attn:
            return CircuitComponent.Z_FEATURE, layer
        off -= n_attn
        if off < NUM_MLP_AUXILARY_FEATURES:
            return _MLP_AUX[off], layer
        return CircuitComponent.MLP_FEATURE, layer

    def get_top_k_features(self, activations: torch.Tensor, k=10):
        """``[(component, layer, feature, value)]`` for the k largest activations.  As in the
        reference (:161-238) the list ENDS at the first embed / pos_embed row it meets (the loop
        there ``break``s out entirely on rows 0 and 1); callers see that truncation, so it is kept."""
        k = min(k, activations.size(0))
        values, indices = activations.topk(k=k)
        feats = self.features[indices].tolist()
        out = []
        for v, i, f in zip(values.tolist(), indices.tolist(), feats):
            kind, layer = self.classify_row(i)
            out.append((kind, layer, f, v))
            if i < 2:
                break
        return out

    def get_top_k_lens_runs(self, activations, circuit_lens: "CircuitLens", seq_index: int, k=10) -> List["ComponentLensWithValue"]:
        runs = []
        for component, layer, feature_i, value in self.get_top_k_features(activations, k=k):
            runs.append((ComponentLens(circuit_lens=circuit_lens, component=component,
                                       run_data={"layer": layer, "seq_index": int(seq_index), "feature": feature_i}), value))
        return runs

    def get_top_k_labels(self, activations, k=10):
        return [f"{kind} | Layer: {layer} | Feature: {feature_i} | Contribution: {value*100:.3g}%"
                for kind, layer, feature_i, value in self.get_top_k_features(activations, k=k)]


ComponentLensWithValue = Tuple["ComponentLens", float]


@dataclass
class ComponentLens:
    circuit_lens: "CircuitLens"
    component: str
    run_data: Dict[str, Any]

    @property
    def tuple_id(self):
        d = self.run_data
        if self.component == CircuitComponent.UNEMBED:
            return (self.component, int(d["seq_index"]), int(d["token_id"]))
        if self.component == CircuitComponent.UNEMBED_AT_TOKEN:
            return (self.component, d["seq_index"])
        if self.component == CircuitComponent.ATTN_HEAD:
            return (self.component, int(d["layer"]), int(d["head"]), int(d["source_index"]),
                    int(d["destination_index"]), int(d["feature"]))
        return (self.component, int(d["layer"]), int(d["seq_index"]), int(d["feature"]))

    @classmethod
    def init_from_tuple_id(cls, tuple_id, circuit_lens):
        component = tuple_id[0]
        if component == CircuitComponent.UNEMBED:
            data = {"seq_index": tuple_id[1], "token_id": tuple_id[2]}
        elif component == CircuitComponent.UNEMBED_AT_TOKEN:
            data = {"seq_index": tuple_id[1]}
        elif component == CircuitComponent.ATTN_HEAD:
            data = {"layer": tuple_id[1], "head": tuple_id[2], "source_index": tuple_id[3],
                    "destination_index": tuple_id[4], "feature": tuple_id[5]}
        else:
            data = {"layer": tuple_id[1], "seq_index": tuple_id[2], "feature": tuple_id[3]}
        return cls(circuit_lens=circuit_lens, component=component, run_data=data)

    @classmethod
    def create_unembed_at_token_lens(cls, circuit_lens: "CircuitLens", seq_index):
        return cls(circuit_lens=circuit_lens, component=CircuitComponent.UNEMBED_AT_TOKEN, run_data={"seq_index": seq_index})

    @classmethod
    def create_unembed_lens(cls, circuit_lens: "CircuitLens", seq_index, token):
        if isinstance(token, str):
            token = circuit_lens.model.to_single_token(token)
        return cls(circuit_lens=circuit_lens, component=CircuitComponent.UNEMBED, run_data={"seq_index": seq_index, "token_id": token})

    def __str__(self):
        return f"{self.component} | " + " | ".join(f"{k}: {v}" for k, v in self.run_data.items())

    __repr__ = __str__

    @property
    def feature(self):
        return self.run_data.get("feature", -1)

    def __call__(self, head_type=None, **kwargs) -> Tuple[List[ComponentLensWithValue], Optional[ActiveFeatures]]:
        cl, d = self.circuit_lens, self.run_data
        if self.component == CircuitComponent.UNEMBED:
            return cl.get_unembed_lens(d["token_id"], d["seq_index"], **kwargs)
        if self.component == CircuitComponent.UNEMBED_AT_TOKEN:
            seq_index = cl.process_seq_index(d["seq_index"])
            return cl.get_unembed_lens(cl._token_at(seq_index + 1), d["seq_index"], **kwargs)
        if self.component == CircuitComponent.Z_FEATURE:
            return cl.get_head_seq_lens_for_z_feature(d["layer"], d["seq_index"], d["feature"], **kwargs), None
        if self.component == CircuitComponent.Z_SAE_ERROR:
            return cl.get_head_seq_lens_for_z_error(d["layer"], d["seq_index"], **kwargs), None
        if self.component == CircuitComponent.ATTN_HEAD:
            head_type = head_type or "q"
            if head_type == "q":
                return cl.get_q_lens_on_head_seq(d["layer"], d["head"], d["source_index"], d["destination_index"], **kwargs)
            if head_type == "k":
                return cl.get_k_lens_on_head_seq(d["layer"], d["head"], d["source_index"], d["destination_index"], **kwargs)
            if head_type == "v":
                return cl.get_v_lens_at_seq(d["layer"], d["head"], d["source_index"], d["feature"], **kwargs)
        if self.component == CircuitComponent.MLP_FEATURE:
            return cl.get_mlp_feature_lens_at_seq(d["layer"], d["seq_index"], d["feature"], **kwargs)
        return [], None


class _LazyActiveFeatures:
    """What the fused lens calls hand back in place of ``ActiveFeatures``: the reference returns the position's
    ``ActiveFeatures`` next to the ranked list, its callers in the circuit search drop it (circuit_discovery.py:448, 558),
    and building it -- the ``[components, d_model]`` matrix -- is exactly what the fused path avoids.  Any attribute
    access builds the real object (``CircuitLens.get_active_features``) and forwards to it."""

    def __init__(self, lens: "CircuitLens", seq_index: int):
        self.__dict__["_lens"], self.__dict__["_seq"] = lens, seq_index

    def __getattr__(self, name):
        return getattr(self._lens.get_active_features(self._seq), name)


class _PromptCacheView:
    """One prompt's slice of a batched activation cache, with the batch axis kept (size 1) and the padding
    positions cut off, so that every ``cache[...][0, pos]`` of the single-prompt code reads the same numbers."""

    def __init__(self, cache, index: int, length: int):
        self._cache, self._i, self._n = cache, index, length

    def __getitem__(self, key):
        t = self._cache[key]
        name = key if isinstance(key, str) else key[0]
        if name in ("pattern", "attn_scores") or (isinstance(name, str) and name.endswith(("hook_pattern", "hook_attn_scores"))):
            return t[self._i:self._i + 1, :, :self._n, :self._n]
        return t[self._i:self._i + 1, :self._n]


class _ResidualSpaceZ:
    """A z-SAE as the direction lens sees it: the same encoder, its decoder expressed in the residual stream --
    ``W_dec @ W_O[layer]`` rows and ``b_dec @ W_O[layer]`` -- so that a feature's contribution to a residual direction
    ``d`` is ``act x (W_dec W_O)[f] . d`` and nothing of the pass depends on ``W_O @ d``: the reference projects every
    z vector through W_O before it is scored (src/circuit_lens.py:618-627); folding W_O into the decoder ONCE replaces
    the per-pass pull-back of the directions (a [R, 768] x [768, 768] product per layer).  Quacks like the encoder for
    ``run_dictionary_pass`` (``device_dict``, ``W_dec``, ``b_dec``).  The library handle and the folded decoder are
    rebuilt when a parameter of the z-SAE is re-allocated or updated in place, or when W_O changes."""

    def __init__(self, z_sae, w_o_flat: torch.Tensor):
        self.z_sae = z_sae
        self._w_o = w_o_flat              # [n_heads * d_head, d_model]
        self._w_o_version = w_o_flat._version
        self._handles: dict = {}          # screened -> (key, tsc_hip.DeviceDict)
        self._folded = None               # (key, W_dec W_O, b_dec W_O)

    def same_projection(self, w_o_flat: torch.Tensor) -> bool:
        """Is this the projection the adapter was made for?  (Asked once per batch.  Same storage: yes -- an in-place
        update shows in the version the handle key carries; other storage, e.g. a property that stacks W_O anew on
        every access: compare the values.)"""
        w = self._w_o
        if w.shape != w_o_flat.shape:
            return False
        return w.data_ptr() == w_o_flat.data_ptr() or bool(torch.equal(w, w_o_flat))

    def _key(self):
        z = self.z_sae
        return tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in (z.W_enc, z.b_enc, z.b_dec, z.W_dec, self._w_o))

    def _fold(self, key):
        if self._folded is None or self._folded[0] != key:
            z = self.z_sae
            with torch.no_grad():
                self._folded = (key, torch.matmul(z.W_dec.detach(), self._w_o).contiguous(),
                                torch.mv(self._w_o.t(), z.b_dec.detach()).contiguous())
        return self._folded[1], self._folded[2]

    @property
    def W_dec(self) -> torch.Tensor:
        return self._fold(self._key())[0]

    @property
    def b_dec(self) -> torch.Tensor:  # the ENCODER's input bias (row bounds of the screen): unchanged
        return self.z_sae.b_dec

    @property
    def b_out(self) -> torch.Tensor:  # what the reconstruction adds: b_dec W_O
        return self._fold(self._key())[1]

    def device_dict(self, bf16: bool = False, screened: bool = False):
        import tsc_hip

        z = self.z_sae
        if screened and (bf16 or z.W_enc.shape[0] % 128 != 0 or z.W_enc.shape[0] > 768 or self._w_o.shape[1] > 768):
            screened = False
        key = self._key()
        cur = self._handles.get(screened)
        if cur is None or cur[0] != key:
            z._require_gpu_fp32()
            W_r, b_r = self._fold(key)
            if cur is not None:
                cur[1].close()
            h = tsc_hip.DeviceDict(z.W_enc.detach().contiguous(), z.b_enc.detach().contiguous(), z.b_dec.detach().contiguous(),
                                   W_r, b_r, compute_dtype=tsc_hip.F32_SCREEN if screened else tsc_hip.F32)
            cur = self._handles[screened] = (key, h)
        return cur[1]


def _residual_z(z_sae, w_o_flat: torch.Tensor) -> _ResidualSpaceZ:
    """The z-SAE's residual-space view for this W_O (kept on the module: one fold per model, not per batch or lens)."""
    adapter = z_sae.__dict__.get("_resid_adapter")
    if adapter is None or not adapter.same_projection(w_o_flat):
        adapter = _ResidualSpaceZ(z_sae, w_o_flat)
        object.__setattr__(z_sae, "_resid_adapter", adapter)
    return adapter

_PASS_STREAMS: Dict[str, list] = {}


_TRACE = [] if os.environ.get("TSC_LENS_TRACE") else None  # profiling aid: (label, event on the current stream, host time)


def _mark(label: str) -> None:
    if _TRACE is not None:
        import time

        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        _TRACE.append((label, ev, time.perf_counter()))


def _pass_streams(dev, n: int = 4):
    key = str(dev)
    if key not in _PASS_STREAMS:
        _PASS_STREAMS[key] = [torch.cuda.Stream(device=dev) for _ in range(n)]
    return _PASS_STREAMS[key]


def run_dictionary_pass(encoders, inputs, max_active: int = 512, dirs=None, want_recon: bool = True,
                        screened: bool = True, row_bounds: Optional[torch.Tensor] = None, then=None,
                        handle_cache: Optional[dict] = None):
    """All dictionaries of ``encoders`` (a list) over their ``inputs`` (a list of ``[R, d_in]`` tensors, the
    same R for all) with ONE host synchronisation: the counts of every dictionary live in one tensor that is
    zeroed once, checked against the capacity once and copied to the host once, and the outputs of all
    dictionaries are slices of stacked buffers (no per-dictionary allocation).  ``dirs``: optional ``[D, R, d_out]``
    tensor (or list) of per-dictionary, per-row directions for the fused contribution scores -- or a callable that
    returns them: it is then called AFTER the emits have been queued (they do not need the directions), so that the
    device work behind the directions and the host time of preparing them hide under the emit kernel.
    ``screened``: use the handles that screen the emit with the bf16 MFMA and recompute the candidates exactly in
    fp32 (``TSC_F32_SCREEN``, include/tsc_hip.h) -- the same fp32 results, several times faster at CircuitLens sizes;
    ``row_bounds [D, 2]`` or ``[D, 2 + 2 R]`` (with every row's own pair) are the screens' row bounds
    (``tsc_hip.screen_row_bound``), computed here when not given.
    ``then(feat, val, count, recon, scores)``: optional device-only continuation, launched before the counts are
    read (its result is appended to the returned tuple).
    ``handle_cache``: a dict the caller keeps between passes over the SAME encoders and inputs.  With it the library
    handles (24 look-ups, each checking five parameter addresses and versions) and the marshalled call tables are
    reused WITHOUT validation on the way in -- the GPU idles until the emit is queued -- and validated while the
    GPU works: a handle borrows tensors it keeps alive, so a stale one is harmless to launch, and a pass that ran on
    stale handles (a parameter was re-allocated or updated in place since) is simply run again.
    Returns ``(feat [D,R,cap], val [D,R,cap], count [D,R], recon [D,R,d_out] or None, scores [D,R,cap] or None,
    counts on host [D,R])``."""
    import tsc_hip

    n_d = len(encoders)
    rows = inputs[0].shape[0]
    dev = inputs[0].device
    cap = max(1, min(int(max_active), tsc_hip.LENS_PASS_MAX_ACTIVE))
    screened = screened and rows <= tsc_hip.SCREEN_MAX_ROWS

    def look_up():
        hs = [enc.device_dict(screened=screened) for enc in encoders]
        return hs, [h._versions for h in hs]

    cached = handle_cache.get(screened) if handle_cache is not None else None
    unvalidated = cached is not None
    if cached is None:
        handles, versions = look_up()
        cached = {"handles": handles, "versions": versions, "group": None, "plan": None}
        if handle_cache is not None:
            handle_cache[screened] = cached
    handles = cached["handles"]
    if screened and row_bounds is None:
        row_bounds = torch.stack([tsc_hip.screen_row_bound(x, enc.b_dec.detach(), per_row=True) for enc, x in zip(encoders, inputs)])
    tail = None
    dirs_fn = dirs if callable(dirs) else None
    while True:
        # Everything between here and the emit's launch is host time with the GPU idle: the shape of the pass (which
        # calls, which streams) is decided once per set of handles, and only the emit's own buffers are allocated
        # before it is queued.
        plan = cached.get("plan")
        if plan is None or plan[0] != (rows, n_d):
            # The dictionaries are independent and, at CircuitLens sizes, each one's two launches are too short to fill
            # the chip: with several dictionaries, stop each after its emit and finish them all in ONE launch
            # (tsc_sparse_lens_finish_many): one dictionary's 512 rows are two workgroups per CU, 24 x 512 fill the chip.
            grouped = 1 < n_d <= 32 and rows <= 8192 and len({(h.compute_dtype, h.d_in, h.d_out) for h in handles}) == 1 and \
                len({id(h) for h in handles}) == n_d and os.environ.get("TSC_LENS_GROUPED", "1") != "0"
            n_side = int(os.environ.get("TSC_LENS_STREAMS", "4"))  # 1: everything on the current stream (profiling)
            # (the grouped screened emit is one launch on the current stream: no side streams, no cross-stream events)
            one_launch = grouped and screened and handles[0].compute_dtype == tsc_hip.F32_SCREEN
            use_side = rows <= 8192 and n_d > 1 and n_side > 1 and not one_launch
            plan = cached["plan"] = ((rows, n_d), grouped, one_launch, use_side, n_side, handles[0].d_out)
        _, grouped, one_launch, use_side, n_side, d_out = plan
        feat = torch.empty(n_d, rows, cap, dtype=torch.int32, device=dev)
        val = torch.empty(n_d, rows, cap, dtype=torch.float32, device=dev)
        # (the grouped emit call zeroes the counts itself: one memset queued by the library, no torch launch)
        count = torch.empty(n_d, rows, dtype=torch.int32, device=dev) if grouped else torch.zeros(n_d, rows, dtype=torch.int32, device=dev)
        recon = scores = None
        if not grouped:
            recon = torch.empty(n_d, rows, d_out, dtype=torch.float32, device=dev) if want_recon else None
            scores = torch.empty(n_d, rows, cap, dtype=torch.float32, device=dev) if dirs is not None else None
        # Every buffer is allocated on the calling stream and outlives the pass; the side streams start after the
        # calling stream's work so far and the calling stream waits for all of them before the counts are read.
        main = torch.cuda.current_stream(dev)
        side = _pass_streams(dev, n_side) if use_side else []
        _mark("pass: buffers ready")
        if side:
            start = torch.cuda.Event()
            start.record(main)
        if dirs_fn is not None and not grouped:
            dirs = dirs_fn()  # the fused per-dictionary calls take their directions with the emit
        if grouped:  # every emit in one foreign call (24 separate calls cost more host time than their kernels run)
            for stream in side:
                stream.wait_event(start)
            group = cached["group"]
            if group is None or len(group.inputs) != n_d or any(a is not b for a, b in zip(group.inputs, inputs)):
                group = cached["group"] = tsc_hip.LensGroup(handles, inputs)
            group.emit(feat, val, count, screen=screened,
                       row_bounds=row_bounds if screened and handles[0].compute_dtype == tsc_hip.F32_SCREEN else None,
                       streams=side or None, count_is_zero=False)
            # (allocated while the emit runs)
            recon = torch.empty(n_d, rows, d_out, dtype=torch.float32, device=dev) if want_recon else None
            scores = torch.empty(n_d, rows, cap, dtype=torch.float32, device=dev) if dirs is not None else None
        for i, (h, x) in enumerate(zip(handles, inputs) if not grouped else ()):
            stream = side[i % len(side)] if side else main
            if side and i < len(side):
                stream.wait_event(start)
            with torch.cuda.stream(stream):
                h.sparse_lens_pass(x, cap, dirs=None if dirs is None else dirs[i], screen=screened,
                                   row_bound=row_bounds[i] if screened and h.compute_dtype == tsc_hip.F32_SCREEN else None,
                                   out={"feat": feat[i], "val": val[i], "count": count[i],
                                        "recon": recon[i] if want_recon else None,
                                        "scores": scores[i] if scores is not None else None},
                                   want_recon=want_recon, emit_only=grouped)
        for stream in side:
            main.wait_stream(stream)
        _mark("pass: emits done")
        if grouped:
            if dirs_fn is not None:
                dirs = dirs_fn()
            group.finish(feat, val, count, dirs=dirs, recon=recon, scores=scores, screen=screened)
        # work that only consumes this pass's device buffers is queued BEFORE the host looks at the counts, so the
        # GPU never drains in the middle of the pass; it is redone in the (rare) retry with a larger capacity
        _mark("pass: finish done")
        if then is not None:
            tail = then(feat, val, count, recon, scores)
        _mark("pass: tail queued")
        if unvalidated:  # the look-ups that were skipped on the way in, now that the GPU has work
            unvalidated = False
            fresh, versions = look_up()
            # (a refresh -- in-place parameter update -- replaces the handle's version tuple)
            if any(f is not h for f, h in zip(fresh, handles)) or any(v is not w for v, w in zip(versions, cached["versions"])):
                cached.update(handles=fresh, versions=versions, group=None, plan=None)
                handles = fresh
                continue  # the pass ran on handles of older parameters: again, with the current ones
        # the one synchronisation of the pass (a pinned buffer + non-blocking copy + stream synchronise measured slower
        # than this blocking copy: 57 against 34 us from the tail's last launch to the return)
        counts_host = count.cpu()
        _mark("pass: counts on host")
        worst = int(counts_host.max()) if counts_host.numel() else 0
        if worst <= cap:
            if then is not None:
                return feat, val, count, recon, scores, counts_host, tail
            return feat, val, count, recon, scores, counts_host
        if cap >= tsc_hip.LENS_PASS_MAX_ACTIVE:
            raise tsc_hip.TscError(f"a position has {worst} active features in one dictionary, above the "
                                   f"{tsc_hip.LENS_PASS_MAX_ACTIVE} the fused lens pass holds")
        cap = min(tsc_hip.LENS_PASS_MAX_ACTIVE, max(2 * cap, worst))


class CircuitLens:
    def __init__(self, prompt, _batch: Optional["BatchedCircuitLens"] = None, _index: int = 0):
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        model, z_saes, mlp_transcoders = get_model_encoders(self.device)
        self.z_saes = z_saes
        self.mlp_transcoders = mlp_transcoders
        self.model = model
        self.prompt = prompt
        self._batch, self._index = _batch, _index
        if _batch is not None:  # a per-prompt view of a batched pass (BatchedCircuitLens.lens)
            n = _batch.lengths[_index]
            self.tokens = _batch.tokens[_index:_index + 1, :n]
            self.logits = _batch.logits[_index:_index + 1, :n]
            self.cache = _PromptCacheView(_batch.cache, _index, n)
        else:
            self.tokens = self.model.to_tokens(prompt)
            if self.tokens.size(0) != 1:
                raise ValueError("Can only do CircuitLens on a single prompt!")  # many prompts: BatchedCircuitLens
            with torch.no_grad():
                self.logits, self.cache = self.model.run_with_cache(self.tokens, return_type="logits")
        self._seq_activations: Dict[int, ActiveFeatures] = {}
        self._pass = None  # (feat, val, count, recons, counts on host), rows = positions of this prompt

    @property
    def n_tokens(self):
        return self.tokens.size(1)

    def normalize_activations(self, activations):
        return activations / activations.sum()

    def process_seq_index(self, seq_index):
        if seq_index < 0:
            seq_index += self.n_tokens
        return seq_index

    # ------------------------------------------------------------------ batched dictionary passes
    def _dictionary_input(self, kind: str, layer: int) -> torch.Tensor:
        if kind == "attn":
            z = self.cache["z", layer][0]
            return z.reshape(z.shape[0], -1).contiguous()           # "n_heads d_head -> (n_heads d_head)" per position
        return self.cache["normalized", layer, "ln2"][0].contiguous()

    def _dict_index(self, kind: str, layer: int) -> int:
        return 2 * layer + (0 if kind == "attn" else 1)

    def _ensure_pass(self):
        """All 2 x n_layers dictionaries over all positions of the prompt, one host sync in total."""
        if self._pass is None:
            if self._batch is not None:
                self._pass = self._batch._pass_for_prompt(self._index)
            else:
                encs, xs = [], []
                for layer in range(self.model.cfg.n_layers):
                    encs += [self.z_saes[layer], self.mlp_transcoders[layer]]
                    xs += [self._dictionary_input("attn", layer), self._dictionary_input("mlp", layer)]
                with torch.no_grad():
                    feat, val, count, recons, _, counts_host = run_dictionary_pass(encs, xs)
                self._pass = (feat, val, count, recons, counts_host.tolist(), None)
        return self._pass

    def _sparse(self, kind: str, layer: int):
        """``(feat [n,cap], val [n,cap], count [n], recon [n,d], counts on host)`` for the positions covered."""
        feat, val, count, recons, counts, _ = self._ensure_pass()
        i = self._dict_index(kind, layer)
        return feat[i], val[i], count[i], recons[i], counts[i]

    def _row(self, seq_index: int) -> int:
        rows = self._ensure_pass()[5]
        if rows is None:
            return seq_index
        r = rows.get(seq_index)
        if r is None:
            raise ValueError(f"position {seq_index} is not covered by the batched pass this view belongs to "
                             "(build the BatchedCircuitLens with positions='all')")
        return r

    def _active(self, kind: str, layer: int, seq_index: int):
        feat, val, _, recon, counts = self._sparse(kind, layer)
        r = self._row(seq_index)
        c = counts[r]
        return feat[r, :c], val[r, :c], recon[r]

    def _w_o_flat(self, layer: int) -> torch.Tensor:
        w = self.model.W_O[layer]
        return w.reshape(w.shape[0] * w.shape[1], w.shape[2])

    def get_active_features(self, seq_index: int, cache=True) -> ActiveFeatures:
        seq_index = self.process_seq_index(seq_index)
        hit = self._seq_activations.get(seq_index)
        if cache and hit is not None:
            return hit
        dev = self.cache["z", 0].device
        ones = lambda n: torch.ones(n, device=dev)                                    # noqa: E731
        minus = lambda n: torch.full((n,), -1, dtype=torch.int64, device=dev)         # noqa: E731
        vectors = [self.cache["embed"][0, seq_index][None], self.cache["pos_embed"][0, seq_index][None]]
        values, features, keys = [ones(2)], [minus(2)], []
        with torch.no_grad():
            for layer in range(self.model.cfg.n_layers):
                z_sae, tc = self.z_saes[layer], self.mlp_transcoders[layer]
                w_o = self._w_o_flat(layer)
                zf, zv, z_recon = self._active("attn", layer, seq_index)
                layer_z = self._dictionary_input("attn", layer)[seq_index]
                aux = torch.stack([layer_z - z_recon, z_sae.b_dec]) @ w_o
                z_rows = z_sae.device_dict().decode_rows(zf, zv) @ w_o if zf.numel() else aux.new_zeros(0, aux.shape[1])
                vectors += [self.model.b_O[layer][None], aux, z_rows]
                values += [ones(NUM_ATTN_AUXILARY_FEATURES), zv]
                features += [minus(NUM_ATTN_AUXILARY_FEATURES), zf.to(torch.int64)]

                mf, mv, m_recon = self._active("mlp", layer, seq_index)
                mlp_error = self.cache["mlp_out", layer][0, seq_index] - m_recon
                m_rows = tc.device_dict().decode_rows(mf, mv) if mf.numel() else aux.new_zeros(0, aux.shape[1])
                vectors += [mlp_error[None], tc.b_dec_out[None], m_rows]
                values += [ones(NUM_MLP_AUXILARY_FEATURES), mv]
                features += [minus(NUM_MLP_AUXILARY_FEATURES), mf.to(torch.int64)]
                keys.append({"mlp": int(mf.numel()), "attn": int(zf.numel())})
        af = ActiveFeatures(vectors=torch.cat(vectors, dim=0), values=torch.cat(values, dim=0),
                            features=torch.cat(features, dim=0), keys=keys)
        self._seq_activations[seq_index] = af
        return af

    # ------------------------------------------------------------------ fused per-node lens (no [comp, d_model] matrix)
    FUSED_LENS = os.environ.get("TSC_LENS_FUSED", "1") != "0"

    def _fused_state(self):
        """Persistent device buffers and marshalled call tables of :meth:`_fused_lens` (made on first use)."""
        st = self.__dict__.get("_fused")
        if st is None:
            import tsc_hip

            feat, val, count, recons, _, _ = self._ensure_pass()
            n_layers = self.model.cfg.n_layers
            dev, cap, dm = feat.device, feat.shape[2], self.model.W_U.shape[0]
            w_o = [self._w_o_flat(l) for l in range(n_layers)]                           # [n*d, d_model] per layer
            ok = all(w.shape[1] == dm for w in w_o) and all(t.W_dec.shape[1] == dm for t in self.mlp_transcoders[:n_layers])
            st = {"ok": ok}
            if ok:
                # every dictionary decodes into the residual stream (the z-SAEs through W_O, folded once: _ResidualSpaceZ),
                # so a query meets ONE direction and does no W_O @ d product; what the auxiliary rows need of the z side
                # -- z W_O and the z reconstructions through W_O -- is projected once per prompt
                with torch.no_grad():
                    z_views = [_residual_z(self.z_saes[l], w_o[l]) for l in range(n_layers)]
                    handles = []
                    for l in range(n_layers):
                        handles += [z_views[l].device_dict(), self.mlp_transcoders[l].device_dict()]
                    z_resid = [(self._dictionary_input("attn", l) @ w_o[l]).contiguous() for l in range(n_layers)]
                    recon_resid = [(recons[2 * l] @ w_o[l]).contiguous() for l in range(n_layers)]
                st.update(
                    group=tsc_hip.ContribGroup(handles), n_layers=n_layers, cap=cap,
                    d=torch.empty(dm, dtype=torch.float32, device=dev),
                    scores=torch.empty(2 * n_layers, 1, cap, dtype=torch.float32, device=dev),
                    feat=torch.empty(2 * n_layers, 1, cap, dtype=torch.int32, device=dev),
                    count=torch.empty(2 * n_layers, 1, dtype=torch.int32, device=dev),
                    head=torch.empty(1, 2, dtype=torch.float32, device=dev),
                    aux=torch.empty(1, 5 * n_layers, dtype=torch.float32, device=dev),
                    b_o=torch.stack([self.model.b_O[l] for l in range(n_layers)]).contiguous(),
                    z_b_out=[v.b_out for v in z_views],                                   # b_dec W_O
                    t_b_out=torch.stack([self.mlp_transcoders[l].b_dec_out.detach() for l in range(n_layers)]).contiguous(),
                    z_resid=z_resid, recon_resid=recon_resid, z_views=z_views,           # kept: the tables point into them
                    tables={}, tags={})
                names = ["scores", "feat", "count", "head", "aux"]
                width = 2 + 5 * n_layers + 2 * n_layers * cap
                if width > 15360:  # rows beyond tsc_lens_assemble's LDS are ranked out of its dense outputs
                    st.update(raw=torch.empty(1, width, dtype=torch.float32, device=dev),
                              acts=torch.empty(1, width, dtype=torch.float32, device=dev),
                              mask=torch.empty(1, width, dtype=torch.bool, device=dev))
                    names += ["raw", "acts", "mask"]
                st["query"] = tsc_hip.LensQuery(st["group"], feat, val, count, {n: st[n] for n in names}, [3, 2] * n_layers)
                ptrs = st["group"].ds
                for g in range(2 * n_layers):
                    ptrs[g] = st["d"].data_ptr()
            self.__dict__["_fused"] = st
        return st

    def _fused_tags(self, st, n_groups: int):
        """``tsc_lens_assemble``'s column labels for the first ``n_groups`` dictionaries (component << 16 | layer)."""
        t = st["tags"].get(n_groups)
        if t is None:
            tags = [0 << 16, 1 << 16]
            for g in range(n_groups):
                layer = g // 2
                tags += [(c << 16) | layer for c in ((2, 3, 4) if g % 2 == 0 else (6, 7))]
            tags += [((5 if g % 2 == 0 else 8) << 16) | (g // 2) for g in range(n_groups)]
            t = st["tags"][n_groups] = torch.tensor(tags, dtype=torch.int32, device=st["d"].device)
        return t

    def _fused_lens(self, direction: torch.Tensor, seq_index: int, before: Optional[Tuple[str, int]], k: int):
        """``normalize_activations(vectors_before(kind, layer) @ direction)`` of position ``seq_index`` and its top-k as
        ``[(component, layer, feature, value)]`` -- the list ``ActiveFeatures.get_top_k_features`` returns, including its
        stop at the first embed / pos_embed row -- in four launches and one device-to-host copy: the direction's copy,
        ``tsc_contrib_scores_many`` (every dictionary's active features at this position, all in residual space),
        ``tsc_rowdot_many`` (the auxiliary rows), ``tsc_lens_assemble`` over the dictionaries in front of the cut-off.  Returns None when the
        model's shapes rule the fused path out (the caller then takes the ``vectors`` path)."""
        import tsc_hip

        st = self._fused_state()
        n_layers = self.model.cfg.n_layers
        n_groups = 2 * n_layers if before is None else 2 * before[1] + (1 if before[0] == "mlp" else 0)
        if not st["ok"] or n_groups == 0 or k > 64:
            return None
        feat, val, count, recons, counts, _ = self._ensure_pass()
        r = self._row(seq_index)
        # lists just long enough for THIS position (one crowded position of the prompt dictates the pass's capacity)
        out_cap = (max(1, max(counts[g][r] for g in range(n_groups))) + 63) // 64 * 64
        with torch.no_grad():
            st["d"].copy_(direction.reshape(-1))
            tab = st["tables"].get(seq_index)
            if tab is None:
                dm = st["d"].shape[0]
                tab = tsc_hip.RowDotTable(2 + 5 * n_layers, 1, dm)
                row = lambda t: t.reshape(1, dm)  # noqa: E731
                head, aux, d = st["head"], st["aux"], st["d"]
                tab.set(0, row(self.cache["embed"][0, seq_index]), None, d, head[:, 0])
                tab.set(1, row(self.cache["pos_embed"][0, seq_index]), None, d, head[:, 1])
                for l in range(n_layers):
                    j = 2 + 5 * l
                    tab.set(j, st["b_o"][l], None, d, aux[:, 5 * l])
                    tab.set(j + 1, row(st["z_resid"][l][seq_index]), row(st["recon_resid"][l][r]), d, aux[:, 5 * l + 1])
                    tab.set(j + 2, st["z_b_out"][l], None, d, aux[:, 5 * l + 2])
                    tab.set(j + 3, row(self.cache["mlp_out", l][0, seq_index]), row(recons[2 * l + 1][r]), d, aux[:, 5 * l + 3])
                    tab.set(j + 4, st["t_b_out"][l], None, d, aux[:, 5 * l + 4])
                st["tables"][seq_index] = tab
            # scores of the position's active features, its auxiliary rows, normalise + rank: three library calls, one
            # copy to pinned host memory, one synchronisation
            host = st["query"].run(r, n_groups, k, tab, self._fused_tags(st, n_groups), out_cap)
        info = host[:4 * k].view(k, 4).tolist()
        values = host[4 * k:].view(torch.float32).tolist()
        out = []
        for (col, comp, layer, feature), v in zip(info, values):
            if col < 0:
                break  # fewer than k components
            out.append((_COMPONENT_ORDER[comp], layer, feature, v))
            if comp < 2:
                break  # the reference's loop ends at the first embed / pos_embed row (circuit_lens.py:171-176)
        return out

    def _fused_runs(self, direction, seq_index: int, before, k):
        """The fused lens as the reference's lens functions return it: ``([(ComponentLens, value)], active features)``."""
        top = self._fused_lens(direction, seq_index, before, 10 if k is None else k)
        if top is None:
            return None
        runs = [(ComponentLens(circuit_lens=self, component=component,
                               run_data={"layer": layer, "seq_index": int(seq_index), "feature": feature}), value)
                for component, layer, feature, value in top]
        return runs, _LazyActiveFeatures(self, seq_index)

    # ------------------------------------------------------------------ contribution scores (sparse form)
    def contribution_scores(self, kind: str, layer: int, directions: torch.Tensor, seq_index: Optional[int] = None):
        """``feature_act x W_dec row . direction`` for the active features of ``layer``'s dictionary
        at every position (or one): returns ``(feat, scores [n, cap, n_dir], count)``.  For attention
        dictionaries the W_O projection is folded into the directions."""
        enc = self.z_saes[layer] if kind == "attn" else self.mlp_transcoders[layer]
        feat, val, count, _, _ = self._sparse(kind, layer)
        dirs = directions.reshape(-1, directions.shape[-1]).to(torch.float32)
        if kind == "attn":
            dirs = dirs @ self._w_o_flat(layer).t()
        if seq_index is not None:
            s = self._row(self.process_seq_index(seq_index))
            feat, val, count = feat[s:s + 1].contiguous(), val[s:s + 1].contiguous(), count[s:s + 1].contiguous()
        return feat, enc.device_dict().contrib_scores(feat, val, count, dirs.contiguous()), count

    # ------------------------------------------------------------------ lens plumbing
    def get_next_lens_runs(self, active_features: ActiveFeatures, activations, seq_index: int, title: str,
                           visualize=True, k=None):
        if k is None:
            k = 10
        if visualize:
            print(title)
            pprint(active_features.get_top_k_labels(activations, k=k))
        return active_features.get_top_k_lens_runs(activations, self, seq_index, k=k), active_features

    def get_unembed_lens_for_prompt_token(self, seq_index: int, visualize=True, k=None):
        seq_index = self.process_seq_index(seq_index)
        return self.get_unembed_lens(self._token_at(seq_index + 1), seq_index, visualize, k)

    def _token_at(self, index: int) -> int:
        """Token id at a position, from a host copy made once (``.item()`` would synchronise on every lens call)."""
        host = self.__dict__.get("_tokens_host")
        if host is None:
            host = self.__dict__["_tokens_host"] = self.tokens[0].tolist()
        return host[index]

    def get_str_token_at_seq(self, seq_index: int):
        return self.model.tokenizer.decode([self.tokens[0, self.process_seq_index(seq_index)]])

    def get_unembed_lens(self, token_i: Union[int, str], seq_index: int, visualize=True, k=None):
        if isinstance(token_i, str):
            token_i = self.model.to_tokens(token_i, prepend_bos=False)[0]
        seq_index = self.process_seq_index(seq_index)
        if self.FUSED_LENS and not visualize:
            fused = self._fused_runs(self.model.W_U[:, token_i], seq_index, None, k)
            if fused is not None:
                return fused
        af = self.get_active_features(seq_index)
        activations = self.normalize_activations(af.vectors @ self.model.W_U[:, token_i])
        return self.get_next_lens_runs(af, activations, seq_index,
                                       title=f"Unembed Lens for token '{self.model.tokenizer.decode([token_i])}'/{token_i}",
                                       visualize=visualize, k=k)

    def get_direction_lens(self, direction: torch.Tensor, seq_index: int, visualize=True, k=None):
        """``get_unembed_lens`` for an arbitrary residual direction (e.g. ``TaskEvaluation.logit_difference``)."""
        seq_index = self.process_seq_index(seq_index)
        if self.FUSED_LENS and not visualize:
            fused = self._fused_runs(direction.to(self.model.W_U.device, torch.float32), seq_index, None, k)
            if fused is not None:
                return fused
        af = self.get_active_features(seq_index)
        activations = self.normalize_activations(af.vectors @ direction.to(af.vectors.device, af.vectors.dtype).reshape(-1))
        return self.get_next_lens_runs(af, activations, seq_index, title="Direction Lens", visualize=visualize, k=k)

    def get_head_seq_activations_for_vector(self, layer: int, seq_index: int, vector: torch.Tensor):
        """Contribution of every (head, source position) to ``vector`` read from z at ``seq_index``:
        ``pattern[h, seq, src] * (v[src, h] . vector[h])`` flattened head-major."""
        seq_index = self.process_seq_index(seq_index)
        v = self.cache["v", layer][0]
        pattern = self.cache["pattern", layer][0, :, seq_index]
        return (torch.einsum("shd,hd->hs", v, vector) * pattern).reshape(-1)

    def get_head_seq_activations_for_z_feature(self, layer: int, seq_index: int, feature: int):
        w = self.z_saes[layer].W_enc[:, feature].reshape(12, -1)
        return self.get_head_seq_activations_for_vector(layer, seq_index, w)

    def get_head_seq_lens_for_activations(self, feature_act, layer: int, seq_index: int, feature: int,
                                          visualize=True, k=10) -> List[ComponentLensWithValue]:
        feature_act = self.normalize_activations(feature_act)
        values, indices = feature_act.topk(k=k)
        runs = []
        for index, value in zip(indices.tolist(), values.tolist()):
            head, source = index // self.n_tokens, index % self.n_tokens
            runs.append((ComponentLens(circuit_lens=self, component=CircuitComponent.ATTN_HEAD,
                                       run_data={"layer": layer, "head": head, "source_index": int(source),
                                                 "feature": feature, "destination_index": int(seq_index)}), value))
        if visualize:
            pprint([f"Head {r.run_data['head']} | Source: {r.run_data['source_index']} | Destination: "
                    f"{r.run_data['destination_index']} | Contribution: {v * 100:.3g}%" for r, v in runs])
        return runs

    def get_head_seq_lens_for_z_feature(self, layer: int, seq_index: int, feature: int, visualize=True, k=10):
        act = self.get_head_seq_activations_for_z_feature(layer, seq_index, feature)
        return self.get_head_seq_lens_for_activations(act, layer, seq_index, feature, visualize=visualize, k=k)

    def get_head_seq_lens_for_z_error(self, layer: int, seq_index: int, visualize=True, k=10):
        seq_index = self.process_seq_index(seq_index)
        err = self.get_active_features(seq_index).get_z_error(layer).reshape(12, -1)
        act = self.get_head_seq_activations_for_vector(layer, seq_index, err)
        return self.get_head_seq_lens_for_activations(act, layer, seq_index, -1, visualize=visualize, k=k)

    def get_mlp_feature_lens_at_seq(self, layer: int, seq_index: int, feature: int, visualize=True, k=None):
        seq_index = self.process_seq_index(seq_index)
        if self.FUSED_LENS and not visualize:
            fused = self._fused_runs(self.mlp_transcoders[layer].W_enc[:, feature], seq_index, ("mlp", layer), k)
            if fused is not None:
                return fused
        af = self.get_active_features(seq_index)
        vectors = af.get_vectors_before_comp("mlp", layer)
        activation = self.normalize_activations(vectors @ self.mlp_transcoders[layer].W_enc[:, feature])
        return self.get_next_lens_runs(af, activation, seq_index,
                                       title=f"MLP Lens for Feature {feature} at '{self.get_str_token_at_seq(seq_index)}'/{seq_index}",
                                       visualize=visualize, k=k)

    def get_v_lens_activations(self, layer: int, head, seq_index: int, feature: int):
        seq_index = self.process_seq_index(seq_index)
        vectors = self.get_active_features(seq_index).get_vectors_before_comp("attn", layer)
        d_head = self.model.cfg.d_head if hasattr(self.model.cfg, "d_head") else self.model.W_V.shape[-1]
        eff_feature = self.z_saes[layer].W_enc[head * d_head:(head + 1) * d_head, feature]
        return (vectors @ self.model.W_V[layer, head]) @ eff_feature

    def get_v_lens_at_seq(self, layer: int, head: int, seq_index: int, feature: int, visualize=True, k=None):
        seq_index = self.process_seq_index(seq_index)
        if self.FUSED_LENS and not visualize:  # (vectors @ W_V) @ w = vectors @ (W_V @ w)
            d_head = self.model.W_V.shape[-1]
            w = self.z_saes[layer].W_enc[head * d_head:(head + 1) * d_head, feature]
            fused = self._fused_runs(self.model.W_V[layer, head] @ w, seq_index, ("attn", layer), k)
            if fused is not None:
                return fused
        af = self.get_active_features(seq_index)
        activation = self.normalize_activations(self.get_v_lens_activations(layer, head, seq_index, feature))
        return self.get_next_lens_runs(af, activation, seq_index,
                                       title=f"V Lens | Layer {layer} | Head {head} | Feature {feature} at {seq_index}",
                                       visualize=visualize, k=k)

    def get_q_lens_activations(self, layer: int, head: int, source_index, destination_index):
        source_index = self.process_seq_index(source_index)
        destination_index = self.process_seq_index(destination_index)
        vectors = self.get_active_features(destination_index).get_vectors_before_comp("attn", layer)
        return (vectors @ self.model.W_Q[layer, head]) @ self.cache["k", layer][0, source_index, head]

    def get_q_lens_on_head_seq(self, layer: int, head: int, source_index, destination_index, visualize=True, k=None):
        source_index = self.process_seq_index(source_index)
        destination_index = self.process_seq_index(destination_index)
        if self.FUSED_LENS and not visualize:
            direction = self.model.W_Q[layer, head] @ self.cache["k", layer][0, source_index, head]
            fused = self._fused_runs(direction, destination_index, ("attn", layer), k)
            if fused is not None:
                return fused
        af = self.get_active_features(destination_index)
        activation = self.normalize_activations(self.get_q_lens_activations(layer, head, source_index, destination_index))
        return self.get_next_lens_runs(af, activation, destination_index,
                                       title=f"Q Lens | Layer {layer} | Head {head} | {source_index} -> {destination_index}",
                                       visualize=visualize, k=k)

    def get_k_lens_activations(self, layer: int, head: int, source_index, destination_index):
        source_index = self.process_seq_index(source_index)
        destination_index = self.process_seq_index(destination_index)
        vectors = self.get_active_features(source_index).get_vectors_before_comp("attn", layer)
        return (vectors @ self.model.W_K[layer, head]) @ self.cache["q", layer][0, destination_index, head]

    def get_k_lens_on_head_seq(self, layer: int, head: int, source_index, destination_index, visualize=True, k=None):
        source_index = self.process_seq_index(source_index)
        destination_index = self.process_seq_index(destination_index)
        if self.FUSED_LENS and not visualize:
            direction = self.model.W_K[layer, head] @ self.cache["q", layer][0, destination_index, head]
            fused = self._fused_runs(direction, source_index, ("attn", layer), k)
            if fused is not None:
                return fused
        af = self.get_active_features(source_index)
        activation = self.normalize_activations(self.get_k_lens_activations(layer, head, source_index, destination_index))
        return self.get_next_lens_runs(af, activation, source_index,
                                       title=f"K Lens | Layer {layer} | Head {head} | {source_index} -> {destination_index}",
                                       visualize=visualize, k=k)

    # ------------------------------------------------------------------ input-x-gradient style pull-backs
    def get_transcoder_ixg(self, transcoder, layer, seq_index, feature_vector, is_transcoder_post_ln=True,
                           return_feature_activs=True):
        """``(W_dec @ feature_vector) * feature_acts`` over all ``d_sae`` features (reference
        :1138-1170).  The activations are sparse, so only the active features' ``W_dec`` rows are
        read (``tsc_contrib_scores``) and scattered into a zero vector; inactive entries are
        exactly 0 on both sides."""
        if not is_transcoder_post_ln:
            raise NotImplementedError("pre-LN transcoder inputs (resid_mid) are not cached in the batched pass")
        seq_index = self.process_seq_index(seq_index)
        if transcoder is not self.mlp_transcoders[layer]:
            raise ValueError("get_transcoder_ixg expects the transcoder of the given layer")
        feat, scores, count = self.contribution_scores("mlp", layer, feature_vector.reshape(1, -1), seq_index)
        c = int(count[0])
        d_sae = transcoder.W_enc.shape[1]
        pulled = torch.zeros(d_sae, device=scores.device)
        pulled[feat[0, :c].long()] = scores[0, :c, 0]
        if not return_feature_activs:
            return pulled
        mf, mv, _ = self._active("mlp", layer, seq_index)
        acts = torch.zeros(d_sae, device=scores.device)
        acts[mf.long()] = mv
        return pulled, acts

    def get_ln_constant(self, vector, layer, token, is_ln2=False, recip=False):
        x = self.cache[("resid_mid", layer) if is_ln2 else ("resid_pre", layer)][0, token]
        y = self.cache[("normalized", layer, "ln2") if is_ln2 else ("normalized", layer, "ln1")][0, token]
        if torch.dot(vector, x) == 0:
            return torch.tensor(0.0, device=vector.device)
        return torch.dot(vector, y) / torch.dot(vector, x) if not recip else torch.dot(vector, x) / torch.dot(vector, y)

    def get_attn_head_contribs(self, layer: int, seq_index: int, feature_vector: torch.Tensor):
        """Active z features of ``layer`` at ``seq_index`` as residual vectors ``[n_active, d_model]``
        (the reference ignores ``feature_vector`` here too, :1189-1233)."""
        seq_index = self.process_seq_index(seq_index)
        zf, zv, _ = self._active("attn", layer, seq_index)
        if zf.numel() == 0:
            return torch.zeros(0, self.model.W_O.shape[-1], device=zv.device)
        return self.z_saes[layer].device_dict().decode_rows(zf, zv) @ self._w_o_flat(layer)

    def get_transcoder_contribs(self, layer: int, seq_index: int, feature_vector: torch.Tensor, k=5):
        transcoder = self.mlp_transcoders[layer]
        is_post_ln = "ln2" in transcoder.cfg.hook_point
        if feature_vector.dim() == 2:
            feature_vector = feature_vector.reshape(-1)
        pulled, _ = self.get_transcoder_ixg(transcoder, layer, seq_index, feature_vector)
        top_contribs, top_indices = torch.topk(pulled, k=k)
        out = []
        for contrib, index in zip(top_contribs, top_indices):
            vector = transcoder.W_enc[:, index] * torch.dot(transcoder.W_dec[index], feature_vector)
            if is_post_ln:
                vector = vector * self.get_ln_constant(vector, layer, seq_index)  # ln1 constant, as the reference (:1269)
            out.append((vector, layer, seq_index, index.item(), contrib.item()))
        return out

    def get_top_contribs(self, feature_vector: torch.Tensor, layer: int, seq_index: int, k=5):
        """MLP-feature pull-backs of layers ``<= layer`` merged by score.  The reference's attention
        branch (:1285-1310) unpacks a 2-D ``unravel_index`` into three names and raises for any
        layer with an active z feature; that branch is therefore not reproduced."""
        contribs = []
        for cur in range(layer + 1):
            contribs += self.get_transcoder_contribs(cur, seq_index, feature_vector, k=k)
        scores = torch.tensor([c[4] for c in contribs])
        _, top = torch.topk(scores, k=min(k, len(scores)))
        return [contribs[i.item()] for i in top]


# ---------------------------------------------------------------------- many prompts at once
_COMPONENT_ORDER = [CircuitComponent.EMBED, CircuitComponent.POS_EMBED, CircuitComponent.BIAS_O,
                    CircuitComponent.Z_SAE_ERROR, CircuitComponent.Z_SAE_BIAS, CircuitComponent.Z_FEATURE,
                    CircuitComponent.TRANSCODER_ERROR, CircuitComponent.TRANSCODER_BIAS, CircuitComponent.MLP_FEATURE]


class BatchedCircuitLens:
    """CircuitLens over MANY prompts (the reference stops at one, circuit_lens.py:538-539).

    One ``run_with_cache`` over the padded token matrix, then ONE ``tsc_sparse_lens_pass`` per dictionary over all
    covered rows -- every real position of every prompt (``positions="all"``) or one position per prompt
    (``positions=-1``: the IOI evaluation position, task_evaluation.py:116) -- and one host sync for the pass.

    * ``lens(i)`` -> a ``CircuitLens`` for prompt ``i`` that shares this pass: same methods, same results as
      ``CircuitLens(prompts[i])`` (padding sits to the right of a causal model, so real positions are unaffected).
    * ``direction_lens(directions)`` -> for every covered row the contribution of every component row of
      ``ActiveFeatures`` to the row's direction, normalised as ``normalize_activations`` does, and its top-k --
      ``feature_act x W_dec row . direction`` fused into the pass (no ``[comp, 768]`` matrix is built)."""

    def __init__(self, prompts, positions: Union[str, int] = "all", max_active: int = 512, screened: bool = True):
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        model, z_saes, mlp_transcoders = get_model_encoders(self.device)
        self.model, self.z_saes, self.mlp_transcoders = model, z_saes, mlp_transcoders
        self.prompts = [prompts] if isinstance(prompts, str) else list(prompts)
        if not self.prompts:
            raise ValueError("BatchedCircuitLens needs at least one prompt")
        self.lengths = [int(model.to_tokens(p).shape[1]) for p in self.prompts]
        self.tokens = model.to_tokens(self.prompts)
        self.max_active = max_active
        self.screened = screened  # bf16-screened emit (exact results, TSC_F32_SCREEN); False: the plain fp32 emit
        with torch.no_grad():
            self.logits, self.cache = model.run_with_cache(self.tokens, return_type="logits")
        self.positions = positions
        rp, rs, self._row_start = [], [], []
        for i, n in enumerate(self.lengths):
            self._row_start.append(len(rp))
            cover = range(n) if positions == "all" else [int(positions) % n]
            rp += [i] * len(cover)
            rs += list(cover)
        dev = self.tokens.device
        self.row_prompt = torch.tensor(rp, device=dev)
        self.row_pos = torch.tensor(rs, device=dev)
        self._row_pos_host = rs
        self._pass = None
        self._views: Dict[int, CircuitLens] = {}
        self._handle_cache: dict = {}  # run_dictionary_pass: library handles + call tables of this batch's passes

    @property
    def n_prompts(self):
        return len(self.prompts)

    @property
    def n_rows(self):
        return len(self._row_pos_host)

    def _rows_of(self, key):
        """``cache[key]`` at the covered rows -> ``[R, ...]``."""
        return self.cache[key][self.row_prompt, self.row_pos]

    def _w_o_flat(self, layer: int) -> torch.Tensor:
        w = self.model.W_O[layer]
        return w.reshape(w.shape[0] * w.shape[1], w.shape[2])

    def _encoders_and_inputs(self):
        """Encoders in dictionary order ``[z_0, mlp_0, z_1, mlp_1, ...]`` and their inputs at the covered rows
        (gathered once: the pass, the error terms and every later ``direction_lens`` reuse them)."""
        if getattr(self, "_inputs", None) is None:
            encs, xs = [], []
            for layer in range(self.model.cfg.n_layers):
                z = self._rows_of(("z", layer))
                encs += [self.z_saes[layer], self.mlp_transcoders[layer]]
                xs += [z.reshape(z.shape[0], -1).contiguous(), self._rows_of(("normalized", layer, "ln2")).contiguous()]
            self._inputs = (encs, xs)
        return self._inputs

    def _row_bounds(self) -> Optional[torch.Tensor]:
        """Row bounds of the bf16 screen for every dictionary ``[D]``: they depend on the inputs only, so one
        batched computation serves every pass over this batch."""
        if self.n_rows > 65536 or not self.screened:  # tsc_hip.SCREEN_MAX_ROWS
            return None
        if getattr(self, "_bounds", None) is None:
            import tsc_hip

            encs, xs = self._encoders_and_inputs()
            with torch.no_grad():
                self._bounds = tsc_hip.screen_row_bound(torch.stack(xs), torch.stack([e.b_dec.detach() for e in encs]),
                                                        per_row=True)
        return self._bounds

    def _tables(self):
        """Per-layer constants of the direction lens, stacked over layers once per batch."""
        if getattr(self, "_tab", None) is None:
            n_layers = self.model.cfg.n_layers
            _, xs = self._encoders_and_inputs()
            self._tab = {
                "w_o_t": torch.stack([self._w_o_flat(l).t() for l in range(n_layers)]),            # [L, d_model, n*d]
                "b_o": torch.stack([self.model.b_O[l] for l in range(n_layers)]),                   # [L, d_model]
                "z_b_dec": torch.stack([self.z_saes[l].b_dec for l in range(n_layers)]),            # [L, n*d]
                "t_b_out": torch.stack([self.mlp_transcoders[l].b_dec_out for l in range(n_layers)]),
                "z": torch.stack(xs[0::2]),                                                         # [L, R, n*d]
                "mlp_out": torch.stack([self._rows_of(("mlp_out", l)) for l in range(n_layers)]),   # [L, R, d_model]
                "embed": self._rows_of("embed"), "pos_embed": self._rows_of("pos_embed"),
            }
        return self._tab

    # The direction lens scores residual-stream directions: with the z-SAE decoders folded through W_O once
    # (_ResidualSpaceZ) no pass pulls its directions back through W_O.  TSC_LENS_RESIDUAL=0: the z-space form
    # (W_O @ d per pass, the z-SAEs' own handles).
    RESIDUAL_DECODER = os.environ.get("TSC_LENS_RESIDUAL", "1") != "0"

    def _residual_world(self):
        """``(encoders, z W_O [L, R, d_model])``: the dictionaries of the pass with every z-SAE seen through W_O, and
        the attention outputs the z-SAEs reconstruct (without b_O), projected once per batch."""
        if getattr(self, "_resid", None) is None:
            _, xs = self._encoders_and_inputs()
            encs = []
            with torch.no_grad():
                for layer in range(self.model.cfg.n_layers):
                    encs += [_residual_z(self.z_saes[layer], self._w_o_flat(layer)), self.mlp_transcoders[layer]]
                z_resid = torch.stack([xs[2 * l] @ self._w_o_flat(l) for l in range(self.model.cfg.n_layers)])
            self._resid = (encs, z_resid)
        return self._resid

    def dictionary_pass(self, directions: Optional[torch.Tensor] = None):
        """Run (or return the cached) pass.  With ``directions`` (``[R, d_model]``, ``[n_prompts, d_model]`` with
        one row per prompt, or ``[d_model]``) the contribution scores are computed in the same launches; the
        attention dictionaries get ``directions @ W_O[layer]^T`` (the W_O projection of circuit_lens.py:618-627
        folded into the direction)."""
        if directions is None and self._pass is not None:
            return self._pass
        encs, xs = self._encoders_and_inputs()
        dirs = None
        if directions is not None:
            dirs, _ = self._dictionary_directions(self._row_directions(directions))
        with torch.no_grad():
            out = run_dictionary_pass(encs, xs, self.max_active, dirs=dirs, screened=self.screened,
                                      row_bounds=self._row_bounds(), handle_cache=self._handle_cache)
        if self._pass is None or directions is None:
            self._pass = out
        return out

    def _dictionary_directions(self, d: torch.Tensor):
        """Row directions as each dictionary sees them, one ``[R, d_out]`` tensor per dictionary (z-SAEs: pulled back
        through W_O; every transcoder sees ``d`` itself -- no copy), and the stacked z directions ``[L, R, n*d]``."""
        dz = torch.matmul(d[None], self._tables()["w_o_t"])                       # [L, R, n*d]
        dirs = []
        for layer in range(self.model.cfg.n_layers):
            dirs += [dz[layer], d]
        return dirs, dz

    def _row_directions(self, directions: torch.Tensor) -> torch.Tensor:
        d = directions.to(self.tokens.device, torch.float32)
        if d.dim() == 1:
            d = d.expand(self.n_rows, -1)
        elif d.shape[0] == self.n_prompts and self.n_rows != self.n_prompts:
            d = d[self.row_prompt]
        elif d.shape[0] != self.n_rows:
            raise ValueError(f"directions must have {self.n_rows} rows (one per covered position), "
                             f"{self.n_prompts} (one per prompt) or be a single vector; got {tuple(d.shape)}")
        return d.contiguous()

    def _pass_for_prompt(self, index: int):
        feat, val, count, recon, _, counts_host = self.dictionary_pass()
        a = self._row_start[index]
        b = self._row_start[index + 1] if index + 1 < self.n_prompts else self.n_rows
        rowmap = None if self.positions == "all" else {self._row_pos_host[a]: 0}
        return (feat[:, a:b], val[:, a:b], count[:, a:b], recon[:, a:b], counts_host[:, a:b].tolist(), rowmap)

    def lens(self, index: int) -> CircuitLens:
        if index not in self._views:
            self._views[index] = CircuitLens(self.prompts[index], _batch=self, _index=index)
        return self._views[index]

    def __len__(self):
        return self.n_prompts

    def __getitem__(self, index: int) -> CircuitLens:
        return self.lens(index)

    # ------------------------------------------------------------------ batched direction lens
    def direction_lens(self, directions: torch.Tensor, k: int = 10, dense: bool = True):
        """For every covered row: ``normalize_activations(vectors @ direction)`` of its ``ActiveFeatures`` without
        building the vectors, and its top-``k``.  Returns a dict of device tensors:

        ``activations [R, W]`` (padded column layout, 0 in padding), ``mask [R, W]``, ``top_values [R, k]`` and the
        identity of each winner ``top_component [R, k]`` (index into ``component_names``), ``top_layer [R, k]``,
        ``top_feature [R, k]`` (-1 on auxiliary rows); ``-inf`` / -1 where a row has fewer than k components.
        Column layout per row = the ActiveFeatures row order with every feature block padded to the capacity.
        ``dense=False`` skips the three ``[R, W]`` outputs (``activations``, ``mask``, ``raw``).
        The assembly, the normalisation and the ranking are ONE launch (``tsc_lens_assemble``), queued behind the
        dictionaries before the host reads the counts; ties rank by column (the reference's ``topk`` leaves them open)."""
        import tsc_hip

        _mark("lens: start")
        d = self._row_directions(directions)
        encs, xs = self._encoders_and_inputs()
        residual = self.RESIDUAL_DECODER and self.model.cfg.d_model == d.shape[1]
        with torch.no_grad():
            t = self._tables()
            n_layers, rows = self.model.cfg.n_layers, self.n_rows
            state = {}
            if residual:
                encs, z_resid = self._residual_world()
                cache = self.__dict__.setdefault("_handle_cache_resid", {})
                same = state["dirs"] = [d] * len(encs)  # every dictionary decodes into the residual stream: one direction
            else:
                cache = self._handle_cache

            def dirs_after_emit():
                # queued behind the emits, which do not need them: the pull-back through W_O and its host time are hidden
                state["dirs"], state["dz"] = self._dictionary_directions(d)
                _mark("lens: directions queued")
                return state["dirs"]

            def tail(feat, val, count, recon, scores):
                # (everything here is queued behind the dictionaries' launches: the host works while the GPU does)
                # head = embed, pos_embed; auxiliary rows of every layer (row order of ActiveFeatures: b_O, z error,
                # z bias | transcoder error, transcoder bias), each dotted with the row's direction: ONE launch
                head = torch.empty(rows, 2, dtype=torch.float32, device=d.device)
                aux = torch.empty(rows, 5 * n_layers, dtype=torch.float32, device=d.device)
                if residual:
                    self._aux_dots_residual(d, z_resid, encs, recon, head, aux)
                else:
                    self._aux_dots(d, state["dz"], recon, head, aux)
                res = tsc_hip.lens_assemble(head, aux, [3, 2] * n_layers, scores, count, k=k, dense=dense,
                                            column_tags=self._column_tags(d.device), feat=feat)
                res["top_info_long"] = res["top_info"].long()  # (queued here: nothing is launched after the host sync)
                return res

            feat, val, count, recon, scores, counts_host, res = run_dictionary_pass(
                encs, xs, self.max_active, dirs=same if residual else dirs_after_emit, screened=self.screened,
                row_bounds=self._row_bounds(), then=tail, handle_cache=cache)
            if self._pass is None and not residual:  # (the residual form's z reconstructions are not the z-space ones)
                self._pass = (feat, val, count, recon, scores, counts_host)
            _mark("lens: pass returned")
            info = res["top_info_long"]  # [R, k, 4]: column, component, layer, feature (-1 where a row ran out)
            out = {
                "top_values": res["top_values"],
                "top_component": info[:, :, 1], "top_layer": info[:, :, 2], "top_feature": info[:, :, 3],
                "component_names": list(_COMPONENT_ORDER),
                "row_prompt": self.row_prompt, "row_pos": self.row_pos,
            }
            if dense:
                out.update(activations=res["activations"], mask=res["mask"], raw=res["raw"])
            return out

    def _aux_dots(self, d, dz, recon, head, aux) -> None:
        """``head [R, 2]`` / ``aux [R, 5 L]`` = the non-feature rows of every position's ``ActiveFeatures`` dotted with
        the row's direction (``tsc_rowdot_many``, one launch).  The term table is built once per batch; a pass only
        patches the addresses of what it allocated (directions, reconstructions, outputs)."""
        import tsc_hip

        t = self._tables()
        n_layers, rows, dm = self.model.cfg.n_layers, self.n_rows, d.shape[1]
        tab = getattr(self, "_rd", None)
        if tab is None:
            tab = tsc_hip.RowDotTable(2 + 5 * n_layers, rows, dm)
            tab.set(0, t["embed"], None, d, head[:, 0])
            tab.set(1, t["pos_embed"], None, d, head[:, 1])
            for l in range(n_layers):
                j = 2 + 5 * l
                tab.set(j, t["b_o"][l], None, d, aux[:, 5 * l])
                tab.set(j + 1, t["z"][l], recon[2 * l], dz[l], aux[:, 5 * l + 1])
                tab.set(j + 2, t["z_b_dec"][l], None, dz[l], aux[:, 5 * l + 2])
                tab.set(j + 3, t["mlp_out"][l], recon[2 * l + 1], d, aux[:, 5 * l + 3])
                tab.set(j + 4, t["t_b_out"][l], None, d, aux[:, 5 * l + 4])
            self._rd = tab
        else:  # same shapes and strides as when the table was made: only base addresses move
            if not (d.is_contiguous() and dz.is_contiguous() and recon.is_contiguous()) or dz.shape[2] != dm:
                raise tsc_hip.TscError("direction lens: directions / reconstructions must be contiguous")
            dp, zp, rp, hp, ap = d.data_ptr(), dz.data_ptr(), recon.data_ptr(), head.data_ptr(), aux.data_ptr()
            zl, rl = dz.stride(0) * 4, recon.stride(0) * 4
            b, c, out = tab.b, tab.c, tab.out
            b[0] = b[1] = dp
            out[0], out[1] = hp, hp + 4
            for l in range(n_layers):
                j = 2 + 5 * l
                b[j] = b[j + 3] = b[j + 4] = dp
                b[j + 1] = b[j + 2] = zp + l * zl
                c[j + 1], c[j + 3] = rp + 2 * l * rl, rp + (2 * l + 1) * rl
                o = ap + 20 * l
                out[j], out[j + 1], out[j + 2], out[j + 3], out[j + 4] = o, o + 4, o + 8, o + 12, o + 16
        tab.run(d.device)

    def _aux_dots_residual(self, d, z_resid, encs, recon, head, aux) -> None:
        """``_aux_dots`` with the z-SAEs in residual space: z error = ``(z W_O - reconstruction) . d`` (the reconstruction
        of a ``_ResidualSpaceZ`` handle already lies in the residual stream and carries ``b_dec W_O``), z bias =
        ``(b_dec W_O) . d``; every term meets the row's direction itself."""
        import tsc_hip

        t = self._tables()
        n_layers, rows, dm = self.model.cfg.n_layers, self.n_rows, d.shape[1]
        tab = getattr(self, "_rd_resid", None)
        if tab is None:
            tab = tsc_hip.RowDotTable(2 + 5 * n_layers, rows, dm)
            tab.set(0, t["embed"], None, d, head[:, 0])
            tab.set(1, t["pos_embed"], None, d, head[:, 1])
            for l in range(n_layers):
                j = 2 + 5 * l
                tab.set(j, t["b_o"][l], None, d, aux[:, 5 * l])
                tab.set(j + 1, z_resid[l], recon[2 * l], d, aux[:, 5 * l + 1])
                tab.set(j + 2, encs[2 * l].b_out, None, d, aux[:, 5 * l + 2])
                tab.set(j + 3, t["mlp_out"][l], recon[2 * l + 1], d, aux[:, 5 * l + 3])
                tab.set(j + 4, t["t_b_out"][l], None, d, aux[:, 5 * l + 4])
            self._rd_resid = tab
        else:  # same shapes and strides as when the table was made: only base addresses move
            if not (d.is_contiguous() and recon.is_contiguous()):
                raise tsc_hip.TscError("direction lens: directions / reconstructions must be contiguous")
            dp, rp, hp, ap = d.data_ptr(), recon.data_ptr(), head.data_ptr(), aux.data_ptr()
            rl = recon.stride(0) * 4
            a, b, c, out = tab.a, tab.b, tab.c, tab.out
            b[0] = b[1] = dp
            out[0], out[1] = hp, hp + 4
            for l in range(n_layers):
                j = 2 + 5 * l
                b[j] = b[j + 1] = b[j + 2] = b[j + 3] = b[j + 4] = dp
                a[j + 2] = encs[2 * l].b_out.data_ptr()  # (refolded when a parameter of the z-SAE changed)
                c[j + 1], c[j + 3] = rp + 2 * l * rl, rp + (2 * l + 1) * rl
                o = ap + 20 * l
                out[j], out[j + 1], out[j + 2], out[j + 3], out[j + 4] = o, o + 4, o + 8, o + 12, o + 16
        tab.run(d.device)

    def _column_tags(self, dev):
        """``component << 16 | layer`` for every column class of the padded layout, in ``tsc_lens_assemble``'s order:
        embed, pos_embed; per layer b_O, z error, z bias, transcoder error, transcoder bias; then one per dictionary."""
        cache = self.__dict__.setdefault("_col_tags", {})
        if str(dev) not in cache:
            n_layers = self.model.cfg.n_layers
            tags = [0 << 16, 1 << 16]
            for layer in range(n_layers):
                tags += [(c << 16) | layer for c in (2, 3, 4, 6, 7)]
            for layer in range(n_layers):
                tags += [(5 << 16) | layer, (8 << 16) | layer]
            cache[str(dev)] = torch.tensor(tags, dtype=torch.int32, device=dev)
        return cache[str(dev)]

    def direction_lens_runs(self, directions: torch.Tensor, k: int = 10) -> List[List[ComponentLensWithValue]]:
        """``direction_lens`` as the per-prompt lists the reference's lens functions return (one list per covered
        row): ``[(ComponentLens, value)]``, ending at the first embed / pos_embed row like ``get_top_k_features``."""
        res = self.direction_lens(directions, k=k)
        tv, tc, tl, tf = (res[n].tolist() for n in ("top_values", "top_component", "top_layer", "top_feature"))
        out = []
        for r in range(self.n_rows):
            lens = self.lens(int(self.row_prompt[r]))
            runs = []
            for v, c, l, f in zip(tv[r], tc[r], tl[r], tf[r]):
                if c < 0:
                    break
                runs.append((ComponentLens(circuit_lens=lens, component=_COMPONENT_ORDER[c],
                                           run_data={"layer": l, "seq_index": self._row_pos_host[r], "feature": f}), v))
                if c < 2:
                    break
            out.append(runs)
        return out
