"""Shared machinery of the two dictionary classes: the link between an ``nn.Module``'s parameters and
its registration with ``libtsc_hip`` (``tsc_hip.DeviceDict``).  The handle borrows the parameters'
device memory, so it is rebuilt whenever a parameter is re-allocated (``.to()``, ``load_state_dict``
into new storage)."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

import tsc_hip

try:  # the reference uses TransformerLens hook points; identical-behaving stand-ins when it is absent
    from transformer_lens.hook_points import HookPoint  # type: ignore
except Exception:  # pragma: no cover - transformer_lens is not installed offline

    class HookPoint(nn.Module):
        """Identity unless a forward hook is attached (what a TransformerLens HookPoint is)."""

        def forward(self, x):
            return x


class HipDictionaryMixin:
    """Expects ``W_enc, b_enc, b_dec, W_dec`` parameters and ``_out_bias()`` on the host class."""

    _hip: Optional[tsc_hip.DeviceDict] = None
    _hip_bf16: Optional[tsc_hip.DeviceDict] = None
    _hip_screen: Optional[tsc_hip.DeviceDict] = None

    def _out_bias(self) -> torch.Tensor:  # pragma: no cover - overridden
        raise NotImplementedError

    _HANDLE_ATTRS = ("_hip", "_hip_bf16", "_hip_screen", "_projectors", "_resid_adapter")

    def __getstate__(self):
        """Library handles (ctypes pointers into this process' GPU context) never travel with a copy or a pickle:
        ``copy.deepcopy`` / ``torch.save(module)`` get the parameters only and re-register on first use."""
        state = dict(self.__dict__)
        for name in self._HANDLE_ATTRS:
            state.pop(name, None)
        return state

    def __deepcopy__(self, memo):
        import copy

        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__getstate__().items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def _refuse_autograd(self, x: torch.Tensor) -> None:
        """The reference's encode is differentiable; the HIP kernels have no backward.  A caller that would
        back-propagate through the dictionary gets an error instead of a silently missing gradient."""
        if torch.is_grad_enabled() and x.requires_grad:
            raise tsc_hip.TscError(
                "the MI355X encode has no backward: the input requires grad and autograd is enabled -- wrap the call "
                "in torch.no_grad() (every analysis caller of the reference does, e.g. max_act_analysis.py:216) or use "
                "the reference's PyTorch encode for attribution / training")

    def _require_gpu_fp32(self) -> None:
        p = self.W_enc
        if not p.is_cuda:
            raise tsc_hip.TscError(
                f"{type(self).__name__} lives on {p.device}: the MI355X path runs on a GPU only and has no CPU "
                "fallback (the reference's CPU branch is the oracle's job, not the product's)")
        if p.dtype != torch.float32:
            raise tsc_hip.TscError(f"{type(self).__name__} weights must be float32 (got {p.dtype})")

    def device_dict(self, bf16: bool = False, screened: bool = False) -> tsc_hip.DeviceDict:
        """The library handle for the current parameter storage: fp32, the bf16-MFMA variant, or (``screened``)
        the fp32 handle that also carries the operands of the lens pass' bf16 screen -- fp32 results from every
        entry point; dictionaries whose ``d_in`` is not a multiple of 128 get the plain fp32 handle."""
        try:  # fast path (this runs once per dictionary and call): same parameter storage as when the handle was made
            prm = self._parameters
            W_enc, b_enc, b_dec, W_dec, ob = prm["W_enc"], prm["b_enc"], prm["b_dec"], prm["W_dec"], self._out_bias()
            if screened and (bf16 or W_enc.shape[0] % 128 != 0 or W_enc.shape[0] > 768 or W_dec.shape[1] > 768):
                screened = False  # the screened finish kernel holds a row of <= 768 inputs in registers
            attr = "_hip_bf16" if bf16 else "_hip_screen" if screened else "_hip"
            inst = self.__dict__
            if attr == "_hip" and inst.get("_hip") is None and inst.get("_hip_screen") is not None:
                attr = "_hip_screen"  # a screening handle does everything a plain fp32 handle does: do not keep two
            cur = inst.get(attr)
            if cur is not None and cur.__dict__.get("_param_key") == (W_enc.data_ptr(), b_enc.data_ptr(), b_dec.data_ptr(),
                                                                      W_dec.data_ptr(), ob.data_ptr(), W_enc.shape, W_dec.shape):
                cur.refresh_if_modified()
                return cur
        except (KeyError, AttributeError):
            pass
        self._require_gpu_fp32()
        if screened and (bf16 or self.W_enc.shape[0] % 128 != 0 or self.W_enc.shape[0] > 768 or self.W_dec.shape[1] > 768):
            screened = False
        originals = (self.W_enc, self.b_enc, self.b_dec, self.W_dec, self._out_bias())
        tensors = tuple(t.detach() for t in originals)
        tensors = tuple(t if t.is_contiguous() else t.contiguous() for t in tensors)
        attr = "_hip_bf16" if bf16 else "_hip_screen" if screened else "_hip"
        if attr == "_hip" and self._hip is None and self._hip_screen is not None:
            attr = "_hip_screen"
        cur = getattr(self, attr)
        if cur is None or not cur.same_storage(*tensors):
            if cur is not None:
                cur.close()
            cur = tsc_hip.DeviceDict(*tensors, compute_dtype=tsc_hip.BF16 if bf16 else
                                     tsc_hip.F32_SCREEN if screened else tsc_hip.F32)
            if all(a.data_ptr() == b.data_ptr() for a, b in zip(originals, tensors)):  # no contiguous copies were needed
                # (storage addresses AND shapes: a parameter re-allocated at a freed address with another shape is new)
                cur._param_key = tuple(t.data_ptr() for t in originals) + (self.W_enc.shape, self.W_dec.shape)
            object.__setattr__(self, attr, cur)
        else:
            cur.refresh_if_modified()
        return cur

    # ---- capabilities beyond the reference API (the fused hot path) ----
    def new_topk_state(self, k: int = 20, n_slots: Optional[int] = None) -> tsc_hip.TopkState:
        return tsc_hip.TopkState(self.W_enc.shape[1], k, self.W_enc.device, n_slots=n_slots)

    def encode_topk(self, x: torch.Tensor, token_id_base: int, state: tsc_hip.TopkState, bf16: bool = False,
                    screened: bool = False) -> None:
        """Fused encode + per-feature running top-k; the activation matrix is never materialised.  ``screened``: go
        through ``tsc_encode_topk_screened`` (bf16 MFMA screen + exact fp32 recompute: the same lists bit for bit,
        several times faster on warm lists) when the dictionary's shape allows it."""
        x = x.to(device=self.W_enc.device, dtype=torch.float32).contiguous()
        if screened and not bf16 and state.floor is not None:
            handle = self.device_dict(screened=True)
            if handle.compute_dtype == tsc_hip.F32_SCREEN:
                handle.encode_topk_screened(x, token_id_base, state)
                return
        self.device_dict(bf16).encode_topk(x, token_id_base, state)

    def encode_sparse(self, x: torch.Tensor, max_active: int = 512):
        """Per row the active features, value-descending: ``(feat, val, count)``."""
        x = x.to(device=self.W_enc.device, dtype=torch.float32).contiguous()
        return self.device_dict().encode_sparse(x, max_active)

    def encode_features(self, x: torch.Tensor, features, relu: bool = True) -> torch.Tensor:
        """Activations of a few selected features only -> ``[..., len(features)]``."""
        self._require_gpu_fp32()
        x = x.to(device=self.W_enc.device, dtype=torch.float32).contiguous()
        feats = torch.as_tensor(features, device=self.W_enc.device, dtype=torch.long).reshape(-1)
        out = self.feature_projector(feats)(x, relu=relu)
        return out.reshape(*x.shape[:-1], feats.numel())

    def feature_projector(self, feats: torch.Tensor) -> "tsc_hip.FeatureProjector":
        """Projector onto the selected encoder columns, cached per feature list for as long as the
        parameters keep their storage and values (callers loop over batches with one feature list)."""
        params = (self.W_enc, self.b_enc, self.b_dec)
        key = (tuple(feats.tolist()), tuple(p.data_ptr() for p in params), tuple(p._version for p in params))
        cache = self.__dict__.setdefault("_projectors", {})
        proj = cache.get(key)
        if proj is None:
            while len(cache) >= 4:  # a handful of feature lists is all a sweep uses at a time
                cache.pop(next(iter(cache))).close()
            Pt = self.W_enc.detach().index_select(1, feats).t().contiguous()
            bias = self.b_enc.detach().index_select(0, feats).contiguous()
            proj = cache[key] = tsc_hip.FeatureProjector(Pt, bias, self.b_dec.detach().contiguous())
        return proj
