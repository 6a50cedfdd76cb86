"""ctypes binding of ``libtsc_hip.so`` (C-ABI in ``include/tsc_hip.h``) for torch-ROCm tensors.

torch is plumbing here: it owns device memory and the HIP stream.  Every function passes raw
``data_ptr()`` values and ``torch.cuda.current_stream().cuda_stream`` across the C boundary.
There is NO fallback: if the library is not built, or a tensor is not on a GPU, the call raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import torch

# TSC_HIP_LIBRARY selects another build of the same library (kernel A/B experiments); never a fallback
_LIB_PATH = os.environ.get("TSC_HIP_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtsc_hip.so")
_lib: Optional[ctypes.CDLL] = None

F32, BF16, F32_SCREEN = 0, 1, 2  # include/tsc_hip.h: TSC_F32, TSC_BF16, TSC_F32_SCREEN
MAX_ACTIVE_LIMIT = 8192
MAX_DIRS = 8
LENS_PASS_MAX_ACTIVE = 2048  # tsc_sparse_lens_pass
TOPK_MERGE_MAX = 2048              # include/tsc_hip.h: TSC_TOPK_MERGE_MAX
STATIC_FLOOR_MAX_D_SAE = 131072    # include/tsc_hip.h: TSC_STATIC_FLOOR_MAX_D_SAE
SCREEN_MAX_ROWS = 65536            # include/tsc_hip.h: TSC_SCREEN_MAX_ROWS

# name -> (restype, argtypes); mirrors include/tsc_hip.h one to one
_c = ctypes
_SIGNATURES = {
    "tsc_version": (_c.c_int, []),
    "tsc_last_error": (_c.c_char_p, []),
    "tsc_dict_create": (_c.c_int, [_c.POINTER(_c.c_void_p), _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                   _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "tsc_dict_refresh": (_c.c_int, [_c.c_void_p, _c.c_void_p]),
    "tsc_dict_destroy": (_c.c_int, [_c.c_void_p]),
    "tsc_encode_dense": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p]),
    "tsc_encode_dense_pre": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p]),
    "tsc_topk_recommended_slots": (_c.c_int, [_c.c_int]),
    "tsc_set_tuning": (_c.c_int, [_c.c_char_p, _c.c_int]),
    "tsc_topk_state_reset": (_c.c_int, [_c.c_void_p, _c.c_void_p]),
    "tsc_encode_topk": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_void_p]),
    "tsc_encode_topk_static": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_void_p]),
    "tsc_topk_screen_workspace_bytes": (_c.c_size_t, [_c.c_int, _c.c_int]),
    "tsc_encode_topk_screened": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_void_p,
                                            _c.c_size_t, _c.c_void_p, _c.c_void_p]),
    "tsc_topk_merge": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p,
                                  _c.c_void_p, _c.c_void_p]),
    "tsc_topk_finalize": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "tsc_encode_sparse": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int, _c.c_void_p, _c.c_void_p,
                                     _c.c_void_p, _c.c_void_p]),
    "tsc_dense_to_sparse": (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                       _c.c_void_p]),
    "tsc_decode_rows": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p]),
    "tsc_decode_sparse": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int,
                                     _c.c_void_p, _c.c_void_p]),
    "tsc_contrib_scores": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int,
                                      _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p]),
    "tsc_sparse_lens_pass": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int, _c.c_void_p, _c.c_void_p,
                                        _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p,
                                        _c.c_void_p, _c.c_void_p]),
    "tsc_sparse_lens_emit_many": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int, _c.c_void_p,
                                             _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int]),
    "tsc_sparse_lens_finish_many": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int, _c.c_void_p,
                                               _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int64, _c.c_void_p,
                                               _c.c_void_p, _c.c_void_p]),
    "tsc_lens_assemble": (_c.c_int, [_c.c_int64, _c.c_int, _c.c_int, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p,
                                     _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p,
                                     _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "tsc_contrib_scores_many": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int, _c.c_void_p, _c.c_void_p,
                                           _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                           _c.c_int, _c.c_void_p]),
    "tsc_rowdot_many": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                   _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int, _c.c_void_p]),
    "tsc_project_features": (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                        _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class TscError(RuntimeError):
    """Non-zero status from the C-ABI (the reference raises Python exceptions, so do we)."""


class _TopkStateStruct(ctypes.Structure):
    _fields_ = [("vals", _c.c_void_p), ("idx", _c.c_void_p), ("d_sae", _c.c_int32), ("k", _c.c_int32),
                ("n_slots", _c.c_int32), ("flags", _c.c_int32), ("floor", _c.c_void_p)]


def library_path() -> str:
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    """Load the HIP library or fail loudly -- there is no CPU path behind this package."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise TscError(
                f"{_LIB_PATH} is missing: build it with `make -C transcoder-sae-circuits_amd/csrc` "
                "(or __graft_entry__.build()); the MI355X path has no fallback")
        handle = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def set_tuning(key: str, value: int) -> None:
    """Force a kernel variant (tests / measurements): see ``tsc_set_tuning`` in include/tsc_hip.h."""
    _check(lib().tsc_set_tuning(key.encode(), int(value)))


def _check(rc: int) -> None:
    if rc != 0:
        raise TscError(f"libtsc_hip status {rc}: {lib().tsc_last_error().decode()}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, dtype: torch.dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise TscError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise TscError(f"{name} must be {dtype} (got {t.dtype})")
    if not t.is_contiguous():
        raise TscError(f"{name} must be contiguous")
    return t


class DeviceDict:
    """One transcoder / z-SAE registered with the library (``tsc_dict_create``).  Keeps the
    weight tensors alive for as long as the handle borrows their memory."""

    def __init__(self, W_enc, b_enc, b_dec, W_dec=None, out_bias=None, compute_dtype: int = F32):
        self._handle = ctypes.c_void_p()
        self._keep = [_dev(W_enc, torch.float32, "W_enc"), _dev(b_enc, torch.float32, "b_enc"),
                      _dev(b_dec, torch.float32, "b_dec")]
        self.d_in, self.d_sae = W_enc.shape
        self.d_out = self.d_in
        ptr_dec = ptr_bias = None
        if W_dec is not None:
            self._keep.append(_dev(W_dec, torch.float32, "W_dec"))
            self.d_out = W_dec.shape[1]
            ptr_dec = W_dec.data_ptr()
        if out_bias is not None:
            self._keep.append(_dev(out_bias, torch.float32, "out_bias"))
            ptr_bias = out_bias.data_ptr()
        self.device = W_enc.device
        self.compute_dtype = compute_dtype
        with torch.cuda.device(self.device):
            _check(lib().tsc_dict_create(ctypes.byref(self._handle), self.d_in, self.d_sae, self.d_out,
                                         compute_dtype, W_enc.data_ptr(), b_enc.data_ptr(), b_dec.data_ptr(),
                                         ptr_dec, ptr_bias, _stream()))

        self._versions = self._param_versions()

    def _param_versions(self):
        return tuple(t._version for t in self._keep[:3])

    def same_storage(self, *tensors) -> bool:
        return all(a.data_ptr() == b.data_ptr() for a, b in zip(self._keep, tensors))

    def refresh(self) -> None:
        """Recompute what the handle derived from the weight values (``tsc_dict_refresh``)."""
        with torch.cuda.device(self.device):
            _check(lib().tsc_dict_refresh(self._handle, _stream()))
        self._versions = self._param_versions()

    def refresh_if_modified(self) -> None:
        """In-place updates of W_enc / b_enc / b_dec bump torch's version counters."""
        if self._param_versions() != self._versions:
            self.refresh()

    def close(self) -> None:
        if self._handle:
            lib().tsc_dict_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ dense
    def encode_dense(self, x: torch.Tensor) -> torch.Tensor:
        """``relu((x - b_dec) @ W_enc + b_enc)`` for ``x [..., d_in]`` -> ``[..., d_sae]``."""
        x2 = _dev(x, torch.float32, "x").reshape(-1, self.d_in)
        out = torch.empty(x2.shape[0], self.d_sae, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _check(lib().tsc_encode_dense(self._handle, x2.data_ptr(), x2.shape[0], out.data_ptr(), _stream()))
        return out.reshape(*x.shape[:-1], self.d_sae)

    # ------------------------------------------------------------------ top-k
    def encode_topk(self, x: torch.Tensor, token_id_base: int, state: "TopkState") -> None:
        x2 = _dev(x, torch.float32, "x").reshape(-1, self.d_in)
        state._schedule_floor()
        with torch.cuda.device(x.device):
            _check(lib().tsc_encode_topk(self._handle, x2.data_ptr(), x2.shape[0], int(token_id_base),
                                         ctypes.byref(state._struct), _stream()))

    def encode_topk_static(self, x: torch.Tensor, token_id_base: int, state: "TopkState") -> None:
        """``tsc_encode_topk_static``: the same lists as :meth:`encode_topk`, with the cross-slot floor in the
        library's static device scratch -- the call allocates nothing (for callers under HIP graph capture or
        in loops that may not allocate).  ``d_sae <= STATIC_FLOOR_MAX_D_SAE``; calls that overlap on one GPU
        race on the shared scratch."""
        x2 = _dev(x, torch.float32, "x").reshape(-1, self.d_in)
        with torch.cuda.device(x.device):
            _check(lib().tsc_encode_topk_static(self._handle, x2.data_ptr(), x2.shape[0], int(token_id_base),
                                                ctypes.byref(state._struct), _stream()))

    def encode_topk_screened(self, x: torch.Tensor, token_id_base: int, state: "TopkState",
                             row_bound: Optional[torch.Tensor] = None) -> None:
        """``tsc_encode_topk_screened`` (``F32_SCREEN`` handles): the lists of :meth:`encode_topk`, bit for bit, with
        the contraction on the bf16 MFMA as a screen and an exact fp32 recompute of the few pairs that can enter a
        list.  Interchangeable with :meth:`encode_topk` call by call; pays once the lists are warm."""
        x2 = _dev(x, torch.float32, "x").reshape(-1, self.d_in)
        ws = state.screen_workspace()
        state._struct.flags = 0  # the screen compares against the floor: always fresh
        state._calls += 1
        with torch.cuda.device(x.device):
            _check(lib().tsc_encode_topk_screened(
                self._handle, x2.data_ptr(), x2.shape[0], int(token_id_base), ctypes.byref(state._struct), ws.data_ptr(),
                ws.numel(), _dev(row_bound, torch.float32, "row_bound").data_ptr() if row_bound is not None else None,
                _stream()))

    def encode_pre(self, x: torch.Tensor) -> torch.Tensor:
        """``(x - b_dec) @ W_enc + b_enc`` (no ReLU: the reference's ``hook_hidden_pre``)."""
        x2 = _dev(x, torch.float32, "x").reshape(-1, self.d_in)
        out = torch.empty(x2.shape[0], self.d_sae, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _check(lib().tsc_encode_dense_pre(self._handle, x2.data_ptr(), x2.shape[0], out.data_ptr(), _stream()))
        return out.reshape(*x.shape[:-1], self.d_sae)

    # ------------------------------------------------------------------ sparse
    def encode_sparse(self, x: torch.Tensor, max_active: int = 512, check_capacity: bool = True) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Per row the active features, value-descending: ``(feat int32 [M,cap], val [M,cap],
        count int32 [M])``.  Grows the capacity and retries if a row overflows it -- which costs one
        host sync per call; ``check_capacity=False`` skips that (the caller then checks
        ``count.max() <= feat.shape[1]`` itself, e.g. once for a whole pass over many dictionaries)."""
        x2 = _dev(x, torch.float32, "x").reshape(-1, self.d_in)
        m = x2.shape[0]
        cap = max(1, min(int(max_active), MAX_ACTIVE_LIMIT, self.d_sae))
        while True:
            feat = torch.empty(m, cap, dtype=torch.int32, device=x.device)
            val = torch.empty(m, cap, dtype=torch.float32, device=x.device)
            count = torch.empty(m, dtype=torch.int32, device=x.device)
            with torch.cuda.device(x.device):
                _check(lib().tsc_encode_sparse(self._handle, x2.data_ptr(), m, cap, feat.data_ptr(), val.data_ptr(),
                                               count.data_ptr(), _stream()))
            if not check_capacity:
                return feat, val, count
            worst = int(count.max()) if m else 0
            if worst <= cap:
                return feat, val, count
            if cap >= min(MAX_ACTIVE_LIMIT, self.d_sae):
                raise TscError(f"a row has {worst} active features, above the {MAX_ACTIVE_LIMIT} the sparse path holds")
            cap = min(MAX_ACTIVE_LIMIT, self.d_sae, max(2 * cap, worst))

    def sparse_lens_pass(self, x: torch.Tensor, max_active: int = 512, dirs: Optional[torch.Tensor] = None,
                         want_recon: bool = True, out: Optional[dict] = None, row_bound: Optional[torch.Tensor] = None,
                         screen: bool = True, emit_only: bool = False):
        """``tsc_sparse_lens_pass``: sparse emit + (sort, reconstruction, per-row direction scores) in two launches.

        ``dirs``: ``[M, d_out]`` (one direction per row), ``[d_out]`` (shared) or None.  Returns ``(feat, val, count,
        recon or None, scores [M, cap] or None)``.  No capacity check and no host sync: the caller compares
        ``count.max()`` with ``feat.shape[1]`` once for as many dictionaries as it likes (``out`` may carry
        preallocated ``feat/val/count`` (and optionally ``recon/scores``) views -- ``count`` then has to be zero
        already -- so that one memset and one reduction serve a whole pass).  ``row_bound``: two-element device
        tensor for a ``F32_SCREEN`` handle (see :func:`screen_row_bound`; ``2 + 2 M`` elements: with every row's own pair);
        None lets the library compute it;
        ``screen=False`` asks such a handle for the plain fp32 emit.  ``emit_only``: stop after the emit
        (``TSC_LENS_EMIT_ONLY``); :func:`sparse_lens_finish_many` then finishes several dictionaries in one launch."""
        x2 = _dev(x, torch.float32, "x").reshape(-1, self.d_in)
        m = x2.shape[0]
        cap = max(1, min(int(max_active), LENS_PASS_MAX_ACTIVE, self.d_sae))
        dev = x.device
        if out is not None:
            feat, val, count = out["feat"], out["val"], out["count"]
            cap, zeroed = feat.shape[1], 1
        else:
            feat = torch.empty(m, cap, dtype=torch.int32, device=dev)
            val = torch.empty(m, cap, dtype=torch.float32, device=dev)
            count = torch.empty(m, dtype=torch.int32, device=dev)
            zeroed = 0
        recon = out.get("recon") if out is not None else None
        if recon is None and want_recon:
            recon = torch.empty(m, self.d_out, dtype=torch.float32, device=dev)
        scores, stride, dptr = None, 0, None
        if dirs is not None:
            dirs = _dev(dirs, torch.float32, "dirs")
            if dirs.dim() == 2 and dirs.shape[0] == m and dirs.shape[1] == self.d_out:
                stride = self.d_out if m > 1 or dirs.shape[0] == m else 0
            elif dirs.numel() == self.d_out:
                stride = 0
            else:
                raise TscError(f"dirs must be [{m}, {self.d_out}] or [{self.d_out}] (got {tuple(dirs.shape)})")
            scores = out.get("scores") if out is not None else None
            if scores is None:
                scores = torch.empty(m, cap, dtype=torch.float32, device=dev)
            dptr = dirs.data_ptr()
        with torch.cuda.device(dev):
            _check(lib().tsc_sparse_lens_pass(self._handle, x2.data_ptr(), m, cap, feat.data_ptr(), val.data_ptr(),
                                              count.data_ptr(),
                                              zeroed | (0 if screen else 2) | (4 if emit_only else 0)
                                              | (16 if row_bound is not None and row_bound.numel() == 2 + 2 * m else 0), dptr, stride,
                                              recon.data_ptr() if recon is not None else None,
                                              scores.data_ptr() if scores is not None else None,
                                              _dev(row_bound, torch.float32, "row_bound").data_ptr() if row_bound is not None else None,
                                              _stream()))
        return feat, val, count, recon, scores

    DECODE_SPARSE_MIN_ROWS = 256   # below this the dense matmul's launch is as cheap as the two sparse ones
    DECODE_SPARSE_CAPACITY = 1024

    def decode_dense_acts(self, acts: torch.Tensor) -> Optional[torch.Tensor]:
        """``acts @ W_dec + out_bias`` for dense activations ``[..., d_sae]`` that are mostly zeros, as
        ``tsc_dense_to_sparse`` + ``tsc_decode_sparse`` (one read of ``acts``, a gather of the active ``W_dec`` rows)
        instead of the dense product.  Returns None when that is not worth it or not possible (few rows, a row with more
        positives than the lists hold, values that are not plain activations): the caller multiplies densely then.
        One host synchronisation (the capacity check)."""
        a2 = acts.reshape(-1, self.d_sae)
        m = a2.shape[0]
        if m < self.DECODE_SPARSE_MIN_ROWS or not a2.is_cuda or a2.dtype != torch.float32 or not a2.is_contiguous() \
                or len(self._keep) < 5:
            return None
        cap = min(self.DECODE_SPARSE_CAPACITY, self.d_sae)
        feat = torch.empty(m, cap, dtype=torch.int32, device=a2.device)
        val = torch.empty(m, cap, dtype=torch.float32, device=a2.device)
        count = torch.empty(m, dtype=torch.int32, device=a2.device)
        with torch.cuda.device(a2.device):
            _check(lib().tsc_dense_to_sparse(a2.data_ptr(), m, self.d_sae, cap, feat.data_ptr(), val.data_ptr(),
                                             count.data_ptr(), _stream()))
        recon = self.decode_sparse(feat, val, count)  # queued before the check: the GPU does not wait for the host
        # (a row with more positives than the lists hold, or with negative / NaN entries -- a hook may have edited the
        # activations: reported as a negative count --: leave those to the matmul)
        if bool(((count > cap) | (count < 0)).any()):
            return None
        return recon.reshape(*acts.shape[:-1], self.d_out)

    def decode_rows(self, feat: torch.Tensor, val: torch.Tensor) -> torch.Tensor:
        """``val[i] * W_dec[feat[i]]`` -> ``[n, d_out]``."""
        feat = _dev(feat, torch.int32, "feat").reshape(-1)
        val = _dev(val, torch.float32, "val").reshape(-1)
        rows = torch.empty(feat.numel(), self.d_out, dtype=torch.float32, device=feat.device)
        with torch.cuda.device(feat.device):
            _check(lib().tsc_decode_rows(self._handle, feat.data_ptr(), val.data_ptr(), feat.numel(), rows.data_ptr(),
                                         _stream()))
        return rows

    def decode_sparse(self, feat, val, count) -> torch.Tensor:
        """``out_bias + sum_i val[m,i] * W_dec[feat[m,i]]`` -> ``[M, d_out]``."""
        _dev(feat, torch.int32, "feat"); _dev(val, torch.float32, "val"); _dev(count, torch.int32, "count")
        m, cap = feat.shape
        recon = torch.empty(m, self.d_out, dtype=torch.float32, device=feat.device)
        with torch.cuda.device(feat.device):
            _check(lib().tsc_decode_sparse(self._handle, feat.data_ptr(), val.data_ptr(), count.data_ptr(), m, cap,
                                           recon.data_ptr(), _stream()))
        return recon

    def contrib_scores(self, feat, val, count, dirs: torch.Tensor) -> torch.Tensor:
        """``val[m,i] * (W_dec[feat[m,i]] . dirs[d])`` -> ``[M, cap, n_dir]``."""
        _dev(feat, torch.int32, "feat"); _dev(val, torch.float32, "val"); _dev(count, torch.int32, "count")
        dirs = _dev(dirs, torch.float32, "dirs").reshape(-1, self.d_out)
        m, cap = feat.shape
        out = torch.empty(m, cap, dirs.shape[0], dtype=torch.float32, device=feat.device)
        with torch.cuda.device(feat.device):
            _check(lib().tsc_contrib_scores(self._handle, feat.data_ptr(), val.data_ptr(), count.data_ptr(), m, cap,
                                            dirs.data_ptr(), dirs.shape[0], out.data_ptr(), _stream()))
        return out


def sparse_lens_emit_many(handles, inputs, feat: torch.Tensor, val: torch.Tensor, count: torch.Tensor,
                          screen: bool = True, row_bounds: Optional[torch.Tensor] = None, streams=None,
                          count_is_zero: bool = True) -> None:
    """``tsc_sparse_lens_emit_many``: the emits of all ``handles`` into the stacked ``feat / val [D, M, cap]`` and
    ``count [D, M]`` in ONE foreign call, dictionary ``i`` on ``streams[i % len(streams)]`` (``torch.cuda.Stream``s;
    default: the current one).  ``row_bounds [D, 2]``: the screens' row bounds (device, :func:`screen_row_bound`), or None."""
    n = len(handles)
    _dev(feat, torch.int32, "feat"); _dev(val, torch.float32, "val"); _dev(count, torch.int32, "count")
    D, M, cap = feat.shape
    if D != n or tuple(count.shape) != (n, M) or tuple(val.shape) != (n, M, cap):
        raise TscError("sparse_lens_emit_many: stacked buffers do not match the number of dictionaries")
    hs = (ctypes.c_void_p * n)(*[h._handle for h in handles])
    xs = (ctypes.c_void_p * n)(*[_dev(x, torch.float32, "x").data_ptr() for x in inputs])
    for x, h in zip(inputs, handles):
        if x.dim() != 2 or x.shape[0] != M or x.shape[1] != h.d_in:
            raise TscError(f"sparse_lens_emit_many: every input must be [{M}, d_in] (got {tuple(x.shape)})")
    rb = None
    if row_bounds is not None:
        row_bounds = _dev(row_bounds, torch.float32, "row_bounds")
        if tuple(row_bounds.shape) != (n, 2):
            raise TscError("sparse_lens_emit_many: row_bounds must be [n_dicts, 2] (screen_row_bound)")
        rb = (ctypes.c_void_p * n)(*[row_bounds.data_ptr() + 8 * i for i in range(n)])
    st = [s.cuda_stream for s in streams] if streams else [_stream()]
    sa = (ctypes.c_void_p * len(st))(*st)
    with torch.cuda.device(feat.device):
        _check(lib().tsc_sparse_lens_emit_many(n, hs, xs, M, cap, feat.data_ptr(), val.data_ptr(), count.data_ptr(),
                                               (1 if count_is_zero else 0) | (0 if screen else 2), rb, sa, len(st)))


def sparse_lens_finish_many(handles, inputs, feat: torch.Tensor, val: torch.Tensor, count: torch.Tensor, dirs=None,
                            recon: Optional[torch.Tensor] = None, scores: Optional[torch.Tensor] = None,
                            screen: bool = True) -> None:
    """``tsc_sparse_lens_finish_many``: the second kernel of the lens pass for all ``handles`` (``DeviceDict``s, up to
    32) in one launch, after their ``sparse_lens_pass(..., emit_only=True)`` calls into the stacked buffers
    ``feat / val [D, M, cap]``, ``count [D, M]`` (in stream order).  ``inputs``: the ``[M, d_in]`` tensors the emits
    saw; ``dirs``: per-dictionary ``[M, d_out]`` tensors (or None); ``recon [D, M, d_out]`` / ``scores [D, M, cap]``."""
    n = len(handles)
    _dev(feat, torch.int32, "feat"); _dev(val, torch.float32, "val"); _dev(count, torch.int32, "count")
    D, M, cap = feat.shape
    if D != n or tuple(count.shape) != (n, M) or tuple(val.shape) != (n, M, cap):
        raise TscError("sparse_lens_finish_many: stacked buffers do not match the number of dictionaries")
    hs = (ctypes.c_void_p * n)(*[h._handle for h in handles])
    xs = (ctypes.c_void_p * n)(*[_dev(x, torch.float32, "x").data_ptr() for x in inputs])
    ds = None
    if scores is not None:
        if dirs is None:
            raise TscError("scores need directions")
        ds = (ctypes.c_void_p * n)(*[_dev(d, torch.float32, "dirs").data_ptr() for d in dirs])
        _dev(scores, torch.float32, "scores")
    if recon is not None:
        _dev(recon, torch.float32, "recon")
    with torch.cuda.device(feat.device):
        _check(lib().tsc_sparse_lens_finish_many(n, hs, xs, M, cap, feat.data_ptr(), val.data_ptr(), count.data_ptr(),
                                                 0 if screen else 2, ds, handles[0].d_out,
                                                 recon.data_ptr() if recon is not None else None,
                                                 scores.data_ptr() if scores is not None else None, _stream()))


class LensGroup:
    """The dictionaries of a CircuitLens pass with their inputs, marshalled ONCE for the two grouped calls
    (:func:`sparse_lens_emit_many`, :func:`sparse_lens_finish_many`): a caller that runs pass after pass over the same
    batch (one per direction set) pays the 24-entry ctypes tables once, not per call -- the emit call sits on the
    pass's critical path, with the GPU idle until it is queued."""

    def __init__(self, handles, inputs):
        self.handles, self.inputs = list(handles), list(inputs)
        n = len(self.handles)
        if n < 1 or len(self.inputs) != n:
            raise TscError("LensGroup: one input per dictionary")
        self.rows = int(self.inputs[0].shape[0])
        for x, h in zip(self.inputs, self.handles):
            _dev(x, torch.float32, "x")
            if x.dim() != 2 or x.shape[0] != self.rows or x.shape[1] != h.d_in:
                raise TscError(f"LensGroup: every input must be [{self.rows}, d_in] (got {tuple(x.shape)})")
        self.hs = (ctypes.c_void_p * n)(*[h._handle for h in self.handles])
        self.xs = (ctypes.c_void_p * n)(*[x.data_ptr() for x in self.inputs])
        self.d_out = self.handles[0].d_out
        self.device = self.inputs[0].device

    def emit(self, feat, val, count, screen=True, row_bounds=None, streams=None, count_is_zero=True) -> None:
        n = len(self.handles)
        D, M, cap = feat.shape
        if D != n or M != self.rows or tuple(count.shape) != (n, M) or tuple(val.shape) != (n, M, cap):
            raise TscError("LensGroup.emit: stacked buffers do not match the dictionaries")
        rb = None
        if row_bounds is not None:
            width = row_bounds.shape[-1]
            if row_bounds.dim() != 2 or row_bounds.shape[0] != n or width not in (2, 2 + 2 * M) \
                    or not row_bounds.is_contiguous() or row_bounds.dtype != torch.float32:
                raise TscError("LensGroup.emit: row_bounds must be a contiguous float32 [n_dicts, 2] or [n_dicts, 2 + 2 M] "
                               "(screen_row_bound)")
            base = row_bounds.data_ptr()
            if getattr(self, "_rb_base", None) != (base, width):
                self._rb, self._rb_base = (ctypes.c_void_p * n)(*range(base, base + 4 * width * n, 4 * width)), (base, width)
            rb = self._rb
            per_row_flag = 16 if width > 2 else 0
        st = [s.cuda_stream for s in streams] if streams else [_stream()]
        sa = (ctypes.c_void_p * len(st))(*st)
        with torch.cuda.device(self.device):
            _check(lib().tsc_sparse_lens_emit_many(n, self.hs, self.xs, M, cap, feat.data_ptr(), val.data_ptr(),
                                                   count.data_ptr(),
                                                   (1 if count_is_zero else 0) | (0 if screen else 2) | (per_row_flag if rb is not None else 0),
                                                   rb, sa, len(st)))

    def finish(self, feat, val, count, dirs=None, recon=None, scores=None, screen=True) -> None:
        n = len(self.handles)
        D, M, cap = feat.shape
        if D != n or M != self.rows:
            raise TscError("LensGroup.finish: stacked buffers do not match the dictionaries")
        ds = None
        if scores is not None:
            if dirs is None:
                raise TscError("scores need directions")
            ds = (ctypes.c_void_p * n)(*[_dev(d, torch.float32, "dirs").data_ptr() for d in dirs])
        with torch.cuda.device(self.device):
            _check(lib().tsc_sparse_lens_finish_many(n, self.hs, self.xs, M, cap, feat.data_ptr(), val.data_ptr(),
                                                     count.data_ptr(), 0 if screen else 2, ds, self.d_out,
                                                     recon.data_ptr() if recon is not None else None,
                                                     scores.data_ptr() if scores is not None else None, _stream()))


def screen_row_bound(x: torch.Tensor, b_dec: torch.Tensor, per_row: bool = False) -> torch.Tensor:
    """The ``screen_row_bound`` of ``tsc_sparse_lens_pass`` (include/tsc_hip.h) for rows ``x [..., M, d_in]``: a device
    tensor ``[..., 2]``, with ``xc = x - b_dec``, ``g = 2 d_in 2^-24``:
    ``[max_m (||bf16(xc_m) - xc_m|| + g ||xc_m|| + g (||x_m|| + ||b_dec||)),  max_m ||xc_m||]``, rounded up.
    Leading axes batch over dictionaries (``b_dec [..., d_in]``).  ``per_row``: ``[..., 2 + 2 M]`` -- the maxima followed by
    every row's own pair (``TSC_LENS_ROW_BOUNDS``: candidates are re-tested with their row's margin)."""
    g = 2.0 * x.shape[-1] * 2.0 ** -24
    xc = x - b_dec.unsqueeze(-2)
    err = (xc.to(torch.bfloat16).to(torch.float32) - xc).norm(dim=-1) + (x.shape[-1] * 1.2e-38) ** 0.5
    centred = xc.norm(dim=-1)
    first = err + g * centred + g * (x.norm(dim=-1) + b_dec.norm(dim=-1).unsqueeze(-1))
    maxima = torch.stack([first.amax(dim=-1), centred.amax(dim=-1)], dim=-1)
    if per_row:
        rows = torch.stack([first, centred], dim=-1).flatten(-2)                    # [..., 2 M]: (A_0, B_0, A_1, B_1, ...)
        maxima = torch.cat([maxima, rows], dim=-1)
    return (maxima * 1.0001).to(torch.float32).contiguous()


def lens_assemble(head: torch.Tensor, aux: torch.Tensor, group_aux, scores: torch.Tensor, count: torch.Tensor,
                  k: int = 10, dense: bool = True, column_tags: Optional[torch.Tensor] = None,
                  feat: Optional[torch.Tensor] = None, top_out: Optional[torch.Tensor] = None):
    """``tsc_lens_assemble``: normalise-and-rank tail of the CircuitLens pass for ``R`` rows in one launch.
    ``head [R, n_head]``, ``aux [R, sum(group_aux)]``, ``scores [G, R, cap]``, ``count [G, R]`` (int32); optional
    ``column_tags [n_head + sum(group_aux) + G]`` (int32 labels of the column classes) and ``feat [G, R, cap]``.
    Returns ``{"raw", "activations", "mask"}`` (``[R, W]``, only when ``dense``) plus ``top_values [R, k]``,
    ``top_info [R, k, 4]`` (int32: column, tag >> 16, tag & 0xffff, feature id) and its view ``top_columns``
    (-inf / -1 where a row has fewer than k live columns).  ``top_out``: optional preallocated int32 buffer of ``5 R k``
    elements that receives ``top_info`` (first ``4 R k``) and the bits of ``top_values`` (one device-to-host copy then
    fetches both)."""
    scores = _dev(scores, torch.float32, "scores")
    count = _dev(count, torch.int32, "count")
    G, R, cap = scores.shape
    group_aux = [int(a) for a in group_aux]
    head = _dev(head.reshape(R, -1) if R else head.reshape(0, head.shape[-1] if head.dim() > 1 else 0), torch.float32, "head")
    aux = _dev(aux.reshape(R, -1) if R else aux.reshape(0, sum(group_aux)), torch.float32, "aux")
    if len(group_aux) != G or aux.shape[1] != sum(group_aux) or tuple(count.shape) != (G, R):
        raise TscError("lens_assemble: group_aux / aux / count do not match scores [G, R, cap]")
    if column_tags is not None:
        column_tags = _dev(column_tags, torch.int32, "column_tags")
        if column_tags.numel() != head.shape[1] + sum(group_aux) + G:
            raise TscError("lens_assemble: column_tags needs one entry per head scalar, auxiliary scalar and group")
    if feat is not None:
        feat = _dev(feat, torch.int32, "feat")
        if tuple(feat.shape) != (G, R, cap):
            raise TscError("lens_assemble: feat must be [G, R, cap] like scores")
    W = head.shape[1] + sum(group_aux) + G * cap
    dev = scores.device
    if top_out is not None:
        if top_out.dtype != torch.int32 or top_out.numel() < 5 * R * k or not top_out.is_contiguous() or top_out.device != dev:
            raise TscError("lens_assemble: top_out must be a contiguous int32 buffer of 5 R k elements on the scores' device")
        flat = top_out.reshape(-1)
        out = {"top_values": flat[4 * R * k:5 * R * k].view(torch.float32).view(R, k), "top_info": flat[:4 * R * k].view(R, k, 4)}
    else:
        out = {"top_values": torch.empty(R, k, dtype=torch.float32, device=dev),
               "top_info": torch.empty(R, k, 4, dtype=torch.int32, device=dev)}
    if dense or W > 15360:  # rows that do not fit in LDS are ranked out of the dense outputs
        out["activations"] = torch.empty(R, W, dtype=torch.float32, device=dev)
        out["raw"] = torch.empty(R, W, dtype=torch.float32, device=dev)
        out["mask"] = torch.empty(R, W, dtype=torch.bool, device=dev)
    ga = (ctypes.c_int32 * G)(*group_aux)
    ptr = lambda n: out[n].data_ptr() if n in out else None  # noqa: E731
    with torch.cuda.device(dev):
        _check(lib().tsc_lens_assemble(R, head.shape[1], G, ga, cap, head.data_ptr() if head.numel() else None,
                                       aux.data_ptr() if aux.numel() else None, scores.data_ptr(), count.data_ptr(),
                                       column_tags.data_ptr() if column_tags is not None else None,
                                       feat.data_ptr() if feat is not None else None, k,
                                       ptr("raw"), ptr("activations"), ptr("mask"), out["top_values"].data_ptr(),
                                       out["top_info"].data_ptr(), _stream()))
    out["top_columns"] = out["top_info"][:, :, 0]
    return out


class ContribGroup:
    """``tsc_contrib_scores_many`` for a fixed list of dictionaries, marshalled once: scores of a window of rows of
    stacked pass outputs ``feat / val [D, R, cap]``, ``count [D, R]`` (any slices of a larger pass along R) against
    per-dictionary directions ``dirs [D, M, d_out]`` (or ``[D, d_out]``: one direction for all rows of a dictionary)."""

    def __init__(self, handles):
        self.handles = list(handles)
        n = len(self.handles)
        self.hs = (ctypes.c_void_p * n)(*[h._handle for h in self.handles])
        self.ds = (ctypes.c_void_p * n)()
        self.d_out = self.handles[0].d_out if n else 0

    def scores(self, feat, val, count, row0: int, rows: int, dirs: Optional[torch.Tensor], out: torch.Tensor,
               feat_out: Optional[torch.Tensor] = None, count_out: Optional[torch.Tensor] = None, n_dicts: Optional[int] = None,
               dir_stride: int = 0):
        """``dirs=None``: the direction pointers the caller has put into ``self.ds`` (persistent buffers: nothing to
        marshal per call), ``dir_stride`` elements apart from row to row (0: one direction for all rows)."""
        n = len(self.handles) if n_dicts is None else int(n_dicts)
        if n == 0 or rows == 0:
            return out
        cap = feat.shape[2]
        if feat.stride(2) != 1 or feat.stride(1) != cap or val.stride() != feat.stride() or count.stride(1) != 1 \
                or count.stride(0) * cap != feat.stride(0):
            raise TscError("ContribGroup: feat / val / count must be row slices of stacked [D, R, cap] / [D, R] buffers")
        if feat.dtype != torch.int32 or val.dtype != torch.float32 or count.dtype != torch.int32 or not feat.is_cuda:
            raise TscError("ContribGroup: feat / count int32, val float32, on the GPU")
        _dev(out, torch.float32, "out")
        if out.numel() < n * rows * cap:
            raise TscError("ContribGroup: out must hold [D, M, cap]")
        stride = int(dir_stride)
        if dirs is not None:
            _dev(dirs, torch.float32, "dirs")
            stride = 0 if dirs.dim() == 2 else dirs.stride(1)
            if dirs.shape[0] < n or dirs.shape[-1] != self.d_out:
                raise TscError("ContribGroup: dirs must be [D, (M,) d_out]")
            base, step = dirs.data_ptr(), dirs.stride(0) * 4
            for i in range(n):
                self.ds[i] = base + i * step
        with torch.cuda.device(feat.device):
            _check(lib().tsc_contrib_scores_many(
                n, self.hs, rows, count.stride(0), cap, feat.data_ptr() + 4 * row0 * cap,
                val.data_ptr() + 4 * row0 * cap, count.data_ptr() + 4 * row0, self.ds, stride, out.data_ptr(),
                feat_out.data_ptr() if feat_out is not None else None,
                count_out.data_ptr() if count_out is not None else None, 0, _stream()))
        return out


class LensQuery:
    """The three library calls of ONE per-node lens query -- ``tsc_contrib_scores_many`` over the first ``n_groups``
    dictionaries at one row of a finished pass, ``tsc_rowdot_many`` (the caller's :class:`RowDotTable` for that
    position), ``tsc_lens_assemble`` -- with every argument that does not change between queries marshalled once.
    The circuit search makes thousands of such queries; at a few microseconds of GPU work each, the host side of the
    call IS its cost.  All buffers are the caller's persistent ones (``buf``: name -> tensor, see CircuitLens)."""

    def __init__(self, group: "ContribGroup", feat, val, count, buf: dict, group_aux):
        cap = feat.shape[2]
        if feat.stride(2) != 1 or feat.stride(1) != cap or val.stride() != feat.stride() or count.stride(1) != 1 \
                or count.stride(0) * cap != feat.stride(0):
            raise TscError("LensQuery: feat / val / count must be row slices of stacked [D, R, cap] / [D, R] buffers")
        self.group, self.cap, self.pitch = group, cap, count.stride(0)
        self.feat_ptr, self.val_ptr, self.count_ptr = feat.data_ptr(), val.data_ptr(), count.data_ptr()
        self.keep = (feat, val, count, buf)
        self.device = feat.device
        self.p = {n: t.data_ptr() for n, t in buf.items()}
        self.group_aux = [int(a) for a in group_aux]
        self.ga = (ctypes.c_int32 * len(self.group_aux))(*self.group_aux)
        self.top: dict = {}
        self.fn = lib()

    def run(self, row: int, n_groups: int, k: int, table: "RowDotTable", tags: torch.Tensor, out_cap: int = 0) -> torch.Tensor:
        """Returns a pinned host int32 tensor ``[5 k]``: ``top_info [k, 4]`` then the bits of ``top_values [k]``.
        ``out_cap``: list length of the compact per-query buffers (>= the row's largest count; 0 = the pass's capacity)."""
        fn, p, cap = self.fn, self.p, self.cap
        oc = cap if out_cap <= 0 or out_cap > cap else int(out_cap)
        wide = 2 + sum(self.group_aux[:n_groups]) + n_groups * oc > 15360
        if wide and "raw" not in p:
            raise TscError("LensQuery: a row this wide needs the raw / acts / mask buffers")
        # (a prefix of the auxiliary columns of ONE row is contiguous: the same buffers serve every cut-off)
        top = self.top.get(k)
        if top is None:
            top = self.top[k] = (torch.empty(5 * k, dtype=torch.int32, device=self.device),
                                 torch.empty(5 * k, dtype=torch.int32).pin_memory())
        dev_top, host_top = top
        stream = torch.cuda.current_stream(self.device)
        s = stream.cuda_stream
        with torch.cuda.device(self.device):
            _check(fn.tsc_contrib_scores_many(n_groups, self.group.hs, 1, self.pitch, cap, self.feat_ptr + 4 * row * cap,
                                              self.val_ptr + 4 * row * cap, self.count_ptr + 4 * row, self.group.ds, 0,
                                              p["scores"], p["feat"], p["count"], oc, s))
            _check(fn.tsc_rowdot_many(table.n, table.a, table.sa, table.c, table.sc, table.b, table.sb, table.out, table.so,
                                      1, table.d, s))
            _check(fn.tsc_lens_assemble(1, 2, n_groups, self.ga, oc, p["head"], p["aux"], p["scores"], p["count"],
                                        tags.data_ptr(), p["feat"], k, p.get("raw") if wide else None,
                                        p.get("acts") if wide else None, p.get("mask") if wide else None,
                                        dev_top.data_ptr() + 16 * k,
                                        dev_top.data_ptr(), s))
            host_top.copy_(dev_top, non_blocking=True)
            stream.synchronize()
        return host_top


class RowDotTable:
    """Term table of ``tsc_rowdot_many``, kept as ctypes arrays so that a caller that runs the same set of dot products
    again and again (the auxiliary rows of every CircuitLens direction pass) only patches the pointers that changed.
    A term is ``(a, c or None, b, out)``: ``a / c / b`` tensors ``[R, d]`` (per row) or ``[d]`` (shared), ``out`` a 1-D
    view ``[R]`` (any stride) of the output buffer.  The table keeps NO reference to the tensors: the caller does."""

    def __init__(self, n_terms: int, rows: int, d: int):
        self.n, self.rows, self.d = int(n_terms), int(rows), int(d)
        arr, i64 = ctypes.c_void_p * self.n, ctypes.c_int64 * self.n
        self.a, self.c, self.b, self.out = arr(), arr(), arr(), arr()
        self.sa, self.sc, self.sb, self.so = i64(), i64(), i64(), i64()

    def _operand(self, t: Optional[torch.Tensor], name: str):
        if t is None:
            return None, 0
        if not t.is_cuda or t.dtype != torch.float32 or t.stride(-1) != 1 or t.shape[-1] != self.d:
            raise TscError(f"rowdot: {name} must be a float32 GPU tensor [..., {self.d}] with unit inner stride")
        if t.dim() == 1:
            return t.data_ptr(), 0
        if t.dim() != 2 or t.shape[0] != self.rows:
            raise TscError(f"rowdot: {name} must be [{self.rows}, {self.d}] or [{self.d}] (got {tuple(t.shape)})")
        return t.data_ptr(), t.stride(0)

    def set(self, j: int, a, c, b, out) -> None:
        self.a[j], self.sa[j] = self._operand(a, "a")
        self.c[j], self.sc[j] = self._operand(c, "c")
        self.b[j], self.sb[j] = self._operand(b, "b")
        if not out.is_cuda or out.dtype != torch.float32 or out.dim() != 1 or out.shape[0] != self.rows:
            raise TscError(f"rowdot: out must be a float32 GPU view [{self.rows}]")
        self.out[j], self.so[j] = out.data_ptr(), max(1, out.stride(0))

    def run(self, device) -> None:
        with torch.cuda.device(device):
            _check(lib().tsc_rowdot_many(self.n, self.a, self.sa, self.c, self.sc, self.b, self.sb, self.out, self.so,
                                         self.rows, self.d, _stream()))


def rowdot_many(terms, rows: int, d: int, device) -> None:
    """One-shot :class:`RowDotTable`: ``terms`` = list of ``(a, c or None, b, out view [rows])``."""
    tab = RowDotTable(len(terms), rows, d)
    for j, (a, c, b, out) in enumerate(terms):
        tab.set(j, a, c, b, out)
    tab.run(device)


class TopkState:
    """Caller-owned running top-k lists: ``vals [n_slots, d_sae, k]`` fp32 and ``idx`` int64."""

    def __init__(self, d_sae: int, k: int = 20, device="cuda", n_slots: Optional[int] = None,
                 cross_slot_floor: bool = True):
        if n_slots is None:
            n_slots = lib().tsc_topk_recommended_slots(d_sae)
        self.d_sae, self.k, self.n_slots = int(d_sae), int(k), int(n_slots)
        if not (1 <= self.k <= 64 and 1 <= self.n_slots <= 64) or self.n_slots * self.k > TOPK_MERGE_MAX:
            raise TscError(f"top-k state needs 1 <= k <= 64, 1 <= n_slots <= 64 and n_slots * k <= {TOPK_MERGE_MAX} "
                           f"(got k={self.k}, n_slots={self.n_slots}): its slots are merged in one pass")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise TscError("top-k state must live on the GPU")
        self.vals = torch.empty(self.n_slots, d_sae, k, dtype=torch.float32, device=self.device)
        self.idx = torch.empty(self.n_slots, d_sae, k, dtype=torch.int64, device=self.device)
        # scratch for the k-th best over all slots (see include/tsc_hip.h); not part of the state's content
        self.floor = torch.zeros(d_sae, dtype=torch.float32, device=self.device) if cross_slot_floor else None
        self._calls = 0
        self._struct = _TopkStateStruct(self.vals.data_ptr(), self.idx.data_ptr(), d_sae, k, self.n_slots, 0,
                                        self.floor.data_ptr() if cross_slot_floor else None)
        self.reset()

    FLOOR_EVERY = 4      # once the lists are warm the cross-slot floor is recomputed on every 4th call only:
    FLOOR_WARM_CALLS = 8  # an older floor is still a valid bound (include/tsc_hip.h, TSC_TOPK_FLOOR_KEEP)

    def reset(self) -> None:
        with torch.cuda.device(self.device):
            _check(lib().tsc_topk_state_reset(ctypes.byref(self._struct), _stream()))
        self.invalidate_floor()

    def invalidate_floor(self) -> None:
        """Call after writing ``vals`` / ``idx`` directly: the next encode recomputes the floor."""
        self._calls = 0

    def _schedule_floor(self) -> None:
        keep = self._calls >= self.FLOOR_WARM_CALLS and self._calls % self.FLOOR_EVERY != 0
        self._struct.flags = 1 if keep else 0
        self._calls += 1

    def screen_workspace(self) -> torch.Tensor:
        """Candidate queues of ``tsc_encode_topk_screened`` (allocated on first use, scratch between calls)."""
        if self.floor is None:
            raise TscError("the screened top-k needs the state's floor scratch (cross_slot_floor=True)")
        ws = getattr(self, "_screen_ws", None)
        if ws is None:
            n = int(lib().tsc_topk_screen_workspace_bytes(self.d_sae, self.n_slots))
            ws = self._screen_ws = torch.empty(n, dtype=torch.uint8, device=self.device)
        return ws

    def finalize(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """Merged, sorted ``(vals [d_sae,k], idx [d_sae,k])``; the state stays usable."""
        ov = torch.empty(self.d_sae, self.k, dtype=torch.float32, device=self.device)
        oi = torch.empty(self.d_sae, self.k, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _check(lib().tsc_topk_finalize(ctypes.byref(self._struct), ov.data_ptr(), oi.data_ptr(), _stream()))
        return ov, oi

    def load(self, vals: torch.Tensor, idx: torch.Tensor) -> None:
        """Resume from finalized lists: they become slot 0, the other slots are emptied."""
        self.reset()
        self.vals[0].copy_(vals)
        self.idx[0].copy_(idx)
        self.invalidate_floor()


def topk_merge(vals: torch.Tensor, idx: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merge ``[n_lists, d_sae, k]`` list sets (e.g. the all-gathered per-rank lists)."""
    vals = _dev(vals, torch.float32, "vals")
    idx = _dev(idx, torch.int64, "idx")
    n_lists, d_sae, k = vals.shape
    ov = torch.empty(d_sae, k, dtype=torch.float32, device=vals.device)
    oi = torch.empty(d_sae, k, dtype=torch.int64, device=vals.device)
    with torch.cuda.device(vals.device):
        _check(lib().tsc_topk_merge(vals.data_ptr(), idx.data_ptr(), n_lists, d_sae, k, ov.data_ptr(), oi.data_ptr(),
                                    _stream()))
    return ov, oi


class FeatureProjector:
    """``act((x - b_dec) @ Pt.T + bias)`` for a fixed handful of encoder columns ``Pt [F, d_in]``.

    Six or more columns (and ``d_in % 64 == 0``) are registered as a small dictionary of their own, so
    the projection of a long token stream runs on the MFMA encode kernel as a skinny GEMM (one 128-wide
    column tile; 11x faster than the per-token dot products at F = 64); fewer columns, short inputs
    and odd ``d_in`` use ``tsc_project_features``.  Make one per feature set and reuse it across batches.
    Holds COPIES of the selected columns: rebuild it after the source dictionary changes."""

    MIN_ROWS_FOR_GEMM = 1024
    MIN_FEATURES_FOR_GEMM = 6  # measured crossover with the per-row kernel (60 + 17.5 F us vs a flat ~100 us per 65536 rows)

    def __init__(self, Pt: torch.Tensor, bias: torch.Tensor, b_dec: Optional[torch.Tensor] = None):
        self.Pt = _dev(Pt, torch.float32, "Pt").contiguous()
        self.bias = _dev(bias, torch.float32, "bias").contiguous()
        self.b_dec = _dev(b_dec, torch.float32, "b_dec").contiguous() if b_dec is not None else None
        self.n_feat, self.d_in = self.Pt.shape
        self._dict = None
        if self.n_feat >= self.MIN_FEATURES_FOR_GEMM and self.d_in % 64 == 0:
            pad = (-self.n_feat) % 4  # d_sae must be a multiple of 4: repeat the last column
            pt = torch.cat([self.Pt, self.Pt[-1:].expand(pad, -1)]) if pad else self.Pt
            bs = torch.cat([self.bias, self.bias[-1:].expand(pad)]) if pad else self.bias
            bd = self.b_dec if self.b_dec is not None else torch.zeros(self.d_in, dtype=torch.float32, device=Pt.device)
            self._dict = DeviceDict(pt.t().contiguous(), bs.contiguous(), bd)

    def __call__(self, x: torch.Tensor, relu: bool = False) -> torch.Tensor:
        x2 = _dev(x, torch.float32, "x").reshape(-1, x.shape[-1])
        if self._dict is not None and x2.shape[0] >= self.MIN_ROWS_FOR_GEMM:
            out = self._dict.encode_dense(x2) if relu else self._dict.encode_pre(x2)
            return out[:, :self.n_feat].contiguous() if out.shape[1] != self.n_feat else out
        return _project_rows(x2, self.Pt, self.bias, self.b_dec, relu)

    def close(self) -> None:
        if self._dict is not None:
            self._dict.close()
            self._dict = None


def project_features(x: torch.Tensor, Pt: torch.Tensor, bias: torch.Tensor, b_dec: Optional[torch.Tensor] = None,
                     relu: bool = False) -> torch.Tensor:
    """One-shot :class:`FeatureProjector` (``[M, F]`` output)."""
    x2 = _dev(x, torch.float32, "x").reshape(-1, x.shape[-1])
    if (Pt.shape[0] >= FeatureProjector.MIN_FEATURES_FOR_GEMM and x2.shape[0] >= FeatureProjector.MIN_ROWS_FOR_GEMM
            and x2.shape[1] % 64 == 0):
        proj = FeatureProjector(Pt, bias, b_dec)
        try:
            return proj(x2, relu)
        finally:
            proj.close()
    return _project_rows(x2, _dev(Pt, torch.float32, "Pt"), _dev(bias, torch.float32, "bias"), b_dec, relu)


def _project_rows(x2: torch.Tensor, Pt: torch.Tensor, bias: torch.Tensor, b_dec: Optional[torch.Tensor],
                  relu: bool) -> torch.Tensor:
    """``tsc_project_features``: one wavefront per token row, features processed 64 per launch."""
    if b_dec is not None:
        _dev(b_dec, torch.float32, "b_dec")
    m, d_in = x2.shape
    n_feat = Pt.shape[0]
    parts = []
    with torch.cuda.device(x2.device):
        for s in range(0, n_feat, 64):
            pt = Pt[s:s + 64].contiguous()
            bs = bias[s:s + 64].contiguous()
            out = torch.empty(m, pt.shape[0], dtype=torch.float32, device=x2.device)
            _check(lib().tsc_project_features(x2.data_ptr(), m, d_in, b_dec.data_ptr() if b_dec is not None else None,
                                              pt.data_ptr(), bs.data_ptr(), pt.shape[0], int(relu), out.data_ptr(),
                                              _stream()))
            parts.append(out)
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
