"""MLP transcoder with the reference's Python surface and an MI355X HIP backend.

Drop-in for ``SparseTranscoder`` of reference ``src/mlp_transcoder.py`` on the analysis path:
same constructor (``cfg`` with ``d_in, d_sae, l1_coefficient, dtype, device, is_transcoder, d_out,
is_sparse_connection, use_ghost_grads, hook_point, hook_point_layer``), same parameter names and
shapes (:70-99), ``encode(x)`` (:194-213) and ``forward(x)`` returning the 6-tuple ``(sae_out,
feature_acts, loss, mse_loss, l1_loss, mse_loss_ghost_resid)`` (:108-192).  ``encode`` runs the
hand-written fp32-MFMA kernel through the C-ABI; the decoder half of ``forward`` is a plain library
GEMM.  Training helpers (resampling, ghost grads, ``get_test_loss``, ``save_model``) are out of scope
 and absent.
"""
from __future__ import annotations


import torch
from torch import nn

import tsc_hip
from _dictionary import HipDictionaryMixin, HookPoint


class SparseTranscoder(HipDictionaryMixin, nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.d_in = cfg.d_in
        if not isinstance(self.d_in, int):
            raise ValueError(f"d_in must be an int but was {self.d_in=}; {type(self.d_in)=}")
        self.d_sae = cfg.d_sae
        self.l1_coefficient = cfg.l1_coefficient
        self.dtype = cfg.dtype
        self.device = cfg.device
        self.d_out = self.d_in
        if cfg.is_transcoder and cfg.d_out is not None:
            self.d_out = cfg.d_out
        if getattr(cfg, "is_sparse_connection", False):
            raise NotImplementedError("sparse-connection training is outside the accelerated analysis path")

        # same draw order as the reference: W_enc, b_enc, W_dec, b_dec (:69-93)
        self.W_enc = nn.Parameter(nn.init.kaiming_uniform_(torch.empty(self.d_in, self.d_sae, dtype=self.dtype, device=self.device)))
        self.b_enc = nn.Parameter(torch.zeros(self.d_sae, dtype=self.dtype, device=self.device))
        self.W_dec = nn.Parameter(nn.init.kaiming_uniform_(torch.empty(self.d_sae, self.d_out, dtype=self.dtype, device=self.device)))
        with torch.no_grad():
            self.W_dec.data /= torch.norm(self.W_dec.data, dim=1, keepdim=True)
        self.b_dec = nn.Parameter(torch.zeros(self.d_in, dtype=self.dtype, device=self.device))
        self.b_dec_out = None
        if cfg.is_transcoder:
            self.b_dec_out = nn.Parameter(torch.zeros(self.d_out, dtype=self.dtype, device=self.device))

        self.hook_sae_in = HookPoint()
        self.hook_hidden_pre = HookPoint()
        self.hook_hidden_post = HookPoint()
        self.hook_sae_out = HookPoint()

    def _out_bias(self) -> torch.Tensor:
        return self.b_dec_out if self.cfg.is_transcoder else self.b_dec

    def _apply(self, fn, *a, **k):  # keep cfg-style .device in step with .to()/.cuda()
        out = super()._apply(fn, *a, **k)
        self.device = self.W_enc.device
        return out

    def _check_inner_hooks(self):
        for name in ("hook_sae_in", "hook_hidden_pre"):
            if len(getattr(self, name)._forward_hooks) > 0:
                raise NotImplementedError(
                    f"{name} has a hook attached, but x - b_dec and the pre-activation only exist inside the "
                    "fused kernel; hook hook_hidden_post / hook_sae_out instead")

    # ------------------------------------------------------------------ reference surface
    def encode(self, x):
        """``relu((x - b_dec) @ W_enc + b_enc)``; x is moved to the module's dtype/device first."""
        self._check_inner_hooks()
        self._refuse_autograd(x)
        x = x.to(self.dtype).to(self.W_enc.device)
        acts = self.device_dict().encode_dense(x.contiguous())
        return self.hook_hidden_post(acts)

    def forward(self, x, dead_neuron_mask=None, mse_target=None):
        x = x.to(self.dtype).to(self.W_enc.device)
        feature_acts = self.encode(x)
        # the decoder half on the sparse form (the activations are ~0.3 % dense): tsc_dense_to_sparse + tsc_decode_sparse
        # read the activations once and gather the active W_dec rows; the dense product stays for few rows / odd inputs
        recon = self.device_dict().decode_dense_acts(feature_acts) if not torch.is_grad_enabled() else None
        if recon is None:
            recon = torch.matmul(feature_acts, self.W_dec) + self._out_bias()
        sae_out = self.hook_sae_out(recon)
        target = x if mse_target is None else mse_target
        mse_loss = (torch.pow(sae_out - target.float(), 2) / (target**2).sum(dim=-1, keepdim=True).sqrt()).mean()
        if getattr(self.cfg, "use_ghost_grads", False) and self.training:
            raise NotImplementedError("ghost gradients are a training feature; out of scope")
        mse_loss_ghost_resid = torch.tensor(0.0, dtype=self.dtype, device=self.W_enc.device)
        sparsity = torch.abs(feature_acts).sum(dim=1).mean(dim=(0,))
        l1_loss = self.l1_coefficient * sparsity
        loss = mse_loss + l1_loss + mse_loss_ghost_resid
        return sae_out, feature_acts, loss, mse_loss, l1_loss, mse_loss_ghost_resid

    # ------------------------------------------------------------------ loaders (local files only)
    @classmethod
    def load_from_pretrained(cls, path: str, device=None):
        """Local ``{"cfg", "state_dict"}`` checkpoint (reference :602-659).  ``.pt`` files are read
        with ``torch.load(weights_only=True)``; pickled ``.pkl``/``.pkl.gz`` files execute code on
        load and are refused unless the caller wrote them (``trusted=True`` is deliberately not
        offered: convert them to ``.pt`` state dicts instead)."""
        if not path.endswith(".pt"):
            raise ValueError(f"only .pt checkpoints are loaded (weights_only=True); got {path}")
        blob = torch.load(path, weights_only=True, map_location="cpu")
        sd = blob["state_dict"] if "state_dict" in blob else blob
        from types import SimpleNamespace

        d_in, d_sae = sd["W_enc"].shape
        dev = device or ("cuda" if torch.cuda.is_available() else "cpu")
        raw_cfg = blob.get("cfg", {}) if isinstance(blob, dict) else {}
        get = (lambda k, d: raw_cfg.get(k, d)) if isinstance(raw_cfg, dict) else (lambda k, d: getattr(raw_cfg, k, d))
        cfg = SimpleNamespace(
            d_in=int(d_in), d_sae=int(d_sae), l1_coefficient=get("l1_coefficient", 0.0), dtype=torch.float32,
            device=dev, is_transcoder="b_dec_out" in sd, d_out=int(sd["W_dec"].shape[1]),
            is_sparse_connection=False, use_ghost_grads=False,
            hook_point=get("hook_point", "blocks.0.ln2.hook_normalized"), hook_point_layer=get("hook_point_layer", 0),
        )
        m = cls(cfg)
        m.load_state_dict({k: v for k, v in sd.items() if k in m.state_dict()})
        return m

    @classmethod
    def load_from_hugging_face(cls, layer: int):
        """Reference :590-600 downloads ``pchlenski/gpt2-transcoders``; there is no network here."""
        from huggingface_hub import hf_hub_download

        path = hf_hub_download(repo_id="pchlenski/gpt2-transcoders",
                               filename=f"final_sparse_autoencoder_gpt2-small_blocks.{layer}.ln2.hook_normalized_24576.pt")
        return cls.load_from_pretrained(path)
