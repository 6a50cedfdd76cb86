"""Attention-output (hook_z) sparse autoencoder with the reference's surface and a HIP backend.

Drop-in for ``ZSAE`` of reference ``src/z_sae.py``: ``cfg`` dict with ``dict_size, l1_coeff,
enc_dtype, seed, act_size, device`` (:30-36,52), parameters ``W_enc [act_size,d_hidden]``, ``W_dec``,
``b_enc``, ``b_dec`` (:34-47), ``encode(x)`` (:73-77) and ``forward(x, per_token=False)`` returning
``(loss, x_reconstruct, acts, l2_loss, l1_loss)`` (:57-71).
"""
from __future__ import annotations

import torch
from torch import nn

from _dictionary import HipDictionaryMixin
from dtypes import DTYPES

# layer -> run name on the HF hub (reference :10-23).  Entry 10 names an L0 run there too; kept
# as found because it decides which weights a user of the reference actually gets.
RUN_DICT = {
    0: "gpt2-small_L0_Hcat_z_lr1.20e-03_l11.80e+00_ds24576_bs4096_dc1.00e-06_rsanthropic_rie25000_nr4_v9",
    1: "gpt2-small_L1_Hcat_z_lr1.20e-03_l18.00e-01_ds24576_bs4096_dc1.00e-06_rsanthropic_rie25000_nr4_v5",
    2: "gpt2-small_L2_Hcat_z_lr1.20e-03_l11.00e+00_ds24576_bs4096_dc1.00e-06_rsanthropic_rie25000_nr4_v4",
    3: "gpt2-small_L3_Hcat_z_lr1.20e-03_l19.00e-01_ds24576_bs4096_dc1.00e-06_rsanthropic_rie25000_nr4_v9",
    4: "gpt2-small_L4_Hcat_z_lr1.20e-03_l11.10e+00_ds24576_bs4096_dc1.00e-06_rsanthropic_rie25000_nr4_v7",
    5: "gpt2-small_L5_Hcat_z_lr1.20e-03_l11.00e+00_ds49152_bs4096_dc1.00e-06_rsanthropic_rie25000_nr4_v9",
    6: "gpt2-small_L6_Hcat_z_lr1.20e-03_l11.10e+00_ds24576_bs4096_dc1.00e-06_rsanthropic_rie25000_nr4_v9",
    7: "gpt2-small_L7_Hcat_z_lr1.20e-03_l11.10e+00_ds49152_bs4096_dc1.00e-06_rsanthropic_rie25000_nr4_v9",
    8: "gpt2-small_L8_Hcat_z_lr1.20e-03_l11.30e+00_ds24576_bs4096_dc1.00e-05_rsanthropic_rie25000_nr4_v6",
    9: "gpt2-small_L9_Hcat_z_lr1.20e-03_l11.20e+00_ds24576_bs4096_dc1.00e-06_rsanthropic_rie25000_nr4_v9",
    10: "gpt2-small_L0_Hcat_z_lr1.20e-03_l11.80e+00_ds24576_bs4096_dc1.00e-06_rsanthropic_rie25000_nr4_v9",
    11: "gpt2-small_L11_Hcat_z_lr1.20e-03_l13.00e+00_ds24576_bs4096_dc3.16e-06_rsanthropic_rie25000_nr4_v9",
}


class ZSAE(HipDictionaryMixin, nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        d_hidden = cfg["dict_size"]
        dtype = DTYPES[cfg["enc_dtype"]]
        torch.manual_seed(cfg["seed"])
        self.W_enc = nn.Parameter(nn.init.kaiming_uniform_(torch.empty(cfg["act_size"], d_hidden, dtype=dtype)))
        self.W_dec = nn.Parameter(nn.init.kaiming_uniform_(torch.empty(d_hidden, cfg["act_size"], dtype=dtype)))
        self.b_enc = nn.Parameter(torch.zeros(d_hidden, dtype=dtype))
        self.b_dec = nn.Parameter(torch.zeros(cfg["act_size"], dtype=dtype))
        with torch.no_grad():
            self.W_dec.data[:] = self.W_dec / self.W_dec.norm(dim=-1, keepdim=True)
        self.d_hidden = d_hidden
        self.l1_coeff = cfg["l1_coeff"]
        self.dtype = dtype
        self.device = cfg["device"]
        self.version = 0
        self.to(cfg["device"])

    def _out_bias(self) -> torch.Tensor:
        return self.b_dec

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.device = self.W_enc.device
        return out

    def encode(self, x):
        self._refuse_autograd(x)
        x = x.to(self.W_enc.device)
        return self.device_dict().encode_dense(x.to(torch.float32).contiguous())

    def forward(self, x, per_token=False):
        x = x.to(self.W_enc.device)
        acts = self.encode(x)
        x_reconstruct = self.device_dict().decode_dense_acts(acts) if not torch.is_grad_enabled() else None
        if x_reconstruct is None:  # few rows (or an input the sparse decode does not take): the dense product
            x_reconstruct = torch.matmul(acts, self.W_dec) + self.b_dec
        l2 = (x_reconstruct.float() - x.float()).pow(2).sum(-1)
        l1 = self.l1_coeff * acts.float().abs().sum(dim=-1)
        if not per_token:
            l2 = l2.mean(0)
            l1 = l1.mean(dim=0)
        return l2 + l1, x_reconstruct, acts, l2, l1

    @classmethod
    def load_from_files(cls, cfg_json: str, weights_pt: str, device=None):
        """Local ``<run>_cfg.json`` + ``<run>.pt`` pair (what reference :85-88 downloads)."""
        import json

        cfg = json.load(open(cfg_json))
        cfg["device"] = device or ("cuda" if torch.cuda.is_available() else "cpu")
        self = cls(cfg=cfg)
        self.load_state_dict(torch.load(weights_pt, weights_only=True, map_location="cpu"))
        return self

    @classmethod
    def load_from_hf(cls, version, hf_repo="ckkissane/tinystories-1M-SAES"):
        from huggingface_hub import hf_hub_download

        return cls.load_from_files(hf_hub_download(hf_repo, f"{version}_cfg.json"), hf_hub_download(hf_repo, f"{version}.pt"))

    @classmethod
    def load_zsae_for_layer(cls, layer: int):
        return cls.load_from_hf(RUN_DICT[layer], hf_repo="ckkissane/attn-saes-gpt2-small-all-layers")
