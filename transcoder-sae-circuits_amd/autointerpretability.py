"""Multi-feature score scan with the reference's surface (``src/autointerpretability.py:186-274``).

``get_feature_scores`` returns float16 ``[n_seq, n_feat, n_pos]`` exactly as the reference lays it out
(:249-257) but computes only the requested columns (``tsc_project_features``) instead of the whole
``[tokens, d_sae]`` product.  HTML display, ``CircuitPrediction`` and the LLM prompt code are out of scope.
"""
from __future__ import annotations

import numpy as np
import torch

import tsc_hip
from mlp_transcoder import SparseTranscoder
from z_sae import ZSAE


def get_feature_scores(model, encoder, tokens_arr, feature_indices, batch_size=64, act_name=None,
                       use_raw_scores=False, use_decoder=False, feature_post=None, ignore_endoftext=False):
    if isinstance(encoder, ZSAE):
        act_name = act_name or "attn.hook_z"
        layer = encoder.cfg["layer"]  # the reference reads the layer from the cfg too (:203)
        name_filter = f"blocks.{layer}.attn.hook_z"
    elif isinstance(encoder, SparseTranscoder):
        act_name = act_name or encoder.cfg.hook_point
        layer = encoder.cfg.hook_point_layer
        name_filter = act_name
    else:
        raise ValueError("Unsupported encoder type")
    if isinstance(encoder, SparseTranscoder) and use_decoder and feature_post is None:
        # reference :228 takes W_dec[:, feature_indices] ([d_sae, n]) and multiplies [tokens, d_in] by it:
        # a shape error there as well unless d_in == d_sae
        raise RuntimeError("use_decoder=True needs an explicit feature_post [d_in, n_feat] "
                           "(W_dec[:, features] does not conform with the activations)")
    feats = torch.as_tensor(feature_indices, dtype=torch.long, device=encoder.W_enc.device).reshape(-1)
    eos = model.tokenizer.eos_token  # a *string*: the reference's comparison with int tokens (:244-247) never matches
    scores = []
    post_projector = None
    if use_raw_scores and feature_post is not None:
        post = feature_post.to(encoder.W_enc.device, torch.float32)
        if isinstance(encoder, SparseTranscoder) and use_decoder:
            bias = -(encoder.b_dec @ post)
        else:
            bias = encoder.b_enc[feats] - (encoder.b_dec @ post)
        post_projector = tsc_hip.FeatureProjector(post.t().contiguous(), bias.detach().contiguous(), None)
    for i in range(0, tokens_arr.shape[0], batch_size):
        cur_tokens = tokens_arr[i: i + batch_size]
        with torch.no_grad():
            _, cache = model.run_with_cache(cur_tokens, stop_at_layer=layer + 1, names_filter=name_filter)
            acts = cache[name_filter].reshape(-1, encoder.W_enc.shape[0]).contiguous()
            if use_raw_scores:
                if feature_post is None:
                    cur = encoder.encode_features(acts, feats, relu=False)  # :226-238 (pre-ReLU, bias folded)
                else:
                    cur = post_projector(acts, relu=False)
            else:
                cur = encoder.encode_features(acts, feats, relu=True)    # :240-241
            if ignore_endoftext:
                mask = torch.as_tensor([t == eos for t in cur_tokens.reshape(-1).tolist()], device=cur.device)
                cur[mask] = -torch.inf
        n_pos = tokens_arr.shape[1]
        # "(b pos) n -> b n pos", rounded to fp16 on the device (the reference's .astype(np.float16), :249-255,
        # is the same IEEE round-to-nearest-even); batches stay on the GPU: ONE transfer at the end of the scan
        # instead of the reference's device->host sync per batch
        scores.append(cur.reshape(-1, n_pos, cur.shape[-1]).permute(0, 2, 1).to(torch.float16))
    if post_projector is not None:
        post_projector.close()
    if not scores:
        return np.zeros((0, feats.numel(), tokens_arr.shape[1]), np.float16)
    return torch.cat(scores, dim=0).cpu().numpy()


def get_top_k_activating_examples(feature_scores, tokens, model, k=5):
    """``feature_scores`` is one feature's ``[n_seq, n_pos]`` slab (:260-274)."""
    flat_scores = feature_scores.flatten()
    top_k_indices = np.argsort(flat_scores, kind="stable")[-k:][::-1]
    top_k_batch_indices, top_k_seq_indices = np.unravel_index(top_k_indices, feature_scores.shape)
    top_k_tokens = [tokens[b].tolist() for b in top_k_batch_indices]
    top_k_tokens_str = [[model.to_string(x) for x in seq] for seq in top_k_tokens]
    top_k_scores_per_seq = [feature_scores[b].tolist() for b in top_k_batch_indices]
    return top_k_tokens_str, top_k_scores_per_seq, top_k_seq_indices
