"""CPU tests of the host-side logic: ActiveFeatures index algebra against the reference's golden
rankings, row sharding, and the world_size-2 token-sharded sweep over ``gloo`` (the oracle is injected
as the numerical engine HERE ONLY, to exercise the multi-process plumbing without a GPU)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import golden_inputs as gi
from oracle import lens_ref as L


def _active_features_from_oracle(pos):
    from circuit_lens import ActiveFeatures

    g3, g4, model, z_w, t_w = gi.lens_inputs()
    tokens = model.to_tokens(gi.PROMPT)
    _, cache = model.run_with_cache(tokens)
    vectors, values, features, keys = L.active_features_ref(model, cache, z_w, t_w, pos)
    return ActiveFeatures(vectors=vectors, values=values, features=features, keys=keys), g3, g4, model, tokens


def test_active_features_index_algebra_matches_reference_rankings():
    af, g3, g4, model, tokens = _active_features_from_oracle(12)
    keys = af.keys
    n_layers = len(keys)
    assert af.get_total_auxilary_features() == 2 + 5 * n_layers
    assert af.get_total_active_features() == sum(k["attn"] + k["mlp"] for k in keys)
    assert af.vectors.shape[0] == af.get_total_auxilary_features() + af.get_total_active_features()
    assert af.get_attn_start_index(0) == 2
    assert af.get_mlp_start_index(0) == 2 + 3 + keys[0]["attn"]
    assert af.get_attn_start_index(1) == 2 + 5 + keys[0]["attn"] + keys[0]["mlp"]
    assert af.get_vectors_before_comp("attn", n_layers).shape[0] == af.vectors.shape[0]
    assert af.get_attn_feature_vectors(1).shape[0] == keys[1]["attn"]
    assert af.get_mlp_feature_vectors(2).shape[0] == keys[2]["mlp"]
    assert af.max_active_features == max(max(k["attn"], k["mlp"]) for k in keys)
    # features are -1 exactly on auxiliary rows
    for i in range(af.vectors.shape[0]):
        kind, layer = af.classify_row(i)
        assert (int(af.features[i]) >= 0) == (kind in ("z_feature", "mlp_feature"))
    # rankings produced by the reference's get_top_k_features (incl. its stop at embed/pos_embed)
    tok = int(tokens[0, 13])
    raw = L.unembed_lens_raw_ref(af.vectors, model.W_U, tok)
    top = af.get_top_k_features(L.normalize_ref(raw), k=10)
    assert [[gi.COMPONENT_IDS.index(c), l, f] for c, l, f, _ in top] == g4["unembed_top_ids"].tolist()
    assert np.allclose([v for *_, v in top], g4["unembed_top_vals"], atol=1e-5)
    masked = raw.clone()
    masked[:2] = -1e9
    top = af.get_top_k_features(masked, k=40)
    assert [[gi.COMPONENT_IDS.index(c), l, f] for c, l, f, _ in top] == g4["topk_masked_ids"].tolist()
    labels = af.get_top_k_labels(masked, k=3)
    assert len(labels) == 3 and "Layer" in labels[0]


def test_component_lens_tuple_ids_round_trip():
    from circuit_lens import CircuitComponent, ComponentLens

    for tid in [(CircuitComponent.UNEMBED, 3, 17), (CircuitComponent.UNEMBED_AT_TOKEN, 4),
                (CircuitComponent.ATTN_HEAD, 2, 5, 1, 9, 300), (CircuitComponent.MLP_FEATURE, 1, 7, 42),
                (CircuitComponent.Z_SAE_ERROR, 0, 2, -1)]:
        lens = ComponentLens.init_from_tuple_id(tid, circuit_lens=None)
        assert lens.tuple_id == tid
    assert ComponentLens(None, CircuitComponent.EMBED, {"layer": 0, "seq_index": 1, "feature": -1})() == ([], None)
    assert ComponentLens(None, CircuitComponent.UNEMBED, {"seq_index": 1, "token_id": 2}).feature == -1


def test_shard_rows_partitions_exactly():
    from topk_scan import shard_rows, unravel_token_ids

    for n, world in ((10, 3), (8, 8), (5, 8), (25600, 8), (0, 2)):
        spans = [shard_rows(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_rows(4, 4, 4)
    seq, pos = unravel_token_ids(torch.tensor([[0, 129, -1]]), 128)
    assert seq.tolist() == [[0, 1, -1]] and pos.tolist() == [[0, 1, -1]]


class _OracleEngine:
    """Test-only engine: the CPU oracle behind the product's engine interface."""

    def __init__(self, w, k):
        from oracle.topk_ref import empty_state

        self.w, self.k = w, k
        self.state = empty_state(w["W_enc"].shape[1], k)

    def update(self, x, base):
        from oracle.topk_ref import topk_stream_ref

        self.state = topk_stream_ref(x, self.w, k=self.k, token_id_base=base, state=self.state)

    def finalize(self):
        return self.state

    def load(self, vals, idx):
        self.state = (vals.clone(), idx.clone())

    @staticmethod
    def merge(vals, idx):
        from oracle.topk_ref import topk_merge_ref

        return topk_merge_ref([(vals[i], idx[i]) for i in range(vals.shape[0])], k=vals.shape[-1])


def test_sweep_checkpoint_and_resume(tmp_path):
    """A sweep saved half way and resumed in a new object ends with the lists of an uninterrupted one;
    the file holds plain tensors only (loadable with weights_only=True)."""
    import topk_scan
    from oracle.topk_ref import topk_stream_ref
    from standins.synthetic_dicts import make_dictionary_weights

    ws = {n: make_dictionary_weights(64, 96, seed=i, b_enc=-0.3) for i, n in enumerate(("z_0", "mlp_0"))}
    x = torch.randn(600, 64, generator=torch.Generator().manual_seed(0))
    mk = lambda: topk_scan.MaxActSweep(ws, k=5, engine_factory=lambda w: _OracleEngine(w, 5), distributed=False)
    a = mk()
    a.feed({n: x[:250] for n in ws}, 0)
    a.save(str(tmp_path / "sweep.pt"))
    sd = torch.load(str(tmp_path / "sweep.pt"), weights_only=True)
    assert int(sd["tokens_seen"]) == 250 and set(sd) == {"tokens_seen", "k", "z_0.vals", "z_0.idx", "mlp_0.vals", "mlp_0.idx"}
    b = mk()
    b.load(str(tmp_path / "sweep.pt"))
    b.feed({n: x[250:] for n in ws}, b.tokens_seen)
    res = b.result()
    for n, w in ws.items():
        ev, ei = topk_stream_ref(x, w, k=5)
        assert torch.equal(res[n][0], ev) and torch.equal(res[n][1], ei)
    with pytest.raises(ValueError):
        topk_scan.MaxActSweep(ws, k=6, engine_factory=lambda w: _OracleEngine(w, 6), distributed=False).load(
            str(tmp_path / "sweep.pt"))


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from replay_model import ReplayModel
    from topk_scan import sweep_model

    g, w, xs, n_seq, n_pos = gi.g2_inputs()
    model = ReplayModel(xs, n_seq, n_pos)
    dummy = type("D", (), {"w": w, "cfg": type("C", (), {"hook_point": "blocks.0.ln2.hook_normalized"})()})()
    res = sweep_model(model, model.token_matrix(), transcoders=[dummy], layers=[0], k=20, batch_size=5,
                      rank=rank, world=world, engine_factory=lambda d: _OracleEngine(d.w, 20))
    vals, idx = res["mlp_0"]
    torch.save((vals, idx), os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_token_sharded_sweep_two_ranks_gloo(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g = gi.load("g2_topk.npz")
    outs = [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]
    for vals, idx in outs:  # every rank ends with the identical, complete answer
        assert np.array_equal(idx.numpy(), g["idx"].astype(np.int64))
        assert np.allclose(vals.numpy(), g["vals"], rtol=1e-4, atol=1e-5)


def test_topk_engine_feeds_young_lists_in_pieces_no_longer_than_their_history():
    """``HipTopkEngine.update`` (host logic only; a recording stand-in for the dictionary): the first 16384 rows of a sweep
    go to the fp32 kernel, then -- while fewer than 131072 tokens have been seen -- a batch is fed to the screened top-k
    in pieces no longer than what the lists have already seen (its floor is the k-th best BEFORE a call), then whole;
    pieces tile the batch exactly, with consecutive token ids; an unscreened or bf16 engine passes batches through."""
    import topk_scan

    class Recorder:
        def __init__(self):
            self.calls = []

        def new_topk_state(self, k):
            return object()

        def encode_topk(self, x, base, state, bf16=False, screened=False):
            self.calls.append((x.shape[0], base, bf16, screened))

    rec = Recorder()
    eng = topk_scan.HipTopkEngine(rec, k=20)
    x = torch.zeros(65536, 4)
    eng.update(x, 1000)
    assert rec.calls == [(16384, 1000, False, False), (16384, 1000 + 16384, False, True), (32768, 1000 + 32768, False, True)]
    rec.calls.clear()
    eng.update(x, 1000 + 65536)                      # 65536 seen: one piece as long as the history
    eng.update(x[:50000], 1000 + 131072)             # 131072 seen: no longer young
    assert rec.calls == [(65536, 66536, False, True), (50000, 132072, False, True)]
    rec.calls.clear()
    small = topk_scan.HipTopkEngine(rec, k=20)       # batches shorter than the fp32 head
    for i in range(5):
        small.update(x[:6000], 6000 * i)
    assert [c[0] for c in rec.calls] == [6000, 6000, 4384, 1616, 6000, 6000]
    assert [c[3] for c in rec.calls] == [False, False, False, True, True, True]
    assert [c[1] for c in rec.calls] == [0, 6000, 12000, 16384, 18000, 24000]
    rec.calls.clear()
    for kwargs in ({"screened": False}, {"bf16": True}):
        plain = topk_scan.HipTopkEngine(rec, k=20, **kwargs)
        plain.update(x, 7)
        assert rec.calls[-1] == (65536, 7, bool(kwargs.get("bf16")), False) and plain.tokens_seen == 65536


def test_residual_space_z_folds_w_o_once_and_refolds_on_change():
    """``circuit_lens._ResidualSpaceZ`` (host side, CPU tensors): ``W_dec W_O`` / ``b_dec W_O`` are folded once, reused while
    nothing changed, and redone after an in-place update of the z-SAE's decoder or of W_O; ``same_projection`` accepts the
    same storage or equal values (a model whose ``W_O`` property stacks a new tensor on every access) and nothing else."""
    import circuit_lens

    class Z(torch.nn.Module):
        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(0)
            self.W_enc = torch.nn.Parameter(torch.randn(12, 40, generator=g))
            self.b_enc = torch.nn.Parameter(torch.randn(40, generator=g))
            self.b_dec = torch.nn.Parameter(torch.randn(12, generator=g))
            self.W_dec = torch.nn.Parameter(torch.randn(40, 12, generator=g))

    z = Z()
    w_o = torch.randn(12, 8, generator=torch.Generator().manual_seed(1))
    view = circuit_lens._residual_z(z, w_o)
    assert circuit_lens._residual_z(z, w_o) is view and circuit_lens._residual_z(z, w_o.clone()) is view
    assert circuit_lens._residual_z(z, w_o * 2) is not view
    view = circuit_lens._residual_z(z, w_o)
    first = view.W_dec
    assert view.W_dec is first and view.b_dec is z.b_dec
    assert torch.allclose(first, z.W_dec.detach() @ w_o, atol=1e-6) and torch.allclose(view.b_out, z.b_dec.detach() @ w_o, atol=1e-6)
    with torch.no_grad():
        z.W_dec.mul_(2.0)
    assert view.W_dec is not first and torch.allclose(view.W_dec, 2.0 * first, atol=1e-5)
    second = view.W_dec
    w_o.add_(1.0)
    assert view.W_dec is not second and torch.allclose(view.W_dec, z.W_dec.detach() @ w_o, atol=1e-5)
    import copy
    assert "_resid_adapter" in z.__dict__ and isinstance(copy.deepcopy(z), Z)
