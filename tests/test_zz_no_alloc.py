"""The allocation-free entry point ``tsc_encode_topk_static`` (include/tsc_hip.h) against the existing
``tsc_encode_topk``, through the C-ABI: bit-identical finalized lists on two calls in a row, the second with a
smaller input (the static scratch keeps the first call's contents), and no device allocation inside the call."""
import ctypes

import pytest
import torch

import golden_inputs as gi

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dict(w):
    import tsc_hip

    g = {k: v.to(DEV) for k, v in w.items()}
    return tsc_hip.DeviceDict(g["W_enc"], g["b_enc"], g["b_dec"]), g


def test_static_scratch_entry_point_matches_encode_topk():
    import tsc_hip
    from standins.synthetic_dicts import make_dictionary_weights

    g, w, xs, n_seq, n_pos = gi.g2_inputs()            # d_sae 1536
    wide = make_dictionary_weights(768, 24576, seed=9, b_enc=-0.7)  # a second state of another width on the same scratch
    d_small, _ = _dict(w)
    d_wide, _ = _dict(wide)
    x1 = xs[:3000].to(DEV)
    x2 = torch.cat([xs[3000:3700], xs[:100]]).to(DEV)  # smaller second input, with exact ties against the first
    xw = torch.randn(2500, 768, generator=torch.Generator().manual_seed(3)).to(DEV)

    ref = tsc_hip.TopkState(1536, 20, DEV, n_slots=4)                       # existing entry point, caller-owned floor
    new = tsc_hip.TopkState(1536, 20, DEV, n_slots=4, cross_slot_floor=False)  # no floor of its own at all
    ref_w = tsc_hip.TopkState(24576, 20, DEV)
    new_w = tsc_hip.TopkState(24576, 20, DEV, cross_slot_floor=False)
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
    # call 1 (large), a call of another width in between (overwrites the shared scratch), call 2 (smaller)
    d_small.encode_topk_static(x1, 0, new)
    d_wide.encode_topk_static(xw, 50, new_w)
    d_small.encode_topk_static(x2, 3000, new)
    torch.cuda.synchronize()
    after = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
    assert after[0] == before[0], "torch allocated device memory during the allocation-free calls"
    assert after[1] >= before[1], "free device memory shrank during the allocation-free calls (a hipMalloc happened)"
    d_small.encode_topk(x1, 0, ref)
    d_wide.encode_topk(xw, 50, ref_w)
    d_small.encode_topk(x2, 3000, ref)
    for a, b in ((new, ref), (new_w, ref_w)):
        va, ia = a.finalize()
        vb, ib = b.finalize()
        assert torch.equal(va, vb) and torch.equal(ia, ib)
    # and against the exact ranking of the library's own dense activations
    from test_hip_kernels import _rank_own_dense

    dense = d_small.encode_dense(torch.cat([x1, x2]))
    ev, ei = _rank_own_dense(dense, 20)
    v, i = new.finalize()
    assert torch.equal(v.cpu(), ev) and torch.equal(i.cpu(), ei)


def test_static_scratch_entry_point_rejects_oversized_states():
    import tsc_hip

    g, w, xs, n_seq, n_pos = gi.g2_inputs()
    d, _ = _dict(w)
    lib = tsc_hip.lib()
    too_wide = tsc_hip.STATIC_FLOOR_MAX_D_SAE + 4
    st = tsc_hip._TopkStateStruct(16, 16, too_wide, 20, 1, 0, None)  # never dereferenced: refused on its size
    rc = lib.tsc_encode_topk_static(d._handle, xs[:8].to(DEV).data_ptr(), 8, 0, ctypes.byref(st), None)
    assert rc == -3 and b"static scratch" in lib.tsc_last_error()
