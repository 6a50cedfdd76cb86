"""``tsc_lens_assemble`` (normalise + rank tail of the CircuitLens pass, reference src/circuit_lens.py:551-554 and
:161-164) against a plain torch restatement: ``raw / raw.sum()`` over the live columns and a stable descending sort
(ties by column).  Covers rows that fit LDS and rows that do not, fewer live columns than k, exact ties, and the
many-candidates fallback of the selection."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _layout(n_head, group_aux, cap):
    cols, col = [], n_head
    for a in group_aux:
        cols.append((col, a))
        col += a + cap
    return cols, col


def _reference(head, aux, group_aux, scores, count, k):
    G, R, cap = scores.shape
    parts, masks, off = [head], [torch.ones_like(head, dtype=torch.bool)], 0
    slot = torch.arange(cap, device=scores.device)
    for g, a in enumerate(group_aux):
        parts.append(aux[:, off:off + a])
        masks.append(torch.ones(R, a, dtype=torch.bool, device=scores.device))
        off += a
        live = slot[None, :] < count[g].clamp(max=cap)[:, None]
        parts.append(torch.where(live, scores[g], torch.zeros_like(scores[g])))
        masks.append(live)
    raw, mask = torch.cat(parts, 1), torch.cat(masks, 1)
    acts = raw / raw.double().sum(1, keepdim=True).float()
    ranked = torch.where(mask, acts, torch.full_like(acts, float("-inf")))
    v, c = torch.sort(ranked, dim=1, descending=True, stable=True)
    v, c = v[:, :k], c[:, :k]
    c = torch.where(torch.isfinite(v), c, torch.full_like(c, -1))
    return raw, torch.where(mask, acts, torch.zeros_like(acts)), mask, v, c


def _check(head, aux, group_aux, scores, count, k, exact_values=False):
    import tsc_hip

    out = tsc_hip.lens_assemble(head, aux, group_aux, scores, count, k=k, dense=True)
    raw, acts, mask, v, c = _reference(head, aux, group_aux, scores, count, k)
    assert torch.equal(out["raw"], raw)
    assert torch.equal(out["mask"], mask)
    torch.testing.assert_close(out["activations"], acts, rtol=2e-5, atol=1e-9)  # row sums in another order
    assert torch.equal(out["top_columns"].long(), c), int((out["top_columns"].long() != c).sum())
    fin = torch.isfinite(v)
    assert torch.equal(torch.isfinite(out["top_values"]), fin)
    torch.testing.assert_close(out["top_values"][fin], v[fin], rtol=2e-5, atol=1e-9)
    assert bool((out["top_info"][:, :, 1:] == -1).all())  # no tags, no feature ids given
    if out["activations"].shape[1] <= 15360:  # the lean form ranks out of LDS alone
        lean = tsc_hip.lens_assemble(head, aux, group_aux, scores, count, k=k, dense=False)
        assert torch.equal(lean["top_columns"], out["top_columns"]) and torch.equal(lean["top_values"], out["top_values"])
        assert "raw" not in lean


@pytest.mark.parametrize("R,G,cap,k", [(37, 24, 512, 10), (5, 3, 64, 64), (9, 24, 2048, 10), (3, 1, 7, 4)])
def test_lens_assemble_matches_torch(R, G, cap, k):
    g = torch.Generator(device=DEV).manual_seed(R * 1000 + cap)
    group_aux = [3, 2] * (G // 2) + [1] * (G % 2)
    scores = torch.randn(G, R, cap, generator=g, device=DEV)
    count = torch.randint(0, cap + 1, (G, R), generator=g, device=DEV, dtype=torch.int32)
    count[0, 0] = cap + 5  # an overflowed count is clamped to the capacity
    head = torch.randn(R, 2, generator=g, device=DEV)
    aux = torch.randn(R, sum(group_aux), generator=g, device=DEV)
    head[:, 0] += 40.0  # keep the row sums away from zero
    _check(head, aux, group_aux, scores, count, k)


def test_lens_assemble_ties_and_short_rows():
    G, R, cap, k = 4, 6, 32, 10
    group_aux = [0, 1, 0, 2]
    scores = torch.ones(G, R, cap, device=DEV)  # every live column equal: the first k columns win, in column order
    count = torch.full((G, R), 5, dtype=torch.int32, device=DEV)
    count[:, 1] = 0
    count[:, 2] = 1
    head = torch.zeros(R, 0, device=DEV)
    aux = torch.ones(R, 3, device=DEV)
    aux[3] = 7.0
    _check(head, aux, group_aux, scores, count, k)
    import tsc_hip

    out = tsc_hip.lens_assemble(head, aux, group_aux, scores, count, k=k)
    assert out["top_columns"][1].tolist() == [32, 97, 98] + [-1] * 7  # row 1: only its three auxiliary columns live


def test_lens_assemble_many_candidates_take_the_round_fallback():
    """64 threads' columns hold the 64 largest per-thread maxima and ALL their elements reach the threshold: 12 000
    candidates, more than the LDS list holds, so the k serial rounds run (wide row: ranked out of the dense outputs)."""
    G, R, cap, k = 24, 3, 2048, 64
    group_aux = [3, 2] * 12
    cols, W = _layout(2, group_aux, cap)
    c = torch.arange(W, device=DEV)
    row = torch.where((c & 255) < 64, 100.0 + (c & 255).float(), torch.rand(W, device=DEV))
    raw = row[None, :].repeat(R, 1)
    raw[1] = raw[1].flip(0)
    head = raw[:, :2].contiguous()
    aux = torch.cat([raw[:, c0:c0 + a] for c0, a in cols], 1).contiguous()
    scores = torch.stack([raw[:, c0 + a:c0 + a + cap] for c0, a in cols]).contiguous()
    count = torch.full((G, R), cap, dtype=torch.int32, device=DEV)
    _check(head, aux, group_aux, scores, count, k)


def test_lens_assemble_labels_and_feature_ids():
    """``column_tags`` and ``feat``: the winner's class label (split in two halves) and the feature behind a slot."""
    import tsc_hip

    G, R, cap, k = 4, 8, 16, 6
    group_aux = [3, 2, 3, 2]
    g = torch.Generator(device=DEV).manual_seed(5)
    scores = torch.randn(G, R, cap, generator=g, device=DEV)
    count = torch.randint(1, cap + 1, (G, R), generator=g, device=DEV, dtype=torch.int32)
    feat = torch.randint(0, 24576, (G, R, cap), generator=g, device=DEV, dtype=torch.int32)
    head = torch.randn(R, 2, generator=g, device=DEV) + 30
    aux = torch.randn(R, 10, generator=g, device=DEV)
    tags = torch.arange(2 + 10 + G, dtype=torch.int32, device=DEV) * 65537 + 65536  # hi = i + 1, lo = i
    out = tsc_hip.lens_assemble(head, aux, group_aux, scores, count, k=k, column_tags=tags, feat=feat)
    cols, W = _layout(2, group_aux, cap)
    info = out["top_info"].cpu()
    aux_off = [0, 3, 5, 8]
    for r in range(R):
        for j in range(k):
            c, hi, lo, f = info[r, j].tolist()
            if c < 2:
                cls, want_f = c, -1
            else:
                gi = max(i for i, (c0, _) in enumerate(cols) if c >= c0)
                o = c - cols[gi][0]
                if o < group_aux[gi]:
                    cls, want_f = 2 + aux_off[gi] + o, -1
                else:
                    cls, want_f = 2 + 10 + gi, int(feat[gi, r, o - group_aux[gi]])
            assert (hi, lo, f) == (cls + 1, cls, want_f), (r, j, c)


def test_lens_assemble_argument_errors():
    import tsc_hip

    s = torch.zeros(2, 4, 8, device=DEV)
    cnt = torch.zeros(2, 4, dtype=torch.int32, device=DEV)
    with pytest.raises(tsc_hip.TscError):
        tsc_hip.lens_assemble(torch.zeros(4, 2, device=DEV), torch.zeros(4, 3, device=DEV), [1, 1], s, cnt)
    with pytest.raises(tsc_hip.TscError):
        tsc_hip.lens_assemble(torch.zeros(4, 2), torch.zeros(4, 2, device=DEV), [1, 1], s, cnt)
    with pytest.raises(tsc_hip.TscError):
        tsc_hip.lens_assemble(torch.zeros(4, 2, device=DEV), torch.zeros(4, 130, device=DEV), [1] * 65,
                              torch.zeros(65, 4, 8, device=DEV), torch.zeros(65, 4, dtype=torch.int32, device=DEV))


def test_lens_assemble_zero_rows():
    import tsc_hip

    out = tsc_hip.lens_assemble(torch.zeros(0, 2, device=DEV), torch.zeros(0, 2, device=DEV), [1, 1],
                                torch.zeros(2, 0, 8, device=DEV), torch.zeros(2, 0, dtype=torch.int32, device=DEV), k=3)
    assert out["top_values"].shape == (0, 3) and out["top_info"].shape == (0, 3, 4) and out["activations"].shape == (0, 20)


@pytest.mark.parametrize("cap", [64, 2048])  # a row that is ranked out of LDS, and one that is too wide for it
def test_lens_assemble_live_minus_infinity_is_live_on_both_paths(cap):
    """Round-2 advisor: liveness must not be carried by a value.  A LIVE score of -inf (a masked score) stays live in
    ``mask`` and is divided like any other value, whether the row goes through LDS or not; dead slots read 0."""
    import tsc_hip

    G, R, k = 24, 4, 5
    group_aux = [3, 2] * 12
    g = torch.Generator(device=DEV).manual_seed(cap)
    scores = torch.rand(G, R, cap, generator=g, device=DEV)
    count = torch.full((G, R), cap // 2, dtype=torch.int32, device=DEV)
    scores[3, 1, 2] = float("-inf")          # live (slot 2 < count), row 1
    scores[5, 2, cap - 1] = float("-inf")    # dead slot, row 2: must not matter
    head = torch.rand(R, 2, generator=g, device=DEV) + 10
    aux = torch.rand(R, sum(group_aux), generator=g, device=DEV)
    out = tsc_hip.lens_assemble(head, aux, group_aux, scores, count, k=k, dense=True)
    raw, acts, mask, v, c = _reference(head, aux, group_aux, scores, count, k)
    assert torch.equal(out["mask"], mask) and torch.equal(out["raw"], raw)
    cols, W = _layout(2, group_aux, cap)
    col = cols[3][0] + group_aux[3] + 2
    assert bool(out["mask"][1, col]) and float(out["raw"][1, col]) == float("-inf")
    assert float(out["activations"][1, col]) != 0.0          # -inf / -inf = NaN: divided, not zeroed as a dead slot is
    for r in (0, 2, 3):                                      # the other rows are ordinary
        torch.testing.assert_close(out["activations"][r], acts[r], rtol=2e-5, atol=1e-9)
        assert torch.equal(out["top_columns"][r].long(), c[r])
