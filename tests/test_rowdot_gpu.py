"""``tsc_rowdot_many`` (the auxiliary rows of the CircuitLens direction lens in one launch; reference
src/circuit_lens.py:591-604, 640-645 dotted as in :762-766) against fp64 torch: per-row and shared operands, the
optional subtrahend, strided outputs, more terms than one launch holds, odd widths."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("rows,d,n_terms", [(512, 768, 62), (7, 768, 5), (33, 64, 130), (3, 1028, 9), (1, 4, 1)])
def test_rowdot_many_matches_fp64(rows, d, n_terms):
    import tsc_hip

    g = torch.Generator(device=DEV).manual_seed(rows * 7 + d)
    out = torch.full((rows, n_terms + 3), -7.0, device=DEV)   # strided output views; the extra columns stay untouched
    terms, want = [], []
    for j in range(n_terms):
        a = torch.randn(rows, d, generator=g, device=DEV) if j % 3 else torch.randn(d, generator=g, device=DEV)
        b = torch.randn(rows, d, generator=g, device=DEV) if j % 4 else torch.randn(d, generator=g, device=DEV)
        c = None
        if j % 2:
            c = a[None].expand(rows, -1).clone() if a.dim() == 1 else a + 1e-3 * torch.randn(rows, d, generator=g, device=DEV)
            if j % 5 == 0:
                c = torch.randn(d, generator=g, device=DEV)   # a shared subtrahend
        terms.append((a, c, b, out[:, j + 1]))
        e = a.double() if c is None else a.double() - c.double()
        want.append((e * b.double()).sum(-1).expand(rows))
    tsc_hip.rowdot_many(terms, rows, d, torch.device(DEV))
    got = out[:, 1:n_terms + 1].double()
    ref = torch.stack(want, dim=1)
    scale = torch.stack([((a.double() if c is None else a.double() - c.double()).abs() * b.double().abs()).sum(-1).expand(rows)
                         for a, c, b, _ in terms], dim=1)
    assert float(((got - ref).abs() / (scale + 1e-30)).max()) < 2e-6   # fp32 sums of d terms in a fixed order
    assert bool((out[:, 0] == -7).all()) and bool((out[:, n_terms + 1:] == -7).all())
    # reproducible: the same call gives the same bits
    first = out.clone()
    tsc_hip.rowdot_many(terms, rows, d, torch.device(DEV))
    assert torch.equal(out, first)


def test_rowdot_many_argument_errors():
    import tsc_hip

    a = torch.zeros(4, 8, device=DEV)
    out = torch.zeros(4, device=DEV)
    with pytest.raises(tsc_hip.TscError):
        tsc_hip.rowdot_many([(a, None, torch.zeros(5, 8, device=DEV), out)], 4, 8, torch.device(DEV))   # wrong row count
    with pytest.raises(tsc_hip.TscError):
        tsc_hip.rowdot_many([(a.cpu(), None, a, out)], 4, 8, torch.device(DEV))
    with pytest.raises(tsc_hip.TscError):
        tsc_hip.rowdot_many([(a[:, :6], None, a[:, :6], out)], 4, 6, torch.device(DEV))                   # d % 4
    tsc_hip.rowdot_many([], 4, 8, torch.device(DEV))                                                      # nothing to do
