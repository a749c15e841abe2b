import numpy as np
import torch


def to_dev(x, dev="cuda"):
    if torch.is_tensor(x):
        return x.to(dev).contiguous()
    if isinstance(x, dict):
        return {k: to_dev(v, dev) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(to_dev(v, dev) for v in x)
    return x


def rel_l2(got, want):
    """||got - want|| / ||want|| over ALL elements, in float64."""
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    d = want.norm().item()
    return (got - want).norm().item() / d if d > 0 else (got - want).norm().item()


def bad_frac(got, want, rtol, atol):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return ((got - want).abs() > (atol + rtol * want.abs())).double().mean().item()


def assert_close_frac(got, want, rtol=1e-4, atol=1e-6, max_bad_frac=0.0, name="", max_rel_l2=None):
    """|got-want| <= atol + rtol*|want| for all but ``max_bad_frac`` of the elements, and the
    relative L2 error is below 10*rtol.  The outlier allowance exists for quantities whose
    derivative is discontinuous in the inputs (bilinear floor(), min/argmin ties)."""
    got = got.detach().double().cpu()
    want = want.detach().double().cpu()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    assert torch.isfinite(got).all(), name + ": non-finite values"
    err = (got - want).abs()
    bad = err > (atol + rtol * want.abs())
    frac = bad.double().mean().item()
    denom = want.norm().item()
    rel_l2 = (got - want).norm().item() / denom if denom > 0 else (got - want).norm().item()
    assert frac <= max_bad_frac, "%s: %.3g of elements out of tolerance (max err %.3g, rel-L2 %.3g)" % (
        name, frac, err.max().item(), rel_l2)
    if max_rel_l2 is not None:     # untrimmed: every element counts, outliers included
        assert rel_l2 <= max_rel_l2, "%s: rel-L2 over all elements %.3g > %.3g" % (name, rel_l2, max_rel_l2)
    # rel-L2 over the elements that are in tolerance (all of them when no outliers are allowed)
    good = ~bad
    dg = want[good].norm().item()
    trimmed = (got[good] - want[good]).norm().item() / dg if dg > 0 else 0.0
    assert trimmed <= 10 * rtol or err[good].max().item() <= atol, "%s: trimmed rel-L2 %.3g" % (name, trimmed)
    return frac, rel_l2


def np_t(a):
    return torch.from_numpy(np.asarray(a))
