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


class GradPool(object):
    """fp64-anchored gradient bound, pooled over cases (seeds) and kept per scale.

    For every case add(s, got, ref32, g64): ``got`` is the HIP gradient, ``ref32`` an fp32 run of the reference
    algorithm (the oracle in fp32, or a golden of the reference itself), ``g64`` the oracle in float64.  check() asserts,
    per scale and over ALL elements of all pooled cases,

        rel-L2(got vs g64)  <= 1.5 * rel-L2(ref32 vs g64) + 1e-6
        #bad(got)           <= 1.5 * #bad(ref32) + 1e-3 * numel        (bad: |err| > 1e-4 * (max|g64| + |g64|))

    i.e. HIP may not be further from exact arithmetic than 1.5x the reference's own fp32 conditioning noise.  No
    additive slack beyond 1e-6 / 1e-3: a gradient that is a few per cent wrong fails."""

    def __init__(self, scales=4):
        z = lambda: np.zeros(scales)   # noqa: E731
        self.num_h, self.num_r, self.den, self.bad_h, self.bad_r, self.cnt = z(), z(), z(), z(), z(), z()

    def add(self, s, got, ref32, g64):
        got, ref32, g64 = (t.detach().double().cpu() for t in (got, ref32, g64))
        assert got.shape == g64.shape == ref32.shape, (got.shape, ref32.shape, g64.shape)
        assert torch.isfinite(got).all()
        self.num_h[s] += float((got - g64).pow(2).sum())
        self.num_r[s] += float((ref32 - g64).pow(2).sum())
        self.den[s] += float(g64.pow(2).sum())
        tol = 1e-4 * g64.abs().max().item() + 1e-4 * g64.abs()
        self.bad_h[s] += float(((got - g64).abs() > tol).sum())
        self.bad_r[s] += float(((ref32 - g64).abs() > tol).sum())
        self.cnt[s] += g64.numel()

    def check(self, name=""):
        for s in range(len(self.cnt)):
            if self.cnt[s] == 0:
                continue
            e_h, e_r = (self.num_h[s] / self.den[s]) ** 0.5, (self.num_r[s] / self.den[s]) ** 0.5
            print("%s scale %d: rel-L2 vs fp64  hip %.3g  ref32 %.3g | bad elements hip %d ref32 %d of %d" % (
                name, s, e_h, e_r, self.bad_h[s], self.bad_r[s], self.cnt[s]))
            assert e_h <= 1.5 * e_r + 1e-6, "%s scale %d: HIP rel-L2 %.3g vs fp32 reference %.3g (both against fp64)" % (
                name, s, e_h, e_r)
            assert self.bad_h[s] <= 1.5 * self.bad_r[s] + 1e-3 * self.cnt[s], (name, s, self.bad_h[s], self.bad_r[s],
                                                                             self.cnt[s])
