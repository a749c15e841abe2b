"""An independent check of oracle/tv082.py (the torchvision-0.8.2 restatement every attack / prep_adv_data golden and K3
inherit) that does NOT go through the builder's own arithmetic:

* ``perspective`` against Pillow.  torchvision's tensor ``perspective`` was written to reproduce
  ``PIL.Image.transform(size, PERSPECTIVE, coeffs, BILINEAR)``: the same 8-coefficient convention
  (``_get_perspective_coeffs`` serves both back ends), output pixel centres at +0.5, bilinear taps at the input
  position - 0.5.  Pillow evaluates it in C with double arithmetic on mode-"F" images, shares no code with
  ``F.grid_sample``, and is installed here.  (The two differ only in the outermost half pixel of the INPUT image --
  Pillow clamps, grid_sample pads with zeros -- which the reference's zero ``Pad`` makes moot: physicalTrans.py:114-116.)
* identity / integer-translation / pure-scale quads against closed forms.
* ``resize`` against a two-tap lerp written out index by index (Pillow's own resize antialiases, so it is no reference
  for the 0.8.2 tensor path).

This does not pin torchvision 0.8.2 itself (not installed, not vendored: "parity unpinned" stays in the headers), but it
removes "the restatement checked against itself".  Reference call sites: physicalTrans.py:107-166,
torchattacks/attacks/phy_obj_atk.py:89-90.
"""
import numpy as np
import pytest
import torch
from PIL import Image

from oracle import tv082
from oracle.synth import kitti_like, make_object

SH, SW = 375, 1242          # the attack scenes' frame


def _padded_object():
    obj, mask = make_object()
    # image-like texture (a box-blurred field) on top of the seeded noise patch, so that both smooth and rough texels occur
    obj = (0.5 * obj + 0.5 * kitti_like(1, 3, obj.shape[2], obj.shape[3], torch.Generator().manual_seed(3))).contiguous()
    h, w = obj.shape[-2:]
    l_pad, t_pad = (SW - w) // 2, (SH - h) // 2
    padding = [l_pad, t_pad, SW - w - l_pad, SH - h - t_pad]
    start = [[l_pad, t_pad], [l_pad + w, t_pad], [l_pad + w, t_pad + h], [l_pad, t_pad + h]]
    return tv082.pad(obj, padding), tv082.pad(mask, padding), start, (l_pad, t_pad, h, w)


def _pil_perspective(plane, coeffs):
    """One [H,W] float plane through Pillow's C implementation."""
    im = Image.fromarray(np.ascontiguousarray(plane, dtype=np.float32), mode="F")
    out = im.transform((plane.shape[1], plane.shape[0]), Image.PERSPECTIVE, [float(c) for c in coeffs], Image.BILINEAR)
    return np.asarray(out, dtype=np.float32)


def test_pad_is_a_zero_frame_around_the_object():
    img, msk, start, (l_pad, t_pad, h, w) = _padded_object()
    assert tuple(img.shape) == (1, 3, SH, SW) and tuple(msk.shape) == (1, 1, SH, SW)
    ref = np.zeros((3, SH, SW), np.float32)
    ref[:, t_pad:t_pad + h, l_pad:l_pad + w] = img[0, :, t_pad:t_pad + h, l_pad:l_pad + w].numpy()
    assert np.array_equal(img[0].numpy(), ref)
    assert float(msk.sum()) == float(msk[0, 0, t_pad:t_pad + h, l_pad:l_pad + w].sum()) > 0


def test_perspective_matches_pillow_on_all_pose_quads(golden):
    """All 25 x 13 (z0, alpha) quads of PhysicalTrans.objPosOnImage (the reference's integer quads, tests/golden/
    geometry.npz): the restatement in float64 is Pillow's result to float32 rounding; in float32 (what the goldens
    were generated with) it stays within the coordinate-rounding error of a ~1000-pixel-wide normalised grid."""
    g = golden("geometry")
    img, msk, start, _ = _padded_object()
    quads = g["quads"].reshape(-1, 4, 2)
    assert np.array_equal(g["start"], np.asarray(start))
    worst64 = worst32 = 0.0
    for i, quad in enumerate(quads):
        end = [[int(p[0]), int(p[1])] for p in quad]
        coeffs = tv082.get_perspective_coeffs([list(map(float, p)) for p in start], [list(map(float, p)) for p in end])
        # the mask for every pose, the three colour planes for every fourth one (Pillow takes ~10 ms per plane)
        planes = [(msk, 0)] + ([(img, c) for c in range(3)] if i % 4 == 0 else [])
        for src, c in planes:
            pil = _pil_perspective(src[0, c].numpy(), coeffs)
            got64 = tv082.perspective_coeffs(src[:, c:c + 1].double(), coeffs)[0, 0].numpy()
            got32 = tv082.perspective_coeffs(src[:, c:c + 1], coeffs)[0, 0].numpy()
            e64 = np.abs(got64 - pil).max()
            worst64 = max(worst64, e64)
            assert e64 <= 2e-6, ("fp64 restatement vs Pillow", i, c, e64)
            e32 = np.abs(got32 - pil)
            worst32 = max(worst32, e32.max())
            # fp32: the sample coordinate carries ~1e-4 px of rounding at x ~ 1000; a binary mask edge or a noise texel
            # turns that into <= ~1e-3 of value, on a vanishing share of the pixels
            assert e32.max() <= 2e-3 and (e32 > 1e-4).mean() <= 2e-3, ("fp32 restatement vs Pillow", i, c, e32.max())
            assert pil.max() > 0.5, "the object must be visible in the frame"
    print("perspective vs Pillow over %d quads: worst |diff| fp64 %.3g, fp32 %.3g" % (len(quads), worst64, worst32))


@pytest.mark.parametrize("dx,dy", [(0, 0), (7, 0), (0, -5), (-31, 12)])
def test_identity_and_integer_translation_quads_are_exact(dx, dy):
    img, msk, start, _ = _padded_object()
    end = [[p[0] + dx, p[1] + dy] for p in start]
    out = tv082.perspective(img.double(), start, end)
    want = torch.zeros_like(out)
    ys, yd = (slice(0, SH - dy), slice(dy, SH)) if dy >= 0 else (slice(-dy, SH), slice(0, SH + dy))
    xs, xd = (slice(0, SW - dx), slice(dx, SW)) if dx >= 0 else (slice(-dx, SW), slice(0, SW + dx))
    want[:, :, yd, xd] = img.double()[:, :, ys, xs]
    assert float((out - want).abs().max()) <= 1e-12
    out32 = tv082.perspective(img, start, end)
    assert float((out32.double() - want).abs().max()) <= 2e-4          # fp32 grid rounding only
    pil = _pil_perspective(img[0, 1].numpy(), tv082.get_perspective_coeffs(
        [list(map(float, p)) for p in start], [list(map(float, p)) for p in end]))
    assert np.abs(pil - want[0, 1].numpy()).max() <= 1e-6


def test_pure_scale_quad_is_the_closed_form_two_tap_lerp():
    """Magnify by 2 about the origin: output pixel x samples the input at (x + 0.5)/2 - 0.5, i.e. taps (x-1)/2, (x+1)/2
    with weights 1/4, 3/4 for odd x and x/2 - 1, x/2 with 3/4... written out below; zeros beyond the image."""
    g = torch.Generator().manual_seed(4)
    src = torch.rand(1, 1, 40, 64, generator=g, dtype=torch.float64)
    H, W = src.shape[-2:]
    start = [[0, 0], [W, 0], [W, H], [0, H]]
    end = [[0, 0], [2 * W, 0], [2 * W, 2 * H], [0, 2 * H]]
    out = tv082.perspective(src, start, end)[0, 0].numpy()
    a = np.pad(src[0, 0].numpy(), 1)          # zeros padding (grid_sample padding_mode="zeros")

    def taps(n):
        pos = (np.arange(n) + 0.5) / 2 - 0.5
        i0 = np.floor(pos).astype(int)
        return i0 + 1, pos - i0               # +1: index into the padded array
    iy, fy = taps(H)
    ix, fx = taps(W)
    want = ((1 - fy)[:, None] * ((1 - fx) * a[iy][:, ix] + fx * a[iy][:, ix + 1]) +
            fy[:, None] * ((1 - fx) * a[iy + 1][:, ix] + fx * a[iy + 1][:, ix + 1]))
    assert np.abs(out - want).max() <= 1e-12
    assert set(np.round(fx, 6)) == {0.25, 0.75}


def test_resize_is_the_non_antialiased_two_tap_lerp():
    """transforms.Resize on a tensor in 0.8.2 (phy_obj_atk.py:89-90: 375x1242 -> 320x1024): source index
    (i + 0.5) * in/out - 0.5 clamped at 0, two taps, upper tap clamped to the last row/column; NO antialiasing
    (a down-scale, so an antialiased resize -- newer torchvision's default, Pillow's -- gives different images)."""
    g = torch.Generator().manual_seed(5)
    x = kitti_like(2, 3, SH, SW, g).double()
    OH, OW = 320, 1024

    def taps(n_in, n_out):
        src = np.maximum((np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5, 0.0)
        i0 = np.minimum(np.floor(src).astype(int), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        return i0, i1, src - i0
    y0, y1, fy = taps(SH, OH)
    x0, x1, fx = taps(SW, OW)
    a = x.numpy()
    top = a[:, :, y0][:, :, :, x0] * (1 - fx) + a[:, :, y0][:, :, :, x1] * fx
    bot = a[:, :, y1][:, :, :, x0] * (1 - fx) + a[:, :, y1][:, :, :, x1] * fx
    want = top * (1 - fy)[:, None] + bot * fy[:, None]
    got = tv082.resize(x, (OH, OW)).numpy()
    assert got.shape == (2, 3, OH, OW)
    assert np.abs(got - want).max() <= 1e-12
    got32 = tv082.resize(x.float(), (OH, OW)).numpy()
    assert np.abs(got32 - want).max() <= 2e-6
    # and it is NOT Pillow's antialiased bilinear resize
    pil = np.asarray(Image.fromarray(a[0, 0].astype(np.float32), mode="F").resize((OW, OH), Image.BILINEAR))
    assert np.abs(pil - want[0, 0]).max() > 1e-3


@pytest.mark.gpu
def test_k3_perspective_warp_matches_pillow(golden):
    """The product kernel itself (K3 in warp-only mode, ops.perspective_warp) against Pillow on a spread of the 25 x 13
    pose quads: K3's own bilinear arithmetic in fp32 against Pillow's C double implementation, no oracle in between."""
    from depthmodelhardening_amd import ops
    g = golden("geometry")
    img, msk, start, (l_pad, t_pad, h, w) = _padded_object()
    obj = img[:, :, t_pad:t_pad + h, l_pad:l_pad + w].contiguous()
    mask = msk[:, :, t_pad:t_pad + h, l_pad:l_pad + w].contiguous()
    quads = g["quads"].reshape(-1, 4, 2)[::9]
    coeffs = [tv082.get_perspective_coeffs([list(map(float, p)) for p in start],
                                           [[float(p[0]), float(p[1])] for p in q]) for q in quads]
    c_dev = torch.tensor(coeffs, dtype=torch.float32).cuda()
    w_img, w_msk = ops.perspective_warp(obj.cuda(), mask.cuda(), c_dev, l_pad, t_pad, (SH, SW))
    w_img, w_msk = w_img.cpu().numpy(), w_msk.cpu().numpy()
    for n, cf in enumerate(coeffs):
        for got, src in ((w_msk[n, 0], msk[0, 0]), (w_img[n, 0], img[0, 0]), (w_img[n, 2], img[0, 2])):
            e = np.abs(got - _pil_perspective(src.numpy(), cf))
            assert e.max() <= 2e-3 and (e > 1e-4).mean() <= 2e-3, (n, e.max(), (e > 1e-4).mean())
