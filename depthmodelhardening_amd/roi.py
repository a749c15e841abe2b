"""Window plan of the attack's cropped decoder tail (host side, numpy only).

The attack's cost is ``-mean((disp * mask)^2)`` (torchattacks/attacks/phy_obj_atk.py:88-97): it reads the disparity under
the pasted object only.  The last five convolutions of the depth decoder (MD2/networks/depth_decoder.py:51-63 --
upconv(1,0), upconv(1,1), upconv(0,0), upconv(0,1), dispconv(0)) reach a dozen pixels, so inside an attack they are
evaluated on one window per scene instead of the whole frame.  This module turns the per-scene bounding boxes of the
pasted object's mask into those windows:

    level 0 (H x W)        d    dispconv(0) output           >= the mask's box
                           z01  upconv(0,1) output           >= d dilated by 1 (what dispconv(0) reads)
    level 1 (H/2 x W/2)    y00  upconv(0,0) output           >= half of (z01 dilated by 1)
                           z11  upconv(1,1) output           >= y00 dilated by 1
    level 2 (H/4 x W/4)    y10  upconv(1,0) output           >= half of (z11 dilated by 1)

Every window of a step has the same size for all scenes (the convolution kernels take [B, C, h, w] tensors) and its own
origin per scene; sizes are even (the Winograd kernels' tiles) and origins even, so that a window's 2 x 2 tiles and its
nearest-upsampling phase coincide with the full frame's.  A dilated box is clipped to the frame: what lies outside is the
reflection padding, which the glue kernel (csrc/roi_glue.hip) takes from inside the window.
"""
import numpy as np

WINDOWS = ("d", "z01", "y00", "z11", "y10")
LEVEL = {"d": 0, "z01": 0, "y00": 1, "z11": 1, "y10": 2}
_ROW_ALIGN, _COL_ALIGN = 2, 4


def _fit(lo, hi, frame, align):
    """One axis: a common even size >= every (hi - lo) + 1 and per-scene even origins with origin <= lo, origin + size >= hi,
    inside [0, frame]."""
    need = int((hi - lo).max())
    size = -(-(need + 1) // align) * align
    if size >= frame:
        return frame, np.zeros_like(lo)
    org = np.minimum(lo, frame - size) & ~1
    return size, org


def _dilate_clip(org, size, frame):
    return np.maximum(org - 1, 0), np.minimum(org + size + 1, frame)


def _half(lo, hi):
    return lo >> 1, ((hi - 1) >> 1) + 1


class RoiPlan(object):
    """Windows of one attack step.  ``boxes``: int array [B, 4] = (y0, y1, x0, x1), half-open, in the H x W frame of the
    disparity.  ``size[name]`` = (rows, columns), ``org[name]`` = int32 [B, 2]."""

    def __init__(self, boxes, H, W):
        boxes = np.asarray(boxes, dtype=np.int64).reshape(-1, 4)
        if H % 8 or W % 8 or H < 16 or W < 16:
            raise RuntimeError("RoiPlan: frame must be a multiple of 8 and at least 16 x 16")
        self.H, self.W, self.B = int(H), int(W), boxes.shape[0]
        y0 = np.clip(boxes[:, 0], 0, H - 1)
        y1 = np.clip(boxes[:, 1], y0 + 1, H)
        x0 = np.clip(boxes[:, 2], 0, W - 1)
        x1 = np.clip(boxes[:, 3], x0 + 1, W)
        self.size, self.org = {}, {}
        ry, rx = (y0, y1), (x0, x1)
        for name in WINDOWS:
            lvl = LEVEL[name]
            fh, fw = H >> lvl, W >> lvl
            hc, oy = _fit(ry[0], ry[1], fh, _ROW_ALIGN)
            wc, ox = _fit(rx[0], rx[1], fw, _COL_ALIGN)
            self.size[name] = (int(hc), int(wc))
            self.org[name] = np.stack([oy, ox], 1).astype(np.int32)
            # what the convolution producing this window reads: the window dilated by 1, inside the frame
            ry = _dilate_clip(oy, hc, fh)
            rx = _dilate_clip(ox, wc, fw)
            if name in ("z01", "z11"):      # its input is the nearest-x2 upsampling of the next (coarser) window
                ry, rx = _half(*ry), _half(*rx)
        # what is read of the two full-resolution sources: upconv(2,1)'s output (level 2) and feature 0 (level 1)
        self.read_z21 = (ry, rx)

    def table(self):
        """int32 [len(WINDOWS), B, 2] origins, in WINDOWS order."""
        return np.stack([self.org[n] for n in WINDOWS], 0)

    def area_fraction(self):
        """Window area / frame area per window (reporting)."""
        return {n: self.size[n][0] * self.size[n][1] / float((self.H >> LEVEL[n]) * (self.W >> LEVEL[n])) for n in WINDOWS}


def mask_box(quad, src_size, out_size, margin=2):
    """Bounding box, in the resized out_size frame, of everything the warped object can touch.  ``quad``: the integer pixel
    quad [4, 2] (u, v) of PhysicalTrans.objPosOnImage in the src_size = (375, 1242) frame (physicalTrans.py:60-78); the
    perspective warp's bilinear taps reach one pixel beyond it and Resize's two taps (align_corners=False, no antialias)
    one more output pixel; ``margin`` source pixels are added on top.  Returns (y0, y1, x0, x1), half-open, clipped."""
    quad = np.asarray(quad, dtype=np.float64).reshape(4, 2)
    out = []
    for axis, (n_src, n_out) in enumerate(((src_size[0], out_size[0]), (src_size[1], out_size[1]))):
        v = quad[:, 1 - axis]
        lo, hi = v.min() - margin, v.max() + margin
        s = n_src / float(n_out)
        o_lo = int(np.floor((lo - 0.5) / s - 0.5)) - 1
        o_hi = int(np.ceil((hi + 1.5) / s - 0.5)) + 1
        o_lo = min(max(o_lo, 0), n_out - 1)
        o_hi = min(max(o_hi + 1, o_lo + 1), n_out)
        out += [o_lo, o_hi]
    return tuple(out)
