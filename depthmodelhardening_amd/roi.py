"""Window plan of the attack's cropped decoder tail (host side, numpy only).

The attack's cost is ``-mean((disp * mask)^2)`` (torchattacks/attacks/phy_obj_atk.py:88-97): it reads the disparity under
the pasted object only.  The last six convolutions of the depth decoder (MD2/networks/depth_decoder.py:51-63 --
upconv(2,1), upconv(1,0), upconv(1,1), upconv(0,0), upconv(0,1), dispconv(0)) reach two dozen pixels, so inside an attack they are
evaluated on one window per scene instead of the whole frame.  This module turns the per-scene bounding boxes of the
pasted object's mask into those windows:

    level 0 (H x W)        d    dispconv(0) output           >= the mask's box
                           z01  upconv(0,1) output           >= d dilated by 1 (what dispconv(0) reads)
    level 1 (H/2 x W/2)    y00  upconv(0,0) output           >= half of (z01 dilated by 1)
                           z11  upconv(1,1) output           >= y00 dilated by 1
    level 2 (H/4 x W/4)    y10  upconv(1,0) output           >= half of (z11 dilated by 1)
                           z21  upconv(2,1) output           >= y10 dilated by 1

Every window of a step has the same size for all scenes (the convolution kernels take [B, C, h, w] tensors) and its own
origin per scene; sizes are even (the Winograd kernels' tiles) and origins even, so that a window's 2 x 2 tiles and its
nearest-upsampling phase coincide with the full frame's.  A dilated box is clipped to the frame: what lies outside is the
reflection padding, which the glue kernel (csrc/roi_glue.hip) takes from inside the window.
"""
import numpy as np

WINDOWS = ("d", "z01", "y00", "z11", "y10", "z21")
LEVEL = {"d": 0, "z01": 0, "y00": 1, "z11": 1, "y10": 2, "z21": 2, "r_y20": 3, "r_f1": 2, "r_f0": 1, "gz": 1, "l1": 2}
_UPSAMPLED_INPUT = ("z01", "z11", "z21")    # the convolution producing these reads the nearest-x2 upsampling of its source
# rectangles of whole-frame tensors: what the tail reaches of upconv(2,0)'s output, feature 1 and feature 0 (the regions its
# backward writes), and the encoder head's backward windows: "gz" the gradient of conv1's output that the image window reads,
# "l1" the common window of layer1's four backward convolutions
REGIONS = ("r_y20", "r_f1", "r_f0", "gz", "l1")
TABLE = WINDOWS + REGIONS
_ROW_ALIGN, _COL_ALIGN = 2, 4
_L1_RING = 4                                # layer1 = two BasicBlocks = four 3x3 convolutions: each spoils one ring


def _fit(lo, hi, frame, align):
    """One axis: a common even size >= every (hi - lo) + 1 and per-scene even origins with origin <= lo, origin + size >= hi,
    inside [0, frame]."""
    need = int((hi - lo).max())
    size = -(-(need + 1) // align) * align
    if size >= frame:
        return frame, np.zeros_like(lo)
    org = np.minimum(lo, frame - size) & ~1
    return size, org


def _dilate_clip(org, size, frame, r=1):
    return np.maximum(org - r, 0), np.minimum(org + size + r, frame)


def _half(lo, hi):
    return lo >> 1, ((hi - 1) >> 1) + 1


class RoiPlan(object):
    """Windows of one attack step.  ``boxes``: int array [B, 4] = (y0, y1, x0, x1), half-open, in the H x W frame of the
    disparity.  ``size[name]`` = (rows, columns), ``org[name]`` = int32 [B, 2] for every name in TABLE."""

    def __init__(self, boxes, H, W):
        boxes = np.asarray(boxes, dtype=np.int64).reshape(-1, 4)
        if H % 8 or W % 8 or H < 16 or W < 16:
            raise RuntimeError("RoiPlan: frame must be a multiple of 8 and at least 16 x 16")
        self.H, self.W, self.B = int(H), int(W), boxes.shape[0]
        self.head_windowed = False      # set by the encoder when its head's backward will run on the windows below
        y0 = np.clip(boxes[:, 0], 0, H - 1)
        y1 = np.clip(boxes[:, 1], y0 + 1, H)
        x0 = np.clip(boxes[:, 2], 0, W - 1)
        x1 = np.clip(boxes[:, 3], x0 + 1, W)
        self.size, self.org = {}, {}
        reads = {}
        ry, rx = (y0, y1), (x0, x1)
        for name in WINDOWS:
            lvl = LEVEL[name]
            fh, fw = H >> lvl, W >> lvl
            self._put(name, ry, rx, _ROW_ALIGN, _COL_ALIGN)
            # what the convolution producing this window reads: the window dilated by 1, inside the frame
            (hc, wc), o = self.size[name], self.org[name].astype(np.int64)
            ry = _dilate_clip(o[:, 0], hc, fh)
            rx = _dilate_clip(o[:, 1], wc, fw)
            reads[name] = (ry, rx)
            if name in _UPSAMPLED_INPUT:    # its input is the nearest-x2 upsampling of the next (coarser) window
                ry, rx = _half(*ry), _half(*rx)
        # ---- encoder head (backward only).  The image window is "d" (it holds the box); conv1's 7x7/2 adjoint reads rows
        # Y-1 .. Y+2 of its output gradient for image rows 2Y, 2Y+1
        (hd, wd), od = self.size["d"], self.org["d"].astype(np.int64)
        gy = np.maximum((od[:, 0] >> 1) - 1, 0), np.minimum(((od[:, 0] + hd) >> 1) + 2, H >> 1)
        gx = np.maximum((od[:, 1] >> 1) - 1, 0), np.minimum(((od[:, 1] + wd) >> 1) + 2, W >> 1)
        self._put("gz", gy, gx, _ROW_ALIGN, _COL_ALIGN)
        (hs, ws), os_ = self.size["gz"], self.org["gz"].astype(np.int64)
        # pooling cells that cover rows [a, b) of the 1/2 map: a >> 1 .. b >> 1; layer1's four convolutions spoil four rings
        qy = np.maximum((os_[:, 0] >> 1) - _L1_RING, 0), np.minimum(((os_[:, 0] + hs) >> 1) + 1 + _L1_RING, H >> 2)
        qx = np.maximum((os_[:, 1] >> 1) - _L1_RING, 0), np.minimum(((os_[:, 1] + ws) >> 1) + 1 + _L1_RING, W >> 2)
        self._put("l1", qy, qx, 4, 16)
        # ---- regions of the whole-frame sources the tail's backward writes
        self._put("r_y20", *(_half(*reads["z21"][0]), _half(*reads["z21"][1])), 2, 2)
        self._put("r_f1", reads["z21"][0], reads["z21"][1], 2, 2)
        # feature 0's gradient is read by the encoder head on "gz": the written rectangle holds both
        fy = np.minimum(reads["z11"][0][0], os_[:, 0]), np.maximum(reads["z11"][0][1], os_[:, 0] + hs)
        fx = np.minimum(reads["z11"][1][0], os_[:, 1]), np.maximum(reads["z11"][1][1], os_[:, 1] + ws)
        self._put("r_f0", fy, fx, 2, 2)

    def _put(self, name, ry, rx, row_align, col_align):
        lvl = LEVEL[name]
        hc, oy = _fit(ry[0], ry[1], self.H >> lvl, row_align)
        wc, ox = _fit(rx[0], rx[1], self.W >> lvl, col_align)
        self.size[name] = (int(hc), int(wc))
        self.org[name] = np.stack([oy, ox], 1).astype(np.int32)

    def table(self):
        """int32 [len(TABLE), B, 2] origins, in TABLE order."""
        return np.stack([self.org[n] for n in TABLE], 0)

    def area_fraction(self):
        """Window area / frame area per window (reporting)."""
        return {n: self.size[n][0] * self.size[n][1] / float((self.H >> LEVEL[n]) * (self.W >> LEVEL[n])) for n in TABLE}


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
