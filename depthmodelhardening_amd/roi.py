"""Window plan of the attack's cropped network passes (host side, numpy only).

The attack's cost is ``-mean((disp * mask)^2)`` (torchattacks/attacks/phy_obj_atk.py:88-97): it reads the disparity under
the pasted object only, and the patch gradient reads ``d cost / d image`` under the object only (physicalTrans.py:156-165).
A 3x3 convolution reaches one pixel, so inside an attack the depth decoder (MD2/networks/depth_decoder.py:51-63) below its
first stage is evaluated -- forward and backward, exactly -- on one window per scene instead of the whole frame, and so is the
backward pass of the encoder's head (conv1 ... layer1, MD2/networks/resnet_encoder.py:85-98).  This module turns the per-scene
bounding boxes of the pasted object's mask into those windows.  Decoder chain, from the disparity head upwards (STAGES):

    level 0 (H x W)        d    dispconv(0) output     >= the mask's box
                           z01  upconv(0,1) output     >= d dilated by 1 (what dispconv(0) reads)
    level 1 (H/2 x W/2)    y00  upconv(0,0) output     >= half of (z01 dilated by 1)       (nearest x2 upsampling between them)
                           z11  upconv(1,1) output     >= y00 dilated by 1
    level 2                y10, z21        level 3     y20, z31        level 4     y30, z41       ... the same pattern

``depth`` (2, 3 or 4) is the level of the deepest windowed stage: z{depth}1 reads the whole-frame output of upconv(depth,0)
and encoder feature depth-1.  Every window of a step has the same size for all scenes (the convolution kernels take
[B, C, h, w] tensors) and its own origin per scene; sizes and origins are even, so that a window's 2 x 2 Winograd tiles and
its nearest-upsampling phase coincide with the full frame's.  A dilated box is clipped to the frame: what lies outside is the
reflection padding, which the glue kernel (csrc/roi_glue.hip) takes from inside the window.
"""
import numpy as np

# (window, level, its convolution reads an upsampled source, index of the encoder feature concatenated to that source)
STAGES = (("d", 0, False, None), ("z01", 0, True, None), ("y00", 1, False, None), ("z11", 1, True, 0),
          ("y10", 2, False, None), ("z21", 2, True, 1), ("y20", 3, False, None), ("z31", 3, True, 2),
          ("y30", 4, False, None), ("z41", 4, True, 3))
WINDOWS = tuple(s[0] for s in STAGES)
LEVEL = {s[0]: s[1] for s in STAGES}
MAX_DEPTH = 4
# rectangles of whole-frame tensors.  r_f{k}: what the tail reads of encoder feature k (= the region its backward writes of
# that feature's gradient); r_y{k}0: what it reads of upconv(k,0)'s whole-frame output when the chain starts at level k;
# "gz": the gradient of conv1's output that the image window reads; "l1": the common window of layer1's four backward
# convolutions (encoder head)
# Incremental forward of the encoder head (ops.encoder_head_incremental): the pasted object changes the image inside the
# box only, so conv1 ... layer1 are recomputed on "hl" (1/4 map; "hz" = the same window on the 1/2 map, twice the size) and
# the part of feature 1 that really changes, "f1s", is written into the cached feature of the clean scenes
# ... and layer2 on "h3" (1/8 map; "h3in" = the same window on the 1/4 map, its input), writing "f2s" into the cached
# feature 2; "hl_rel" = the origin of "hl" inside "h3in" (layer2's backward hands layer1's its input gradient as that window)
REGIONS = ("r_f0", "r_f1", "r_f2", "r_f3", "r_y20", "r_y30", "r_y40", "gz", "l1", "hl", "hz", "f1s", "h3", "h3in", "f2s",
           "hl_rel")
LEVEL.update({"r_f0": 1, "r_f1": 2, "r_f2": 3, "r_f3": 4, "r_y20": 3, "r_y30": 4, "r_y40": 5, "gz": 1, "l1": 2, "hl": 2,
              "hz": 1, "f1s": 2, "h3": 3, "h3in": 2, "f2s": 3, "hl_rel": 2})
_HEAD_RING = 9      # max-pool (1) + layer1 forward (4) + what layer1's backward needs around its own target (4)
_L2_FWD_RING = 5    # layer2 forward on a compact window: the stride-2 entry + three convolutions spoil 4 rings (top / left)
_L2_BWD_RING = 8    # ... and its backward three more before the stride-2 adjoint reads it
TABLE = WINDOWS + REGIONS
_L1_RING = 4                                # layer1 = two BasicBlocks = four 3x3 convolutions: each spoils one ring


def _size_for(need, frame):
    """Common window size along one axis: the smallest even size >= need + 1 (origins are rounded down to even).  Rounding up
    to whole Winograd regions (4 x 16 tiles = 8 x 32 pixels) was tried: the slack of one level is the need of the next, and
    three levels up the windows had grown by half; a ragged last region costs less."""
    return min(-(-(need + 1) // 2) * 2, frame)


def _fit(lo, hi, frame, min_size=0):
    """One axis: a common size (at least ``min_size``) and per-scene even origins with origin <= lo, origin + size >= hi,
    inside [0, frame]."""
    size = min(max(_size_for(int((hi - lo).max()), frame), int(min_size)), frame)
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
    disparity.  ``size[name]`` = (rows, columns), ``org[name]`` = int32 [B, 2] for every name in TABLE; ``depth`` = level of
    the deepest windowed decoder stage (the windows above it are planned too, and simply unused)."""

    def __init__(self, boxes, H, W, depth=3, min_size=None):
        """``min_size`` (a ``size`` dict of another plan, or None): no window comes out smaller than its entry there.  Windows
        only have to COVER what is read, so a larger one is as exact as the smallest; common_size_plans() uses this to give all
        steps of an attack the same tensor shapes (one captured HIP graph replays them all)."""
        boxes = np.asarray(boxes, dtype=np.int64).reshape(-1, 4)
        self._min = dict(min_size) if min_size else {}
        if H % 32 or W % 32 or H < 64 or W < 64:
            raise RuntimeError("RoiPlan: frame must be a multiple of 32 and at least 64 x 64")
        if depth not in (2, 3, 4):
            raise RuntimeError("RoiPlan: depth must be 2, 3 or 4")
        self.H, self.W, self.B, self.depth = int(H), int(W), boxes.shape[0], int(depth)
        self.head_windowed = False      # set by the encoder when its head's backward will run on the windows below
        self.f0_compact = False         # set by the encoder when feature 0 is handed on as its "hz" window only
        y0 = np.clip(boxes[:, 0], 0, H - 1)
        y1 = np.clip(boxes[:, 1], y0 + 1, H)
        x0 = np.clip(boxes[:, 2], 0, W - 1)
        x1 = np.clip(boxes[:, 3], x0 + 1, W)
        self.size, self.org = {}, {}
        reads = {}
        ry, rx = (y0, y1), (x0, x1)
        for name, lvl, up, skip in STAGES:
            fh, fw = H >> lvl, W >> lvl
            self._put(name, ry, rx)
            # what the convolution producing this window reads: the window dilated by 1, inside the frame
            (hc, wc), o = self.size[name], self.org[name].astype(np.int64)
            ry = _dilate_clip(o[:, 0], hc, fh)
            rx = _dilate_clip(o[:, 1], wc, fw)
            reads[name] = (ry, rx)
            if up:                      # its input is the nearest-x2 upsampling of the next (coarser) window
                ry, rx = _half(*ry), _half(*rx)
                if lvl >= 2:
                    self._put("r_y%d0" % lvl, ry, rx)
                if skip is not None and skip >= 1:
                    self._put("r_f%d" % skip, *reads[name])
        # ---- encoder head (backward only).  The image window is "d" (it holds the box); conv1's 7x7/2 adjoint reads rows
        # Y-1 .. Y+2 of its output gradient for image rows 2Y, 2Y+1
        (hd, wd), od = self.size["d"], self.org["d"].astype(np.int64)
        gy = np.maximum((od[:, 0] >> 1) - 1, 0), np.minimum(((od[:, 0] + hd) >> 1) + 2, H >> 1)
        gx = np.maximum((od[:, 1] >> 1) - 1, 0), np.minimum(((od[:, 1] + wd) >> 1) + 2, W >> 1)
        self._put("gz", gy, gx)
        (hs, ws), os_ = self.size["gz"], self.org["gz"].astype(np.int64)
        # pooling cells that cover rows [a, b) of the 1/2 map: a >> 1 .. b >> 1; layer1's four convolutions spoil four rings
        qy = np.maximum((os_[:, 0] >> 1) - _L1_RING, 0), np.minimum(((os_[:, 0] + hs) >> 1) + 1 + _L1_RING, H >> 2)
        qx = np.maximum((os_[:, 1] >> 1) - _L1_RING, 0), np.minimum(((os_[:, 1] + ws) >> 1) + 1 + _L1_RING, W >> 2)
        hq = min(-(-(int((qy[1] - qy[0]).max()) + 1) // 8) * 8, H >> 2)      # same-size convolutions: whole 4 x 16-tile regions
        wq = min(-(-(int((qx[1] - qx[0]).max()) + 1) // 32) * 32, W >> 2)
        hq, wq = max(hq, self._min.get("l1", (0, 0))[0]), max(wq, self._min.get("l1", (0, 0))[1])
        self.size["l1"] = (int(hq), int(wq))
        self.org["l1"] = np.stack([np.minimum(qy[0], (H >> 2) - hq) & ~1, np.minimum(qx[0], (W >> 2) - wq) & ~1], 1).astype(np.int32)
        # feature 0's gradient is read by the encoder head on "gz": the rectangle the tail writes holds both
        fy = np.minimum(reads["z11"][0][0], os_[:, 0]), np.maximum(reads["z11"][0][1], os_[:, 0] + hs)
        fx = np.minimum(reads["z11"][1][0], os_[:, 1]), np.maximum(reads["z11"][1][1], os_[:, 1] + ws)
        self._put("r_f0", fy, fx)
        # ---- incremental forward of the head.  Image rows [y0, y1) change; conv1 (7x7/2, pad 3) output row r reads rows
        # 2r-3 .. 2r+3, the 3x3/2 max-pool cell i reads rows 2i-1 .. 2i+1, layer1's four convolutions reach one cell each
        zy = np.maximum((y0 - 3) >> 1, 0), np.minimum(((y1 + 2) >> 1) + 1, H >> 1)
        zx = np.maximum((x0 - 3) >> 1, 0), np.minimum(((x1 + 2) >> 1) + 1, W >> 1)
        py = zy[0] >> 1, np.minimum((zy[1] >> 1) + 1, H >> 2)
        px = zx[0] >> 1, np.minimum((zx[1] >> 1) + 1, W >> 2)
        # (exactly the changed cells, no alignment slack: one more cell would lie in the ring the compact window spoils)
        for ax, (lo, hi, frame) in enumerate(((np.maximum(py[0] - 4, 0), np.minimum(py[1] + 4, H >> 2), H >> 2),
                                              (np.maximum(px[0] - 4, 0), np.minimum(px[1] + 4, W >> 2), W >> 2))):
            size = min(max(int((hi - lo).max()), self._min.get("f1s", (0, 0))[ax]), frame)
            fs = (size, np.minimum(lo, frame - size)) if ax == 0 else fs + (size, np.minimum(lo, frame - size))
        self.size["f1s"] = (fs[0], fs[2])
        self.org["f1s"] = np.stack([fs[1], fs[3]], 1).astype(np.int32)
        # a window that min_size made larger than the changed cells + 4 is written into the cached feature all the same: the
        # compact window below must then hold valid values on all of it, i.e. treat "f1s" shrunk by layer1's reach as changed
        py = np.minimum(py[0], fs[1] + 4), np.maximum(py[1], fs[1] + fs[0] - 4)
        px = np.minimum(px[0], fs[3] + 4), np.maximum(px[1], fs[3] + fs[2] - 4)
        # one compact window serves forward and backward: it holds the changed cells and the cells the backward reads (the
        # pooling cells under "gz"), plus the rings the same-size convolutions and the pooling spoil at its edge
        by = np.minimum(py[0], os_[:, 0] >> 1), np.maximum(py[1], ((os_[:, 0] + hs) >> 1) + 1)
        bx = np.minimum(px[0], os_[:, 1] >> 1), np.maximum(px[1], ((os_[:, 1] + ws) >> 1) + 1)
        self._put("hl", (np.maximum(by[0] - _HEAD_RING, 0), np.minimum(by[1] + _HEAD_RING, H >> 2)),
                  (np.maximum(bx[0] - _HEAD_RING, 0), np.minimum(bx[1] + _HEAD_RING, W >> 2)))
        hl, ol = self.size["hl"], self.org["hl"]
        self.size["hz"], self.org["hz"] = (2 * hl[0], 2 * hl[1]), (2 * ol).astype(np.int32)
        # ---- layer2 on a compact window of the 1/8 map.  Forward: the changed cells of feature 1 ("f1s") reach output cell i
        # of the stride-2 entry (rows 2i-1 .. 2i+1) and three more convolutions; backward: layer1's backward reads feature
        # 1's gradient on all of "hl", which the stride-2 adjoint forms from the cells above it
        f1o, (f1h, f1w) = self.org["f1s"].astype(np.int64), self.size["f1s"]
        hlo, (hlh, hlw) = self.org["hl"].astype(np.int64), self.size["hl"]
        p3y = f1o[:, 0] >> 1, np.minimum(((f1o[:, 0] + f1h) >> 1) + 1, H >> 3)
        p3x = f1o[:, 1] >> 1, np.minimum(((f1o[:, 1] + f1w) >> 1) + 1, W >> 3)
        fs = ()
        for ax, (lo, hi, frame) in enumerate(((np.maximum(p3y[0] - 3, 0), np.minimum(p3y[1] + 3, H >> 3), H >> 3),
                                              (np.maximum(p3x[0] - 3, 0), np.minimum(p3x[1] + 3, W >> 3), W >> 3))):
            size = min(max(int((hi - lo).max()), self._min.get("f2s", (0, 0))[ax]), frame)
            fs += (size, np.minimum(lo, frame - size))
        self.size["f2s"] = (fs[0], fs[2])
        self.org["f2s"] = np.stack([fs[1], fs[3]], 1).astype(np.int32)
        f2o = self.org["f2s"].astype(np.int64)
        q3y = hlo[:, 0] >> 1, np.minimum(((hlo[:, 0] + hlh) >> 1) + 1, H >> 3)
        q3x = hlo[:, 1] >> 1, np.minimum(((hlo[:, 1] + hlw) >> 1) + 1, W >> 3)
        cy = (np.maximum(np.minimum(f2o[:, 0] - _L2_FWD_RING, q3y[0] - _L2_BWD_RING), 0),
              np.minimum(np.maximum(f2o[:, 0] + fs[0] + _L2_FWD_RING, q3y[1] + _L2_BWD_RING), H >> 3))
        cx = (np.maximum(np.minimum(f2o[:, 1] - _L2_FWD_RING, q3x[0] - _L2_BWD_RING), 0),
              np.minimum(np.maximum(f2o[:, 1] + fs[2] + _L2_FWD_RING, q3x[1] + _L2_BWD_RING), W >> 3))
        self._put("h3", cy, cx)
        h3, o3 = self.size["h3"], self.org["h3"]
        self.size["h3in"], self.org["h3in"] = (2 * h3[0], 2 * h3[1]), (2 * o3).astype(np.int32)
        self.size["hl_rel"], self.org["hl_rel"] = (hlh, hlw), (hlo - 2 * o3.astype(np.int64)).astype(np.int32)
        rel = self.org["hl_rel"].astype(np.int64)
        self.layer2_incremental_ok = bool((rel >= 0).all() and (rel[:, 0] + hlh <= 2 * h3[0]).all()
                                          and (rel[:, 1] + hlw <= 2 * h3[1]).all() and (rel % 2 == 0).all())
        # the tail reads feature 0 inside "r_f0" and the head's backward reads conv1's gradient inside "gz": both lie in "hz"
        for nm in ("r_f0", "gz"):
            (hh, ww), oo = self.size[nm], self.org[nm]
            self.head_incremental_ok = bool((oo >= 2 * ol).all() and (oo[:, 0] + hh <= 2 * ol[:, 0] + 2 * hl[0]).all()
                                            and (oo[:, 1] + ww <= 2 * ol[:, 1] + 2 * hl[1]).all()
                                            and getattr(self, "head_incremental_ok", True))

    def _put(self, name, ry, rx):
        lvl = LEVEL[name]
        mh, mw = self._min.get(name, (0, 0))
        hc, oy = _fit(ry[0], ry[1], self.H >> lvl, mh)
        wc, ox = _fit(rx[0], rx[1], self.W >> lvl, mw)
        self.size[name] = (int(hc), int(wc))
        self.org[name] = np.stack([oy, ox], 1).astype(np.int32)

    def table(self):
        """int32 [len(TABLE), B, 2] origins, in TABLE order."""
        return np.stack([self.org[n] for n in TABLE], 0)

    def bind_table(self, tab):
        """Tie the device copy of ``table()`` to this plan.  The kernels trust the device origins (a window is read and written
        at org + (i, j) without a frame-bound check), so the plan and its table must not be paired up by hand at every call:
        once bound, ``device_table`` hands out this tensor and refuses any other.  DMH_ROI_CHECK=1 also compares the device
        copy with the host table (one synchronising read; debugging)."""
        import os
        import torch
        want = (len(TABLE), self.B, 2)
        if tuple(tab.shape) != want or tab.dtype != torch.int32 or not tab.is_contiguous():
            raise RuntimeError("RoiPlan: the origin table must be a contiguous int32 %s tensor" % (want,))
        if os.environ.get("DMH_ROI_CHECK", "0") == "1" and not np.array_equal(tab.cpu().numpy(), self.table()):
            raise RuntimeError("RoiPlan: the device table does not hold this plan's origins")
        self._tab = tab
        return tab

    def device_table(self, device, given=None):
        """The device table of this plan on ``device``: the bound one (``given``, if passed, must BE it -- a table of another
        plan is an error, not silently wrong windows), else ``given`` is bound, else the host table is uploaded and bound."""
        tab = getattr(self, "_tab", None)
        if tab is None:
            if given is None:
                import torch
                given = torch.from_numpy(np.ascontiguousarray(self.table())).to(device)
            return self.bind_table(given)
        if given is not None and (given.data_ptr() != tab.data_ptr() or tuple(given.shape) != tuple(tab.shape)):
            raise RuntimeError("RoiPlan: this origin table belongs to another plan (plans[s] paired with tabs[t != s]?)")
        return tab

    def area_fraction(self):
        """Window area / frame area per window (reporting)."""
        return {n: self.size[n][0] * self.size[n][1] / float((self.H >> LEVEL[n]) * (self.W >> LEVEL[n])) for n in TABLE}


def common_size_plans(boxes_per_step, H, W, depth=3, max_rounds=8):
    """One RoiPlan per entry of ``boxes_per_step`` (the steps of one attack), all with the SAME window sizes and the same
    path flags, so that every step launches the same kernels on tensors of the same shapes: the element-wise maximum of the
    steps' sizes is fed back as ``min_size`` until nothing grows any more (a larger window makes the windows derived from it
    larger, hence the iteration; sizes are bounded by the frame).  Returns None when the steps cannot be brought to one shape
    (flags that differ, no fixed point within ``max_rounds``): the caller then runs its steps one by one."""
    mins = None
    for _ in range(max_rounds):
        plans = [RoiPlan(b, H, W, depth=depth, min_size=mins) for b in boxes_per_step]
        top = {n: (max(p.size[n][0] for p in plans), max(p.size[n][1] for p in plans)) for n in TABLE}
        if all(p.size == top for p in plans):
            flags = {(p.layer2_incremental_ok, p.head_incremental_ok) for p in plans}
            return plans if len(flags) == 1 else None
        mins = top
    return None


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
