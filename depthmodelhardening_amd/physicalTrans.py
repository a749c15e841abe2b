"""Physical-object EOT transform: same class, methods and argument meaning as the reference's
``physicalTrans.py:11-196``; the pixel work runs in the fused HIP kernel K3 (csrc/eot_paste.hip).

What stays on the host (as in the reference): the 3-D plane corners of the object at distance z0 and
yaw alpha, their projection to an integer pixel quad (np.int32 truncation, physicalTrans.py:75,78) and
the eight torchvision-0.8.2 perspective coefficients of that quad -- 12 tiny solves per attack step.
"""
from math import cos, sin, radians
from random import sample

import numpy as np
import torch

from . import ops
from .my_utils import KITTI_003086_P2, ori_H, ori_W


def read_calib_P2(path):
    """``Calibration(path).P`` (preprocessing/kitti_util.py:61-96): the 3x4 P2 matrix of a KITTI calib file."""
    with open(path, "r") as f:
        for line in f:
            line = line.rstrip()
            if not line:
                continue
            key, value = line.split(":", 1)
            if key == "P2":
                return np.array([float(x) for x in value.split()], dtype=np.float64).reshape(3, 4)
    raise RuntimeError("no P2 entry in calibration file %s" % path)


def get_perspective_coeffs(startpoints, endpoints):
    """torchvision-0.8.2 ``functional._get_perspective_coeffs``: coefficients mapping an OUTPUT pixel to the
    INPUT pixel.  0.8.2 solves the 8x8 system with fp32 ``torch.lstsq`` (since removed from torch); the
    system is square, so it is solved here in float64 and rounded to fp32."""
    a = np.zeros((8, 8), dtype=np.float64)
    for i, (p1, p2) in enumerate(zip(endpoints, startpoints)):
        a[2 * i, :] = [p1[0], p1[1], 1, 0, 0, 0, -p2[0] * p1[0], -p2[0] * p1[1]]
        a[2 * i + 1, :] = [0, 0, 0, p1[0], p1[1], 1, -p2[1] * p1[0], -p2[1] * p1[1]]
    b = np.asarray(startpoints, dtype=np.float64).reshape(8)
    return np.linalg.solve(a, b).astype(np.float32)


class PhysicalTrans(object):
    def __init__(self, obj_img, obj_mask, cfg, output_size,
                 angle_range=list(range(-30, 31, 5)), dist_range=list(range(5, 10, 2))) -> None:
        """
        obj_img: 1,C,H,W tensor;  obj_mask: 1,1,H,W tensor
        cfg: dictionary with 'path' (KITTI calib file; the built-in 003086 P2 is used when it is absent)
        output_size: 4 dimension sequence, must be (_, _, 375, 1242)
        """
        super().__init__()
        self.obj_img = obj_img
        self.obj_mask = obj_mask
        self.cfg = cfg
        path = (cfg or {}).get("path")
        if path is not None and __import__("os").path.exists(path):
            self.P, self.calib_source = read_calib_P2(path), path
        else:
            self.P, self.calib_source = KITTI_003086_P2.copy(), "builtin:003086"
        self.dist_range = dist_range
        self.angle_range = angle_range
        self.output_size = output_size
        # the output size should be: _, _, 375, 1242, otherwise the calibration cannot be directly used.
        assert output_size[2] == ori_H and output_size[3] == ori_W
        self.padding_img()
        veh_h = 1.6    # BMW: height 1.6 m, width 1.82 m (physicalTrans.py:35-48)
        veh_w = 1.82
        cam_h = 1.65
        self.x0 = 0
        self.y0 = cam_h - veh_h / 2
        self.m = veh_w
        self.n = veh_h

    # ------------------------------------------------------------------ geometry (host, float64 like numpy)
    def fromZA2Coord(self, z0, alpha):
        x_offset = cos(radians(alpha)) * self.m / 2
        x1 = self.x0 - x_offset
        x2 = self.x0 + x_offset
        z_offset = sin(radians(alpha)) * self.m / 2
        zl = z0 - z_offset
        zr = z0 + z_offset
        y1 = self.y0 - self.n / 2
        y2 = self.y0 + self.n / 2
        return np.array([[x1, y1, zl], [x2, y1, zr], [x2, y2, zr], [x1, y2, zl]])  # tl, tr, br, bl

    def _project_rect_to_image(self, pts_3d_rect):
        hom = np.hstack((pts_3d_rect, np.ones((pts_3d_rect.shape[0], 1))))
        pts_2d = np.dot(hom, np.transpose(self.P))
        pts_2d[:, 0] /= pts_2d[:, 2]
        pts_2d[:, 1] /= pts_2d[:, 2]
        return pts_2d[:, 0:2]

    def objPosOnImage(self, z0, alpha, K=None):
        """return: [tl, tr, br, bl] pixel indices, N * 2 (u, v), int32 (truncated)."""
        world_coord = self.fromZA2Coord(z0, alpha)
        if K is not None:
            n = world_coord.shape[0]
            points = np.concatenate((world_coord.T, np.ones((1, n))), axis=0)
            cam_points = np.matmul(K[:3, :], points)
            pix_coords = cam_points[:2, :] / (cam_points[[2], :] + 1e-7)
            return pix_coords.T.astype(np.int32)
        return self._project_rect_to_image(world_coord).astype(np.int32)

    def _objPosOnImage_w_trans(self, T, z0, alpha, K=None):
        world_coord = self.fromZA2Coord(z0, alpha)
        n = world_coord.shape[0]
        points = np.concatenate((world_coord.T, np.ones((1, n))), axis=0)
        if K is not None:
            cam_points = np.matmul(np.matmul(K, T)[:3, :], points)
            pix_coords = cam_points[:2, :] / (cam_points[[2], :] + 1e-7)
            return pix_coords.T.astype(np.int32)
        return self._project_rect_to_image(np.matmul(T, points).T[:, :3]).astype(np.int32)

    def padding_img(self):
        """Record where the zero-padded patch sits in the 375x1242 frame (the kernel pads implicitly)."""
        _, _, H, W = self.obj_img.size()
        _, _, H_out, W_out = self.output_size
        self.l_pad = (W_out - W) // 2
        self.t_pad = (H_out - H) // 2
        l_pad, t_pad = self.l_pad, self.t_pad
        self.pos_obj_img_start = [[l_pad, t_pad], [l_pad + W, t_pad], [l_pad + W, t_pad + H], [l_pad, t_pad + H]]

    def reset_img(self, obj_img, obj_mask):
        self.obj_img = obj_img
        self.obj_mask = obj_mask
        self.padding_img()

    # ------------------------------------------------------------------ sampling + coefficients
    def draw_samples(self, batch_size, z0_sample=None, alpha_sample=None, rs=None):
        """The (z0, alpha) draws of project() (physicalTrans.py:146-155), same RNG, same order."""
        if z0_sample is None:
            z0_sample = rs.choice(self.dist_range, batch_size, replace=False) if rs else \
                sample(self.dist_range, batch_size)
        if alpha_sample is None:
            alpha_sample = rs.choice(self.angle_range, batch_size, replace=False) if rs else \
                sample(self.angle_range, batch_size)
        return z0_sample, alpha_sample

    def coeffs_for(self, z0_sample, alpha_sample, K=None, T=None):
        """[N,8] fp32 perspective coefficients (host numpy) for the given samples.  (z0, alpha) come from small finite
        ranges (13 x 25 in training), so each (camera, pose) is solved once and memoised: the per-step host cost of
        ~200 8x8 solves otherwise leaves the GPU idle for milliseconds (the host is the slower side of the pipeline)."""
        out = np.zeros((len(z0_sample), 8), dtype=np.float32)
        start = [[float(v) for v in p] for p in self.pos_obj_img_start]
        cam = (None if K is None else np.asarray(K).tobytes(), None if T is None else np.asarray(T).tobytes(),
               tuple(map(tuple, start)))
        memo = self.__dict__.setdefault("_coeff_memo", {})
        warmed = self.__dict__.setdefault("_coeff_warm", set())
        if cam not in warmed and len(z0_sample) > 1:        # first use of a camera: solve the whole pose grid once
            warmed.add(cam)
            grid = [(z, al) for z in self.dist_range for al in self.angle_range]
            self.coeffs_for([g[0] for g in grid] + [grid[0][0]], [g[1] for g in grid] + [grid[0][1]], K, T)
        for i in range(len(z0_sample)):
            key = (float(z0_sample[i]), float(alpha_sample[i]), cam)
            c = memo.get(key)
            if c is None:
                quad = self.objPosOnImage(z0_sample[i], alpha_sample[i], K) if T is None else \
                    self._objPosOnImage_w_trans(T, z0_sample[i], alpha_sample[i], K)
                c = memo[key] = np.asarray(get_perspective_coeffs(start, [[float(v) for v in p] for p in quad]),
                                           dtype=np.float32)
            out[i] = c
        return out

    def mask_boxes(self, z0_sample, alpha_sample, out_size):
        """int [N, 4] = (y0, y1, x0, x1), half-open, in the Resize'd ``out_size`` frame: per sample a box that contains
        every pixel the pasted object can touch -- the support of ``obj_masks_out`` of phy_obj_atk.py:87-90 -- from the
        integer quad of objPosOnImage (roi.mask_box).  The attack evaluates its cost inside these boxes only."""
        from .roi import mask_box
        memo = self.__dict__.setdefault("_box_memo", {})
        out = np.zeros((len(z0_sample), 4), dtype=np.int64)
        for i in range(len(z0_sample)):
            key = (float(z0_sample[i]), float(alpha_sample[i]), int(out_size[0]), int(out_size[1]))
            b = memo.get(key)
            if b is None:
                b = memo[key] = mask_box(self.objPosOnImage(z0_sample[i], alpha_sample[i]),
                                         (self.output_size[2], self.output_size[3]), out_size)
            out[i] = b
        return out

    def _warp(self, coeffs_np):
        dev = self.obj_img.device
        coeffs = torch.from_numpy(np.ascontiguousarray(coeffs_np)).to(dev)
        return ops.perspective_warp(self.obj_img, self.obj_mask.to(dev), coeffs, self.l_pad, self.t_pad,
                                    (self.output_size[2], self.output_size[3]))

    # ------------------------------------------------------------------ reference surface
    def project(self, is_all=False, batch_size=1, z0_sample=None, alpha_sample=None, K=None, rs=None):
        if is_all:
            z0_sample, alpha_sample = [], []
            for z0 in self.dist_range:
                for alpha in self.angle_range:
                    z0_sample.append(z0)
                    alpha_sample.append(alpha)
        else:
            z0_sample, alpha_sample = self.draw_samples(batch_size, z0_sample, alpha_sample, rs)
            z0_sample, alpha_sample = z0_sample[:batch_size], alpha_sample[:batch_size]
        obj_imgs_out, obj_masks_out = self._warp(self.coeffs_for(z0_sample, alpha_sample, K))
        return obj_imgs_out, obj_masks_out, z0_sample, alpha_sample

    def project_w_trans(self, T, z0_sample, alpha_sample, K=None):
        """K: camera intrinsics 4x4, T: camera extrinsics 4x4 (physicalTrans.py:168-196)."""
        return self._warp(self.coeffs_for(z0_sample, alpha_sample, K, T))
