#!/usr/bin/env python3
"""Run the K1/K2 launches a few times at config-2 shape (for rocprofv3)."""
import ctypes as C, os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthmodelhardening_amd import _native as N, ops
B, H, W = 32, 320, 1024
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
noise_mode = int(sys.argv[2]) if len(sys.argv) > 2 else N.NOISE_PHILOX
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1234)
left = F.avg_pool2d(torch.rand(B, 3, H + 4, W + 4, device=dev, generator=g), 5, 1).contiguous()
right = torch.roll(left, 8, 3).contiguous()
colors = [left if s == 0 else F.avg_pool2d(left, 2 ** s).contiguous() for s in range(4)]
K = torch.tensor([[0.58 * W, 0, 0.5 * W, 0], [0, 1.92 * H, 0.5 * H, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device=dev)
inv_K = torch.linalg.pinv(K)
K, inv_K = K.repeat(B, 1, 1).contiguous(), inv_K.repeat(B, 1, 1).contiguous()
T = torch.eye(4, device=dev).repeat(B, 1, 1); T[:, 0, 3] = -0.1
disps = [(0.02 + 0.1 * F.avg_pool2d(torch.rand(B, 1, (H >> s) + 8, (W >> s) + 8, device=dev, generator=g), 9, 1)).contiguous() for s in range(4)]
lib = N.lib()
cfg = dict(F=1, NS=4, min_depth=0.1, max_depth=100.0, variant="md2", automask=True, no_ssim=False, smooth_wt=1e-3, want_to_opt=False, hints=False, noise_mode=noise_mode, seed=1, offset=0)
pa = ops._photo_args(cfg, left, [right], [T], K, inv_K, disps, ())
sm = ops._smooth_args(disps, colors)
sel = torch.empty(B, H, W, device=dev, dtype=torch.uint8)
pp = torch.empty(lib.dmh_photo_partials_size(B, H, W, 4), device=dev)
sp = torch.empty(lib.dmh_smooth_partials_size(C.byref(sm)), device=dev)
fin = torch.empty(N.FIN_SIZE, device=dev); sst = torch.empty(4, B, 2, device=dev)
gvec = torch.zeros(N.FIN_SIZE, device=dev); gvec[0] = 1.0
g_disp = [torch.empty_like(d) for d in disps]
stage = torch.empty(lib.dmh_photo_stage_size(C.byref(pa)), device=dev)
st = N.stream()
nullp, gdp = N.ptr_array([None] * 4), N.ptr_array(g_disp)
for _ in range(reps):
    N.check(lib.dmh_photo_loss_fwd(C.byref(pa), N.ptr(sel), nullp, N.ptr(pp), st))
    N.check(lib.dmh_smooth_loss_fwd(C.byref(sm), N.ptr(sp), st))
    N.check(lib.dmh_loss_finalize(N.ptr(pp), N.ptr(sp), B, H, W, C.byref(sm), 0, 1e-3, N.ptr(fin), N.ptr(sst), st))
    N.check(lib.dmh_photo_loss_bwd(C.byref(pa), N.ptr(sel), N.ptr(gvec), N.ptr(fin), N.ptr(stage), gdp, st))
    N.check(lib.dmh_smooth_loss_bwd(C.byref(sm), N.ptr(gvec), N.ptr(sst), 1e-3, gdp, 1, st))
torch.cuda.synchronize()
print("loss", float(fin[0]))
