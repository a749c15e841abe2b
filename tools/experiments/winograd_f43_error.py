"""Rounding error of Winograd F(4x4,3x3) against F(2x2,3x3) and a direct convolution, all in fp32, measured against float64 (CPU,
plain torch einsum with the standard Cook-Toom matrices at the points 0, +-1, +-2, inf): the numbers behind DESIGN.md section 9
item 1 (why K10 stays F(2,3)).  python tools/experiments/winograd_f43_error.py"""
import torch, numpy as np
torch.manual_seed(0)
def wino(x, w, m, dtype):
    # generic F(m x m, 3x3) via Cook-Toom with points
    if m == 2:
        BT = torch.tensor([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], dtype=torch.float64)
        G = torch.tensor([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], dtype=torch.float64)
        AT = torch.tensor([[1,1,1,0],[0,1,-1,-1]], dtype=torch.float64)
    else:
        BT = torch.tensor([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]], dtype=torch.float64)
        G = torch.tensor([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]], dtype=torch.float64)
        AT = torch.tensor([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]], dtype=torch.float64)
    BT, G, AT = BT.to(dtype), G.to(dtype), AT.to(dtype)
    a = m + 2
    B_, C, H, W = x.shape
    K = w.shape[0]
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    th, tw = H // m, W // m
    # tiles [B,C,th,tw,a,a]
    t = xp.unfold(2, a, m).unfold(3, a, m)
    V = torch.einsum('ij,bcyxjk,lk->bcyxil', BT, t, BT)
    U = torch.einsum('ij,kcjl,ml->kcim', G, w, G)
    M = torch.einsum('kcim,bcyxim->bkyxim', U, V)     # accumulate over c in dtype
    Y = torch.einsum('ij,bkyxjl,ml->bkyxim', AT, M, AT)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B_, K, H, W)
for C, K, H, W in ((64, 64, 40, 64), (256, 64, 20, 64), (512, 64, 12, 32)):
    x = torch.randn(2, C, H, W, dtype=torch.float64)
    w = torch.randn(K, C, 3, 3, dtype=torch.float64) * (2.0 / (9 * C)) ** 0.5
    ref = torch.nn.functional.conv2d(x, w, None, 1, 1)
    out = {}
    for m in (2, 4):
        y = wino(x.float(), w.float(), m, torch.float32).double()
        out[m] = float((y - ref).norm() / ref.norm())
    yd = torch.nn.functional.conv2d(x.float(), w.float(), None, 1, 1).double()
    print("C=%d: rel-L2 vs fp64: direct fp32 %.3g  F(2,3) %.3g  F(4,3) %.3g" % (C, float((yd - ref).norm() / ref.norm()), out[2], out[4]))
