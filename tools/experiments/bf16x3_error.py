"""Rounding error of the three-term bf16 split (tools/micro/bf16x3.hip) inside Winograd F(2x2,3x3), beside K10's exact-fp32
F(2,3) and a direct fp32 convolution, all against float64 (CPU, plain torch): the numbers behind DESIGN.md section 9's go / no-go
paragraph (profiles/r06_bf16x3_micro.txt).  python tools/experiments/bf16x3_error.py

The Winograd-domain operands U = G w G^T and V = B^T d B are formed in fp32 as K10 forms them; each is then split once into
x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (round to nearest even), and the element-wise
GEMM over the input channels uses the six products of order <= 2 (hi hi, hi mid, mid hi, hi lo, lo hi, mid mid), every bf16
product exact in fp32 and the sum over terms and channels accumulated in fp32 (what v_mfma_f32_32x32x16_bf16 does up to its
internal summation order).  Also shown: the two-term split (three products), which is what fits the LDS."""
import torch

torch.manual_seed(0)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def split(x, terms):
    out, r = [], x
    for _ in range(terms):
        t = r.to(torch.bfloat16).to(torch.float32)
        out.append(t)
        r = r - t
    return out


def wino(x, w, mode):
    """mode: 'fp32' (K10), ('bf16', terms, products) with products = list of (i, j) term pairs."""
    f = torch.float32
    B_, C, H, W = x.shape
    K = w.shape[0]
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2)
    V = torch.einsum('ij,bcyxjk,lk->bcyxil', BT.to(f), t, BT.to(f))
    U = torch.einsum('ij,kcjl,ml->kcim', G.to(f), w, G.to(f))
    if mode == 'fp32':
        M = torch.einsum('kcim,bcyxim->bkyxim', U, V)
    else:
        _, terms, products = mode
        Us, Vs = split(U, terms), split(V, terms)
        M = torch.zeros((B_, K) + tuple(V.shape[2:]), dtype=f)
        for i, j in products:
            M = M + torch.einsum('kcim,bcyxim->bkyxim', Us[i], Vs[j])      # fp32 accumulation
    Y = torch.einsum('ij,bkyxjl,ml->bkyxim', AT.to(f), M, AT.to(f))
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B_, K, H, W)


SIX = [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]
THREE = [(0, 0), (0, 1), (1, 0)]
print("rel-L2 against float64          direct fp32   F(2,3) fp32   F(2,3) bf16x3 (6 products)   F(2,3) bf16x2 (3 products)")
for C, K, H, W in ((64, 64, 40, 64), (256, 64, 20, 64), (512, 64, 12, 32)):
    x = torch.randn(2, C, H, W, dtype=torch.float64)
    w = torch.randn(K, C, 3, 3, dtype=torch.float64) * (2.0 / (9 * C)) ** 0.5
    ref = torch.nn.functional.conv2d(x, w, None, 1, 1)
    rel = lambda y: float((y.double() - ref).norm() / ref.norm())   # noqa: E731
    yd = torch.nn.functional.conv2d(x.float(), w.float(), None, 1, 1)
    print("C = %3d                         %.3g      %.3g      %.3g                     %.3g" % (
        C, rel(yd), rel(wino(x.float(), w.float(), 'fp32')), rel(wino(x.float(), w.float(), ('bf16', 3, SIX))),
        rel(wino(x.float(), w.float(), ('bf16', 2, THREE)))))
