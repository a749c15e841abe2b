"""CPU restatement (numpy, fp64) of the ALGORITHMS inside the network kernels K9-K13 -- TEST INFRASTRUCTURE ONLY.

The reference runs these layers as torch.nn modules (torchvision BasicBlock under
MD2/networks/resnet_encoder.py:85-98; Conv3x3, MD2/layers.py:127-141; depth_decoder.py:38-44), so the
result oracle for K9-K13 is torch.nn itself on the same inputs (tests/test_gpu_trainer.py,
__graft_entry__.smoke()).  This file pins the *formulations* the kernels use against that oracle on the CPU
(tests/test_oracle_golden.py): the Winograd F(2x2,3x3) identity with the exact transform matrices and the
flipped/transposed filter of the backward-data pass (K10), the stride-2 parity-gather form of the 7x7 stem
convolution's image gradient (K12), and the shifted-sum / Chan combination of the train-mode BatchNorm
statistics (K9).
"""
import numpy as np

# F(2x2, 3x3): Y = A^T [ (G g G^T) .* (B^T d B) ] A   (Lavin & Gray 2016; the matrices of csrc/wino_conv.hip)
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def conv3x3_direct(x, w, pad):
    """y[b,k,oy,ox] = sum_{c,ky,kx} xpad[b,c,oy+ky,ox+kx] w[k,c,ky,kx]  (nn.Conv2d cross-correlation, zero padding)."""
    x = np.pad(np.asarray(x, np.float64), ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    B, C, H, W = x.shape
    K = w.shape[0]
    y = np.zeros((B, K, H - 2, W - 2))
    for ky in range(3):
        for kx in range(3):
            y += np.einsum("bchw,kc->bkhw", x[:, :, ky:ky + H - 2, kx:kx + W - 2], np.asarray(w, np.float64)[:, :, ky, kx])
    return y


def conv3x3_winograd(x, w, pad):
    """The same convolution by Winograd F(2x2,3x3) tiles, as K10 computes it (output height / width even)."""
    x = np.pad(np.asarray(x, np.float64), ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    B, C, H, W = x.shape
    K = w.shape[0]
    Ho, Wo = H - 2, W - 2
    assert Ho % 2 == 0 and Wo % 2 == 0
    U = np.einsum("ij,kcjl,ml->kcim", G, np.asarray(w, np.float64), G)            # [K,C,4,4]
    y = np.zeros((B, K, Ho, Wo))
    for ty in range(Ho // 2):
        for tx in range(Wo // 2):
            d = x[:, :, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]                        # [B,C,4,4]
            V = np.einsum("ij,bcjl,ml->bcim", BT, d, BT)
            M = np.einsum("kcim,bcim->bkim", U, V)                                   # the 16 transform-domain GEMMs
            y[:, :, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = np.einsum("ij,bkjl,ml->bkim", AT, M, AT)
    return y


def backward_filter(w):
    """Filter of the backward-data pass: flipped in both spatial axes, channel roles swapped ([C][K][3][3])."""
    return np.ascontiguousarray(np.asarray(w)[:, :, ::-1, ::-1].transpose(1, 0, 2, 3))


def stem_conv_bwd_data(gy, w, H, W):
    """Image gradient of nn.Conv2d(Cin, K, 7, stride 2, padding 3) in K12's gather form:
    g_x[c, 2Y+py, 2X+px] = sum_k sum_{dy,dx in 0..3} g_y[k, Y-1+dy, X-1+dx] * w[k, c, py+5-2dy, px+5-2dx]   (taps in 0..6)."""
    gy = np.asarray(gy, np.float64)
    w = np.asarray(w, np.float64)
    B, K, Ho, Wo = gy.shape
    Cin = w.shape[1]
    assert H == 2 * Ho and W == 2 * Wo
    gp = np.pad(gy, ((0, 0), (0, 0), (1, 2), (1, 2)))                                # index Y-1+dy -> Y+dy
    gx = np.zeros((B, Cin, H, W))
    for py in range(2):
        for px in range(2):
            for dy in range(4):
                ky = py + 5 - 2 * dy
                if not 0 <= ky <= 6:
                    continue
                for dx in range(4):
                    kx = px + 5 - 2 * dx
                    if not 0 <= kx <= 6:
                        continue
                    gx[:, :, py::2, px::2] += np.einsum("bkhw,kc->bchw", gp[:, :, dy:dy + Ho, dx:dx + Wo], w[:, :, ky, kx])
    return gx


def bn_train_stats(x, splits=7):
    """Per-channel batch mean and biased variance as K9 computes them: shifted sums per partial (shift = the channel's
    first element), partials merged pairwise with Chan's formula."""
    x = np.asarray(x, np.float64)
    B, C = x.shape[:2]
    flat = x.transpose(1, 0, 2, 3).reshape(C, -1)
    mean, var = np.zeros(C), np.zeros(C)
    for c in range(C):
        k = flat[c, 0]
        n_t, m_t, m2_t = 0.0, 0.0, 0.0
        for part in np.array_split(flat[c], splits):
            if part.size == 0:
                continue
            d = part - k
            n, s1, s2 = float(part.size), d.sum(), (d * d).sum()
            m, m2 = k + s1 / n, s2 - s1 * s1 / n
            if n_t == 0.0:
                n_t, m_t, m2_t = n, m, m2
            else:
                tot = n_t + n
                delta, f = m - m_t, n / tot
                m_t, m2_t, n_t = m_t + delta * f, m2_t + m2 + delta * delta * n_t * f, tot
        mean[c], var[c] = m_t, m2_t / n_t
    return mean, var
