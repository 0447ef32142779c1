"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/mvae_oracle.py header).

Independent loop-level numpy restatements of the operator semantics the torch oracle relies on
(TF 2.3.1 'SAME' conv padding, Conv2DTranspose 'SAME' as the conv adjoint, half-pixel bilinear x2,
the fixed Gaussian blur).  They share no code with mvae_oracle.py (which maps onto torch.nn.functional)
and exist so that tests/test_oracle_golden.py can cross-check one restatement against the other on
small cases.  Pure-Python loops: small shapes only.
"""
import numpy as np


def same_out_pad(n, k, s):
    out = (n + s - 1) // s
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2


def conv2d_same_nhwc(x, w, b, stride):
    """x [B,H,W,Ci], w [kh,kw,Ci,Co] -> [B,ceil(H/sh),ceil(W/sw),Co]  (layer_blocks.py:946-949)."""
    B, H, W, Ci = x.shape
    kh, kw, _, Co = w.shape
    oh, pt = same_out_pad(H, kh, stride[0])
    ow, pl = same_out_pad(W, kw, stride[1])
    y = np.zeros((B, oh, ow, Co), np.float64)
    for i in range(oh):
        for j in range(ow):
            for a in range(kh):
                for c in range(kw):
                    yy, xx = i * stride[0] + a - pt, j * stride[1] + c - pl
                    if 0 <= yy < H and 0 <= xx < W:
                        y[:, i, j, :] += x[:, yy, xx, :] @ w[a, c]
    return y + (0 if b is None else b)


def conv2d_transpose_same_nhwc(x, w, b, stride):
    """x [B,h,w,Ci], w [kh,kw,Co,Ci] -> [B,h*sh,w*sw,Co]: scatter form of the adjoint of the SAME conv
    that maps (h*sh, w*sw) -> (h, w)  (layer_blocks.py:950-951)."""
    B, h, wd, Ci = x.shape
    kh, kw, Co, _ = w.shape
    H, W = h * stride[0], wd * stride[1]
    _, pt = same_out_pad(H, kh, stride[0])
    _, pl = same_out_pad(W, kw, stride[1])
    y = np.zeros((B, H, W, Co), np.float64)
    for i in range(h):
        for j in range(wd):
            for a in range(kh):
                for c in range(kw):
                    yy, xx = i * stride[0] + a - pt, j * stride[1] + c - pl
                    if 0 <= yy < H and 0 <= xx < W:
                        y[:, yy, xx, :] += x[:, i, j, :] @ w[a, c].T
    return y + (0 if b is None else b)


def upsample2_bilinear_nhwc(x):
    """keras UpSampling2D(2, 'bilinear') = tf.image.resize half-pixel centres, edge clamp
    (multiscale_vae.py:214-216): out[2i] = .25 in[i-1] + .75 in[i]; out[2i+1] = .75 in[i] + .25 in[i+1]."""
    def up1(a, axis):
        n = a.shape[axis]
        idx = np.arange(n)
        lo = np.take(a, np.maximum(idx - 1, 0), axis)
        hi = np.take(a, np.minimum(idx + 1, n - 1), axis)
        even = 0.25 * lo + 0.75 * a
        odd = 0.75 * a + 0.25 * hi
        out = np.stack([even, odd], axis + 1)
        shp = list(a.shape)
        shp[axis] = 2 * n
        return out.reshape(shp)
    return up1(up1(np.asarray(x, np.float64), 1), 2)


def gaussian_blur_nhwc(x, nsig=(2, 2)):
    """layer_blocks.py:980-1050 with zero 'same' padding."""
    g1 = [np.linspace(-abs(nsig[i]), abs(nsig[i]), 3) for i in range(2)]
    gx, gy = np.meshgrid(g1[0], g1[1])
    g = np.exp(-(gx * gx + gy * gy) / 2.0)
    g = g / g.sum()
    B, H, W, C = x.shape
    xp = np.zeros((B, H + 2, W + 2, C), np.float64)
    xp[:, 1:-1, 1:-1] = x
    y = np.zeros((B, H, W, C), np.float64)
    for a in range(3):
        for c in range(3):
            y += g[a, c] * xp[:, a:a + H, c:c + W]
    return y
