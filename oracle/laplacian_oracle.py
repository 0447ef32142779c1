"""TEST INFRASTRUCTURE -- CPU restatement (numpy, float64) of the reference's stand-alone Laplacian pyramid,
`mvae/layer_blocks.py:23-99` (laplacian_transform_split) and `:107-185` (laplacian_transform_merge, trainable=False).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it; the product path never does.

Pinned by the reference's own tests for these two functions (tests/test_layer_blocks.py:118-190), which
tests/test_laplacian_cpu.py replays: the level shapes for an (18,32,32,3) input with 3 levels, the merged shape, and the
one value test -- merge(split(x)) reproduces x within 1e-3 on the interior [:, 1:31, 1:31, :].  The Gaussian constants
are pinned by tests/test_oracle_golden.py through `gaussian_kernel` (tests/test_layer_blocks.py:9-39).  The operator
semantics restated by hand (TF 2.3.1): DepthwiseConv2D(padding="same") = zero padding; MaxPool2D(pool 1, stride 2,
"valid") = x[:, ::2, ::2]; UpSampling2D(2, "bilinear") = tf.image.resize with half-pixel centres and edge clamp."""
import numpy as np


def gaussian_kernel(size, nsig):
    """layer_blocks.py:980-1002."""
    k1 = [np.linspace(-abs(nsig[i]), abs(nsig[i]), size[i], endpoint=True) for i in range(2)]
    x, y = np.meshgrid(k1[0], k1[1])
    g = np.exp(-((np.sqrt(x * x + y * y)) ** 2 / 2.0))
    return g / g.sum()


def gaussian_filter(x, kernel_size=(3, 3), xy_max=(1, 1)):
    """gaussian_filter_block, layer_blocks.py:1008-1050: depthwise, fixed weights, SAME (zero) padding, stride 1."""
    g = gaussian_kernel(kernel_size, xy_max)
    kh, kw = g.shape
    ph, pw = kh // 2, kw // 2
    xp = np.pad(x, ((0, 0), (ph, ph), (pw, pw), (0, 0)))
    out = np.zeros_like(x, dtype=np.float64)
    h, w = x.shape[1], x.shape[2]
    for a in range(kh):
        for e in range(kw):
            out += g[a, e] * xp[:, a:a + h, e:e + w, :]
    return out


def upsample2_bilinear(x):
    """keras.layers.UpSampling2D(size=(2,2), interpolation="bilinear") (layer_blocks.py:68-71, 141-144)."""
    def axis(v, ax):
        n = v.shape[ax]
        src = (np.arange(2 * n) + 0.5) / 2.0 - 0.5
        i0 = np.floor(src).astype(np.int64)
        f = src - i0
        lo, hi = np.clip(i0, 0, n - 1), np.clip(i0 + 1, 0, n - 1)
        shape = [1] * v.ndim
        shape[ax] = 2 * n
        f = f.reshape(shape)
        return np.take(v, lo, axis=ax) * (1.0 - f) + np.take(v, hi, axis=ax) * f
    return axis(axis(np.asarray(x, np.float64), 1), 2)


def laplacian_split(x, levels, min_value=0.0, max_value=255.0, gaussian_xy_max=(1, 1), gaussian_kernel_size=(3, 3)):
    """layer_blocks.py:23-99: list of `levels` arrays, finest first."""
    n = 2.0 * (np.asarray(x, np.float64) - min_value) / (max_value - min_value) - 1.0      # _normalize, :36-41
    out = []
    for i in range(levels):
        if i == levels - 1:
            out.append(n)
        else:
            f = gaussian_filter(n, gaussian_kernel_size, gaussian_xy_max)                   # :53-58
            down = f[:, ::2, ::2, :]                                                        # MaxPool2D(1, stride 2), :61-65
            out.append(n - upsample2_bilinear(down))                                        # :68-73
            n = down
    return out


def laplacian_merge(levels_in, min_value=0.0, max_value=255.0):
    """layer_blocks.py:107-185, trainable=False."""
    out = None
    for i in range(len(levels_in) - 1, -1, -1):
        x = np.asarray(levels_in[i], np.float64)
        out = x if out is None else upsample2_bilinear(out) + x                             # :137-171
    y = (out + 1.0) * (max_value - min_value) / 2.0 + min_value                             # _denormalize, :123-131
    return np.clip(y, min_value, max_value)


def _conv_same(x, w_hwio, b=None):
    """keras Conv2D(strides 1, padding 'same'), kernel HWIO, odd kernel sizes (zero padding)."""
    kh, kw, ci, co = w_hwio.shape
    ph, pw = kh // 2, kw // 2
    xp = np.pad(np.asarray(x, np.float64), ((0, 0), (ph, ph), (pw, pw), (0, 0)))
    h, w = x.shape[1], x.shape[2]
    out = np.zeros(x.shape[:3] + (co,), np.float64)
    for a in range(kh):
        for e in range(kw):
            out += xp[:, a:a + h, e:e + w, :] @ np.asarray(w_hwio[a, e], np.float64)
    return out if b is None else out + np.asarray(b, np.float64)


def laplacian_merge_mix(levels_in, weights, min_value=0.0, max_value=255.0):
    """layer_blocks.py:107-185 with trainable=True: weights[i] = {'mix.w' [3,3,2C,F], 'mix.b' [F], 'retarget.w'
    [1,1,F,C]} for level i < levels - 1 (activation relu, the retargeting conv has no bias and a tanh)."""
    out = None
    for i in range(len(levels_in) - 1, -1, -1):
        x = np.asarray(levels_in[i], np.float64)
        if out is None:
            out = x
            continue
        cat = np.concatenate([upsample2_bilinear(out), x], axis=-1)                         # :141-150
        hid = np.maximum(_conv_same(cat, weights[i]["mix.w"], weights[i]["mix.b"]), 0.0)     # :151-158
        out = np.tanh(_conv_same(hid, weights[i]["retarget.w"])) + x                         # :159-171
    y = (out + 1.0) * (max_value - min_value) / 2.0 + min_value
    return np.clip(y, min_value, max_value)
