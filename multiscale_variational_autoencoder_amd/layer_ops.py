"""The rest of the reference's block library on the HIP path: `attention_block`,
`self_attention_block` (mvae/layer_blocks.py:654-783), `attenuate_activation`, `excite_inhibit_spatial_mask_block`,
`excite_inhibit_channel_mask_block`, `excite_inhibit_block` (:191-412), and the variants of `resnet_block` /
`mobilenetV2_block` the fused entry points do not cover (strides != (1, 1) with its MaxPooling skip path, :847-853;
`use_batchnorm=True`, :521-537, 884-886).

The reference builds these blocks by composing Keras layers in Python; here the same layers are the stateless device
operators of libmvae_hip.so (`include/mvae_hip.h`, "layer operators": convolution, depthwise convolution, activations,
channel scaling, global / windowed max pooling, BatchNormalization, the attention core), composed in Python in the same
order, forward and backward.  Every number is computed by a HIP kernel on `cuda:<device>`; torch only allocates the
buffers.  There is no CPU fallback.

A block object owns its weights (glorot_normal kernels, zero biases, BatchNorm gamma 1 / beta 0 / moving mean 0 / moving
variance 1 -- what Keras creates), `forward(x)` returns NumPy NHWC and keeps what `backward(dy)` needs; `backward` returns
(dx, {name: gradient}) for the loss sum(y * dy)."""
import ctypes as C
from collections import OrderedDict

import numpy as np

from . import _abi
from .initializers import truncated_normal

ACT = {"linear": 0, "relu": 1, "sigmoid": 2, "tanh": 3, "attenuate": 4}
DEFAULT_ATTENUATION_MULTIPLIER = 4.0       # layer_blocks.py:14
BN_EPS, BN_MOMENTUM = 1e-3, 0.99           # keras.layers.BatchNormalization() defaults
_TRUNC_STD = 0.87962566103423978


def same_out(n, s):
    return -(-int(n) // int(s))


class DeviceOps:
    """Thin typed wrappers of the layer-operator entry points; tensors are torch float32 CUDA tensors (int32 for indices)."""

    def __init__(self, device=0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("the HIP path needs a GPU (cuda:%d); there is no CPU fallback" % device)
        self.torch = torch
        self.device = int(device)
        self.dev = torch.device("cuda", self.device)
        self.lib = _abi.load_library()

    # -- plumbing
    def _s(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr() if t is not None else 0)

    def _ok(self, rc, what):
        if rc != _abi.MVAE_OK:
            raise RuntimeError("%s failed (%d)" % (what, rc))

    def put(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.dev)

    def empty(self, *shape, dtype=None):
        return self.torch.empty(shape, dtype=dtype or self.torch.float32, device=self.dev)

    def zeros(self, *shape):
        return self.torch.zeros(shape, dtype=self.torch.float32, device=self.dev)

    # -- operators
    def conv(self, x, w, b, strides=(1, 1), relu=False):
        B, H, W, Cc = x.shape
        kh, kw, _, Fn = w.shape
        y = self.empty(B, same_out(H, strides[0]), same_out(W, strides[1]), Fn)
        self._ok(self.lib.mvae_conv2d_forward(self.device, self._p(x), B, H, W, Cc, self._p(w), self._p(b), Fn, kh, kw,
                                              strides[0], strides[1], 1 if relu else 0, self._p(y), self._s()), "mvae_conv2d_forward")
        return y

    def conv_bwd(self, x, dpre, w, strides=(1, 1), need_dx=True):
        B, H, W, Cc = x.shape
        kh, kw, _, Fn = w.shape
        dx = self.empty(*x.shape) if need_dx else None
        dw, db = self.zeros(*w.shape), self.zeros(Fn)
        self._ok(self.lib.mvae_conv2d_backward(self.device, self._p(x), self._p(dpre), B, H, W, Cc, self._p(w), Fn, kh, kw,
                                               strides[0], strides[1], self._p(dx), self._p(dw), self._p(db), self._s()),
                 "mvae_conv2d_backward")
        return dx, dw, db

    def dense(self, v, w, b):                                  # Dense = the 1x1 convolution of a [B,1,1,in] tensor
        return self.conv(v.view(v.shape[0], 1, 1, v.shape[1]), w.view(1, 1, *w.shape), b).view(v.shape[0], w.shape[1])

    def dense_bwd(self, v, dpre, w):
        dx, dw, db = self.conv_bwd(v.view(v.shape[0], 1, 1, v.shape[1]), dpre.view(dpre.shape[0], 1, 1, dpre.shape[1]),
                                   w.view(1, 1, *w.shape))
        return dx.view(v.shape), dw.view(w.shape), db

    def dw(self, x, w, b):
        B, H, W, Cc = x.shape
        y = self.empty(*x.shape)
        self._ok(self.lib.mvae_depthwise3x3_forward(self.device, self._p(x), B, H, W, Cc, self._p(w), self._p(b), self._p(y),
                                                    self._s()), "mvae_depthwise3x3_forward")
        return y

    def dw_bwd(self, x, y, dy, w):
        B, H, W, Cc = x.shape
        dx, dwg, db, work = self.empty(*x.shape), self.zeros(*w.shape), self.zeros(Cc), self.empty(x.numel())
        self._ok(self.lib.mvae_depthwise3x3_backward(self.device, self._p(x), self._p(y), self._p(dy), B, H, W, Cc, self._p(w),
                                                     self._p(dx), self._p(dwg), self._p(db), self._p(work), self._s()),
                 "mvae_depthwise3x3_backward")
        return dx, dwg, db

    def act(self, name, x, param=0.0):
        if name == "linear":
            return x
        y = self.empty(*x.shape)
        self._ok(self.lib.mvae_activation_forward(self.device, ACT[name], self._p(x), self._p(y), x.numel(), float(param),
                                                  self._s()), "mvae_activation_forward")
        return y

    def act_bwd(self, name, y, dy, param=0.0):
        if name == "linear":
            return dy
        dx = self.empty(*y.shape)
        self._ok(self.lib.mvae_activation_backward(self.device, ACT[name], self._p(y), self._p(dy), self._p(dx), y.numel(),
                                                   float(param), self._s()), "mvae_activation_backward")
        return dx

    def elt(self, op, a, b):
        out = self.empty(*a.shape)
        self._ok(self.lib.mvae_eltwise(self.device, {"add": 0, "sub": 1, "mul": 2}[op], self._p(a), self._p(b), self._p(out),
                                       a.numel(), self._s()), "mvae_eltwise")
        return out

    def scale_ch(self, x, m):
        B, H, W, Cc = x.shape
        y = self.empty(*x.shape)
        self._ok(self.lib.mvae_scale_channels_forward(self.device, self._p(x), self._p(m), self._p(y), B, H * W, Cc, self._s()),
                 "mvae_scale_channels_forward")
        return y

    def scale_ch_bwd(self, x, m, dy):
        B, H, W, Cc = x.shape
        dx, dm = self.empty(*x.shape), self.empty(B, Cc)
        self._ok(self.lib.mvae_scale_channels_backward(self.device, self._p(x), self._p(m), self._p(dy), self._p(dx), self._p(dm),
                                                       B, H * W, Cc, self._s()), "mvae_scale_channels_backward")
        return dx, dm

    def gmax(self, x):
        B, H, W, Cc = x.shape
        y, idx = self.empty(B, Cc), self.empty(B, Cc, dtype=self.torch.int32)
        self._ok(self.lib.mvae_global_maxpool_forward(self.device, self._p(x), self._p(y), self._p(idx), B, H * W, Cc, self._s()),
                 "mvae_global_maxpool_forward")
        return y, idx

    def gmax_bwd(self, dy, idx, shape):
        B, H, W, Cc = shape
        dx = self.empty(*shape)
        self._ok(self.lib.mvae_global_maxpool_backward(self.device, self._p(dy), self._p(idx), self._p(dx), B, H * W, Cc,
                                                       self._s()), "mvae_global_maxpool_backward")
        return dx

    def maxpool(self, x, pool, strides):
        B, H, W, Cc = x.shape
        oh, ow = same_out(H, strides[0]), same_out(W, strides[1])
        y, idx = self.empty(B, oh, ow, Cc), self.empty(B, oh, ow, Cc, dtype=self.torch.int32)
        self._ok(self.lib.mvae_maxpool_same_forward(self.device, self._p(x), B, H, W, Cc, pool[0], pool[1], strides[0], strides[1],
                                                    self._p(y), self._p(idx), self._s()), "mvae_maxpool_same_forward")
        return y, idx

    def maxpool_bwd(self, dy, idx, shape, pool, strides):
        B, H, W, Cc = shape
        dx = self.empty(*shape)
        self._ok(self.lib.mvae_maxpool_same_backward(self.device, self._p(dy), self._p(idx), B, H, W, Cc, pool[0], pool[1],
                                                     strides[0], strides[1], self._p(dx), self._s()), "mvae_maxpool_same_backward")
        return dx

    def bn(self, x, gamma, beta, mov_mean, mov_var, training):
        Cc = x.shape[-1]
        M = x.numel() // Cc
        mean, invstd, var, y = self.empty(Cc), self.empty(Cc), self.empty(Cc), self.empty(*x.shape)
        self._ok(self.lib.mvae_batchnorm_forward(self.device, self._p(x), M, Cc, self._p(gamma), self._p(beta), BN_EPS,
                                                 1 if training else 0, self._p(mov_mean), self._p(mov_var), self._p(mean),
                                                 self._p(invstd), self._p(var), self._p(y), self._s()), "mvae_batchnorm_forward")
        return y, mean, invstd, var

    def bn_bwd(self, x, dy, gamma, mean, invstd, training):
        Cc = x.shape[-1]
        M = x.numel() // Cc
        dx, dg, db, work = self.empty(*x.shape), self.zeros(Cc), self.zeros(Cc), self.empty(2 * Cc)
        self._ok(self.lib.mvae_batchnorm_backward(self.device, self._p(x), self._p(dy), M, Cc, self._p(gamma), self._p(mean),
                                                  self._p(invstd), 1 if training else 0, self._p(dx), self._p(dg), self._p(db),
                                                  self._p(work), self._s()), "mvae_batchnorm_backward")
        return dx, dg, db

    def attention_core(self, th, ph, g):
        B, H, W, Fn = th.shape
        scores, out = self.empty(B, Fn, Fn), self.empty(B, Fn, H * W)
        self._ok(self.lib.mvae_attention_core_forward(self.device, self._p(th), self._p(ph), self._p(g), B, H * W, Fn,
                                                      self._p(scores), self._p(out), self._s()), "mvae_attention_core_forward")
        return scores, out

    def attention_core_bwd(self, th, ph, g, scores, dout):
        B, H, W, Fn = th.shape
        dth, dph, dg, work = self.empty(*th.shape), self.empty(*th.shape), self.empty(*th.shape), self.empty(B * Fn * Fn)
        self._ok(self.lib.mvae_attention_core_backward(self.device, self._p(th), self._p(ph), self._p(g), self._p(scores),
                                                       self._p(dout), B, H * W, Fn, self._p(dth), self._p(dph), self._p(dg),
                                                       self._p(work), self._s()), "mvae_attention_core_backward")
        return dth, dph, dg


def _shapes(kind, channels, filters, kernel_size, **kw):
    """name -> shape, creation order (mirrors oracle/blocks_oracle.py:layer_param_shapes; kept here so that nothing under the
    package imports the oracle)."""
    kh, kw_ = int(kernel_size[0]), int(kernel_size[1])
    c, f = int(channels), int(filters)
    P = OrderedDict()

    def conv(name, k, ci, co):
        P[name + ".w"] = (k[0], k[1], ci, co); P[name + ".b"] = (co,)

    def bn(name, n):
        for t in ("gamma", "beta", "mean", "var"):
            P[name + "." + t] = (n,)
    k = (kh, kw_)
    if kind in ("attention", "self_attention"):
        for n in ("theta", "phi", "g"):
            conv(n, k, c, f)
        if kind == "self_attention":
            conv("result", k, f, c)
    elif kind == "spatial_mask":
        ch = 1 if kw.get("flatten") else c
        for tag in ("e", "i"):
            conv("smask." + tag + "0", k, c, f)
            conv("smask." + tag + "1", (1, 1), f, ch)
    elif kind == "channel_mask":
        if kw.get("shared", True):
            conv("cmask.conv", k, c, f)
        else:
            conv("cmask.conv_e", k, c, f); conv("cmask.conv_i", k, c, f)
        for tag in ("de", "di"):
            P["cmask." + tag + ".w"] = (f, c); P["cmask." + tag + ".b"] = (c,)
    elif kind == "excite_inhibit":
        P.update(_shapes("spatial_mask", c, f, k))
        P.update(_shapes("channel_mask", c, f, k))
        conv("conv0", k, c, f)
        conv("conv1", (1, 1), f, c)
    elif kind == "resnet":
        conv("conv0", k, c, f); conv("conv1", k, f, f)
        if c != f:
            conv("skip", (1, 1), c, f)
        if kw.get("use_batchnorm"):
            bn("batchnorm", f)
    elif kind == "mnv2":
        conv("conv0", (1, 1), c, f)
        P["conv1.w"] = (3, 3, f, 1); P["conv1.b"] = (f,)
        if kw.get("use_batchnorm"):
            bn("batchnorm0", f)
        conv("conv2", (1, 1), f, c)
        if kw.get("use_batchnorm"):
            bn("batchnorm1", c)
    else:
        raise ValueError(kind)
    return P


class LayerBlock:
    """Common part: weights, input checks, gradient bookkeeping."""
    kind = None

    def __init__(self, input_dims, filters, kernel_size, name, seed=42, device=0, **kw):
        if input_dims is None:
            raise ValueError("input_layer cannot be empty")
        if len(input_dims) != 3:
            raise ValueError("works only on 4d tensors")
        if filters <= 0:
            raise ValueError("Filters should be > 0")
        self.name = name
        self.input_dims = tuple(int(d) for d in input_dims)
        self.filters = int(filters)
        self.kernel_size = (int(kernel_size[0]), int(kernel_size[1]))
        self._device = int(device)
        self._lib = _abi.load_library()          # fails loudly when the HIP library is missing
        self._kw = kw
        rng = np.random.default_rng(seed)
        self._weights = OrderedDict()
        for k, shp in _shapes(self.kind, self.input_dims[2], self.filters, self.kernel_size, **kw).items():
            if k.endswith((".b", ".beta", ".mean")):
                self._weights[k] = np.zeros(shp, np.float32)
            elif k.endswith((".gamma", ".var")):
                self._weights[k] = np.ones(shp, np.float32)
            elif len(shp) == 2:                   # Dense kernel (in, out)
                self._weights[k] = truncated_normal(rng, shp, np.sqrt(2.0 / (shp[0] + shp[1])) / _TRUNC_STD)
            else:
                kh, kw_, ci, co = shp
                self._weights[k] = truncated_normal(rng, shp, np.sqrt(2.0 / (kh * kw_ * ci + kh * kw_ * co)) / _TRUNC_STD)
        self._ops = None
        self._saved = None

    # -- Keras-like surface
    def get_weights(self):
        return OrderedDict((k, v.copy()) for k, v in self._weights.items())

    def set_weights(self, weights):
        for k, v in self._weights.items():
            a = np.ascontiguousarray(np.asarray(weights[k], np.float32))
            if a.shape != v.shape:
                raise ValueError("%s has shape %s, expected %s" % (k, a.shape, v.shape))
            self._weights[k] = a

    def predict(self, x, batch_size=None):
        return self.forward(x, training=False)

    def __call__(self, x, training=False):
        return self.forward(x, training=training)

    # -- helpers
    def _begin(self, x):
        if self._ops is None:
            self._ops = DeviceOps(self._device)
        x = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
        if x.ndim != 4 or tuple(x.shape[1:]) != self.input_dims:
            raise ValueError("expected input of shape [B, %d, %d, %d]" % self.input_dims)
        o = self._ops
        return o, o.put(x), OrderedDict((k, o.put(v)) for k, v in self._weights.items())

    def _out(self, t):
        self._ops.torch.cuda.synchronize(self._ops.dev)
        return t.cpu().numpy()

    def _need_saved(self):
        if self._saved is None:
            raise RuntimeError("backward() needs a preceding forward()")
        return self._ops, self._saved

    def _finish(self, dx, grads):
        o = self._ops
        o.torch.cuda.synchronize(o.dev)
        G = OrderedDict()
        for k in self._weights:
            if k.endswith((".mean", ".var")):
                continue
            G[k] = grads[k].cpu().numpy().reshape(self._weights[k].shape) if k in grads else np.zeros_like(self._weights[k])
        return dx.cpu().numpy(), G

    def _bn_update(self, prefix, mean, var, M):
        """moving = moving * momentum + batch * (1 - momentum); Keras' fused 4-D kernel feeds the UNBIASED batch variance."""
        m = mean.cpu().numpy()
        v = var.cpu().numpy() * (M / max(M - 1.0, 1.0))
        self._weights[prefix + ".mean"] = (self._weights[prefix + ".mean"] * BN_MOMENTUM + m * (1 - BN_MOMENTUM)).astype(np.float32)
        self._weights[prefix + ".var"] = (self._weights[prefix + ".var"] * BN_MOMENTUM + v * (1 - BN_MOMENTUM)).astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------------
class AttentionBlock(LayerBlock):
    """attention_block (layer_blocks.py:654-728).  As the reference writes it this is a CHANNEL attention: the F x F map
    softmax_j(sum_p theta[p, i] phi[p, j]) mixes g's channels, and the (F, HW) result is reshaped -- not transposed -- to
    (H, W, F) (see `mvae_attention_core_forward`)."""
    kind = "attention"

    def __init__(self, input_dims, filters=32, kernel_size=(1, 1), activation="linear", name="attention_", seed=42, device=0):
        if activation not in ("linear", "relu", "sigmoid", "tanh"):
            raise ValueError("activation must be one of linear / relu / sigmoid / tanh")
        self.activation = activation
        super().__init__(input_dims, filters, kernel_size, name, seed, device)

    def _fwd(self, o, x, w):
        acts = {}
        for n in ("theta", "phi", "g"):
            acts[n] = o.act(self.activation, o.conv(x, w[n + ".w"], w[n + ".b"]))
        scores, out = o.attention_core(acts["theta"], acts["phi"], acts["g"])
        B, H, W, _ = x.shape
        return out.view(B, H, W, self.filters), dict(x=x, w=w, acts=acts, scores=scores)

    def _bwd(self, o, sv, dy):
        B, H, W, _ = sv["x"].shape
        a = sv["acts"]
        d = dict(zip(("theta", "phi", "g"), o.attention_core_bwd(a["theta"], a["phi"], a["g"], sv["scores"],
                                                                  dy.reshape(B, self.filters, H * W))))
        grads, dx = {}, None
        for n in ("theta", "phi", "g"):
            dpre = o.act_bwd(self.activation, a[n], d[n])
            dxi, grads[n + ".w"], grads[n + ".b"] = o.conv_bwd(sv["x"], dpre, sv["w"][n + ".w"])
            dx = dxi if dx is None else o.elt("add", dx, dxi)
        return dx, grads

    def forward(self, x, training=False):
        o, xd, w = self._begin(x)
        y, self._saved = self._fwd(o, xd, w)
        return self._out(y)

    def backward(self, dy):
        o, sv = self._need_saved()
        dx, grads = self._bwd(o, sv, o.put(np.asarray(dy, np.float32)))
        return self._finish(dx, grads)


class SelfAttentionBlock(AttentionBlock):
    """self_attention_block (layer_blocks.py:734-783): attention_block -> Conv2D back to the input's channels -> Add."""
    kind = "self_attention"

    def __init__(self, input_dims, filters=32, kernel_size=(1, 1), activation="linear", name="self_attention_", seed=42, device=0):
        super().__init__(input_dims, filters, kernel_size, activation, name, seed, device)

    def forward(self, x, training=False):
        o, xd, w = self._begin(x)
        a, sv = self._fwd(o, xd, w)
        a = a.contiguous()
        r = o.act(self.activation, o.conv(a, w["result.w"], w["result.b"]))
        sv.update(att=a, r=r)
        self._saved = sv
        return self._out(o.elt("add", r, xd))

    def backward(self, dy):
        o, sv = self._need_saved()
        dyd = o.put(np.asarray(dy, np.float32))
        dpre = o.act_bwd(self.activation, sv["r"], dyd)
        datt, gw, gb = o.conv_bwd(sv["att"], dpre, sv["w"]["result.w"])
        dx, grads = self._bwd(o, sv, datt)
        grads["result.w"], grads["result.b"] = gw, gb
        return self._finish(o.elt("add", dx, dyd), grads)


# ---------------------------------------------------------------------------------------------------------------------
def _spatial_mask_fwd(o, x, w, first, second, mult, prefix="smask."):
    br = {}
    for tag in ("e", "i"):
        h = o.act(first, o.conv(x, w[prefix + tag + "0.w"], w[prefix + tag + "0.b"]))
        z = o.act(second, o.conv(h, w[prefix + tag + "1.w"], w[prefix + tag + "1.b"]))
        br[tag] = (h, z)
    m = o.act("attenuate", o.elt("sub", br["e"][1], br["i"][1]), mult)
    return m, dict(br=br, m=m)


def _spatial_mask_bwd(o, x, w, sv, dm, first, second, mult, prefix="smask."):
    dd = o.act_bwd("attenuate", sv["m"], dm, mult)                  # gradient at (excite - inhibit)
    grads, dx = {}, None
    for tag, sign in (("e", 1.0), ("i", -1.0)):
        h, z = sv["br"][tag]
        dz = dd if sign > 0 else o.elt("sub", o.torch.zeros_like(dd), dd)
        dpre1 = o.act_bwd(second, z, dz)
        dh, grads[prefix + tag + "1.w"], grads[prefix + tag + "1.b"] = o.conv_bwd(h, dpre1, w[prefix + tag + "1.w"])
        dpre0 = o.act_bwd(first, h, dh)
        dxi, grads[prefix + tag + "0.w"], grads[prefix + tag + "0.b"] = o.conv_bwd(x, dpre0, w[prefix + tag + "0.w"])
        dx = dxi if dx is None else o.elt("add", dx, dxi)
    return dx, grads


def _channel_mask_fwd(o, x, w, shared, first, second, mult, prefix="cmask."):
    trunks = {}
    for tag in (("conv",) if shared else ("conv_e", "conv_i")):
        h = o.act(first, o.conv(x, w[prefix + tag + ".w"], w[prefix + tag + ".b"]))
        v, idx = o.gmax(h)
        trunks[tag] = (h, v, idx)
    ve = trunks["conv" if shared else "conv_e"][1]
    vi = trunks["conv" if shared else "conv_i"][1]
    ze = o.act(second, o.dense(ve, w[prefix + "de.w"], w[prefix + "de.b"]))
    zi = o.act(second, o.dense(vi, w[prefix + "di.w"], w[prefix + "di.b"]))
    m = o.act("attenuate", o.elt("sub", ze, zi), mult)
    return m, dict(trunks=trunks, ze=ze, zi=zi, m=m)


def _channel_mask_bwd(o, x, w, sv, dm, shared, first, second, mult, prefix="cmask."):
    dd = o.act_bwd("attenuate", sv["m"], dm, mult)
    grads = {}
    neg = o.elt("sub", o.torch.zeros_like(dd), dd)
    te, ti = ("conv", "conv") if shared else ("conv_e", "conv_i")
    dve, grads[prefix + "de.w"], grads[prefix + "de.b"] = o.dense_bwd(sv["trunks"][te][1], o.act_bwd(second, sv["ze"], dd),
                                                                        w[prefix + "de.w"])
    dvi, grads[prefix + "di.w"], grads[prefix + "di.b"] = o.dense_bwd(sv["trunks"][ti][1], o.act_bwd(second, sv["zi"], neg),
                                                                        w[prefix + "di.w"])
    dx = None
    for tag, dv in ((("conv", o.elt("add", dve, dvi)),) if shared else (("conv_e", dve), ("conv_i", dvi))):
        h, _, idx = sv["trunks"][tag]
        dh = o.gmax_bwd(dv, idx, tuple(h.shape))
        dxi, grads[prefix + tag + ".w"], grads[prefix + tag + ".b"] = o.conv_bwd(x, o.act_bwd(first, h, dh), w[prefix + tag + ".w"])
        dx = dxi if dx is None else o.elt("add", dx, dxi)
    return dx, grads


class ExciteInhibitSpatialMask(LayerBlock):
    """excite_inhibit_spatial_mask_block (layer_blocks.py:204-271), add_batchnorm=False."""
    kind = "spatial_mask"

    def __init__(self, input_dims, filters=32, kernel_size=(3, 3), flatten=False, first_level_activation="relu",
                 second_level_activation="sigmoid", multiplier=DEFAULT_ATTENUATION_MULTIPLIER, name="spatial_mask_", seed=42,
                 device=0):
        self.first, self.second, self.multiplier = first_level_activation, second_level_activation, float(multiplier)
        super().__init__(input_dims, filters, kernel_size, name, seed, device, flatten=bool(flatten))

    def forward(self, x, training=False):
        o, xd, w = self._begin(x)
        m, sv = _spatial_mask_fwd(o, xd, w, self.first, self.second, self.multiplier)
        self._saved = dict(x=xd, w=w, sm=sv)
        return self._out(m)

    def backward(self, dy):
        o, sv = self._need_saved()
        dx, grads = _spatial_mask_bwd(o, sv["x"], sv["w"], sv["sm"], o.put(np.asarray(dy, np.float32)), self.first, self.second,
                                      self.multiplier)
        return self._finish(dx, grads)


class ExciteInhibitChannelMask(LayerBlock):
    """excite_inhibit_channel_mask_block (layer_blocks.py:277-350), add_batchnorm=False; output [B, C]."""
    kind = "channel_mask"

    def __init__(self, input_dims, filters=32, kernel_size=(3, 3), shared=True, first_level_activation="linear",
                 second_level_activation="sigmoid", multiplier=DEFAULT_ATTENUATION_MULTIPLIER, name="channel_mask_", seed=42,
                 device=0):
        self.shared, self.first, self.second = bool(shared), first_level_activation, second_level_activation
        self.multiplier = float(multiplier)
        super().__init__(input_dims, filters, kernel_size, name, seed, device, shared=bool(shared))

    def forward(self, x, training=False):
        o, xd, w = self._begin(x)
        m, sv = _channel_mask_fwd(o, xd, w, self.shared, self.first, self.second, self.multiplier)
        self._saved = dict(x=xd, w=w, cm=sv)
        return self._out(m)

    def backward(self, dy):
        o, sv = self._need_saved()
        dx, grads = _channel_mask_bwd(o, sv["x"], sv["w"], sv["cm"], o.put(np.asarray(dy, np.float32)), self.shared, self.first,
                                      self.second, self.multiplier)
        return self._finish(dx, grads)


class ExciteInhibitBlock(LayerBlock):
    """excite_inhibit_block (layer_blocks.py:356-412): both masks with their defaults, x * spatial * channel ->
    Conv2D(filters, k, relu) -> Conv2D(channels, 1x1, linear) -> * spatial."""
    kind = "excite_inhibit"

    def __init__(self, input_dims, filters=32, kernel_size=(3, 3), name="excite_inhibit_", seed=42, device=0):
        super().__init__(input_dims, filters, kernel_size, name, seed, device)

    def forward(self, x, training=False):
        o, xd, w = self._begin(x)
        m = DEFAULT_ATTENUATION_MULTIPLIER
        sm, ssv = _spatial_mask_fwd(o, xd, w, "relu", "sigmoid", m)
        cm, csv = _channel_mask_fwd(o, xd, w, True, "linear", "sigmoid", m)
        xs = o.elt("mul", xd, sm)
        masked = o.scale_ch(xs, cm)
        h = o.conv(masked, w["conv0.w"], w["conv0.b"], relu=True)
        z = o.conv(h, w["conv1.w"], w["conv1.b"])
        self._saved = dict(x=xd, w=w, sm=sm, ssv=ssv, cm=cm, csv=csv, xs=xs, masked=masked, h=h, z=z)
        return self._out(o.elt("mul", sm, z))

    def backward(self, dy):
        o, sv = self._need_saved()
        w, m = sv["w"], DEFAULT_ATTENUATION_MULTIPLIER
        dyd = o.put(np.asarray(dy, np.float32))
        grads = {}
        dz = o.elt("mul", dyd, sv["sm"])
        dsm = o.elt("mul", dyd, sv["z"])
        dh, grads["conv1.w"], grads["conv1.b"] = o.conv_bwd(sv["h"], dz, w["conv1.w"])
        dmasked, grads["conv0.w"], grads["conv0.b"] = o.conv_bwd(sv["masked"], o.act_bwd("relu", sv["h"], dh), w["conv0.w"])
        dxs, dcm = o.scale_ch_bwd(sv["xs"], sv["cm"], dmasked)
        dx = o.elt("mul", dxs, sv["sm"])
        dsm = o.elt("add", dsm, o.elt("mul", dxs, sv["x"]))
        dxs_, gs = _spatial_mask_bwd(o, sv["x"], w, sv["ssv"], dsm, "relu", "sigmoid", m)
        dxc_, gc = _channel_mask_bwd(o, sv["x"], w, sv["csv"], dcm, True, "linear", "sigmoid", m)
        grads.update(gs); grads.update(gc)
        return self._finish(o.elt("add", o.elt("add", dx, dxs_), dxc_), grads)


# ---------------------------------------------------------------------------------------------------------------------
class ResnetBlockGeneral(LayerBlock):
    """resnet_block with strides and / or use_batchnorm (layer_blocks.py:830-886); dropout 0."""
    kind = "resnet"

    def __init__(self, input_dims, filters=32, kernel_size=(3, 3), strides=(1, 1), activation="relu", use_batchnorm=False,
                 name="resnet_", seed=42, device=0):
        if activation not in ("relu", "linear", "sigmoid", "tanh"):
            raise ValueError("activation must be one of relu / linear / sigmoid / tanh")
        self.strides = (int(strides[0]), int(strides[1]))
        self.activation, self.use_batchnorm = activation, bool(use_batchnorm)
        super().__init__(input_dims, filters, kernel_size, name, seed, device, use_batchnorm=bool(use_batchnorm))

    def forward(self, x, training=False):
        o, xd, w = self._begin(x)
        x0 = o.act(self.activation, o.conv(xd, w["conv0.w"], w["conv0.b"]))
        h = o.conv(x0, w["conv1.w"], w["conv1.b"], self.strides)
        sk, pidx = xd, None
        if self.strides != (1, 1):
            sk, pidx = o.maxpool(xd, tuple(s + 1 for s in self.strides), self.strides)
        pooled = sk
        if "skip.w" in w:
            sk = o.conv(sk, w["skip.w"], w["skip.b"])
        y = o.act(self.activation, o.elt("add", h, sk))
        sv = dict(x=xd, w=w, x0=x0, pooled=pooled, pidx=pidx, y=y, training=training)
        out = y
        if self.use_batchnorm:
            out, mean, invstd, var = o.bn(y, w["batchnorm.gamma"], w["batchnorm.beta"], w["batchnorm.mean"], w["batchnorm.var"], training)
            sv.update(mean=mean, invstd=invstd)
            if training:
                self._bn_update("batchnorm", mean, var, y.numel() / y.shape[-1])
        self._saved = sv
        return self._out(out)

    def backward(self, dy):
        o, sv = self._need_saved()
        w = sv["w"]
        d = o.put(np.asarray(dy, np.float32))
        grads = {}
        if self.use_batchnorm:
            d, grads["batchnorm.gamma"], grads["batchnorm.beta"] = o.bn_bwd(sv["y"], d, w["batchnorm.gamma"], sv["mean"],
                                                                          sv["invstd"], sv["training"])
        dpre = o.act_bwd(self.activation, sv["y"], d)
        d0, grads["conv1.w"], grads["conv1.b"] = o.conv_bwd(sv["x0"], dpre, w["conv1.w"], self.strides)
        dx, grads["conv0.w"], grads["conv0.b"] = o.conv_bwd(sv["x"], o.act_bwd(self.activation, sv["x0"], d0), w["conv0.w"])
        dsk = dpre
        if "skip.w" in w:
            dsk, grads["skip.w"], grads["skip.b"] = o.conv_bwd(sv["pooled"], dpre, w["skip.w"])
        if self.strides != (1, 1):
            dsk = o.maxpool_bwd(dsk, sv["pidx"], tuple(sv["x"].shape), tuple(s + 1 for s in self.strides), self.strides)
        return self._finish(o.elt("add", dx, dsk), grads)


class MobileNetV2BlockBN(LayerBlock):
    """mobilenetV2_block with use_batchnorm=True (layer_blocks.py:500-541)."""
    kind = "mnv2"

    def __init__(self, input_dims, filters=32, name="mobilenetV2_", seed=42, device=0):
        super().__init__(input_dims, filters, (1, 1), name, seed, device, use_batchnorm=True)

    def forward(self, x, training=False):
        o, xd, w = self._begin(x)
        t0 = o.conv(xd, w["conv0.w"], w["conv0.b"])
        t1 = o.dw(t0, w["conv1.w"], w["conv1.b"])
        n0, m0, i0, v0 = o.bn(t1, w["batchnorm0.gamma"], w["batchnorm0.beta"], w["batchnorm0.mean"], w["batchnorm0.var"], training)
        u = o.conv(n0, w["conv2.w"], w["conv2.b"], relu=True)
        n1, m1, i1, v1 = o.bn(u, w["batchnorm1.gamma"], w["batchnorm1.beta"], w["batchnorm1.mean"], w["batchnorm1.var"], training)
        if training:
            self._bn_update("batchnorm0", m0, v0, t1.numel() / t1.shape[-1])
            self._bn_update("batchnorm1", m1, v1, u.numel() / u.shape[-1])
        self._saved = dict(x=xd, w=w, t0=t0, t1=t1, n0=n0, u=u, st0=(m0, i0), st1=(m1, i1), training=training)
        return self._out(o.elt("add", n1, xd))

    def backward(self, dy):
        o, sv = self._need_saved()
        w, tr = sv["w"], sv["training"]
        d = o.put(np.asarray(dy, np.float32))
        grads = {}
        du, grads["batchnorm1.gamma"], grads["batchnorm1.beta"] = o.bn_bwd(sv["u"], d, w["batchnorm1.gamma"], *sv["st1"], tr)
        dn0, grads["conv2.w"], grads["conv2.b"] = o.conv_bwd(sv["n0"], o.act_bwd("relu", sv["u"], du), w["conv2.w"])
        dt1, grads["batchnorm0.gamma"], grads["batchnorm0.beta"] = o.bn_bwd(sv["t1"], dn0, w["batchnorm0.gamma"], *sv["st0"], tr)
        dt0, grads["conv1.w"], grads["conv1.b"] = o.dw_bwd(sv["t0"], sv["t1"], dt1, w["conv1.w"])
        dx, grads["conv0.w"], grads["conv0.b"] = o.conv_bwd(sv["x"], dt0, w["conv0.w"])
        return self._finish(o.elt("add", dx, d), grads)


def attenuate_activation(x, multiplier=DEFAULT_ATTENUATION_MULTIPLIER, device=0):
    """layer_blocks.py:191-198 on the device: (tanh(x * multiplier) + 1) / 2 of a NumPy array."""
    o = DeviceOps(device)
    y = o.act("attenuate", o.put(np.asarray(x, np.float32)), multiplier)
    o.torch.cuda.synchronize(o.dev)
    return y.cpu().numpy()
