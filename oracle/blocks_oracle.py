"""TEST INFRASTRUCTURE -- CPU restatement (torch float64, autograd) of the reference's remaining block library, used only by
tests/ to check the HIP layer operators.  "Parity unpinned": the reference pins only the OUTPUT SHAPES of these blocks
(tests/test_layer_blocks.py:42-79, replayed in tests/test_layer_blocks_cpu.py); values follow the Keras 2.4.3 / TF 2.3.1
layer semantics the reference calls.

  attenuate_activation                       mvae/layer_blocks.py:191-198
  excite_inhibit_spatial_mask_block          :204-271
  excite_inhibit_channel_mask_block          :277-350
  excite_inhibit_block                       :356-412
  attention_block / self_attention_block     :654-728, :734-783
  resnet_block (strides, use_batchnorm)      :789-887
  mobilenetV2_block (use_batchnorm)          :468-550

Tensors are NCHW inside (torch), NHWC at the boundary; kernels in the Keras layouts (HWIO, Dense (in, out))."""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from oracle.mvae_oracle import conv2d_same, depthwise3x3_same, same_pads

BN_EPS = 1e-3          # keras.layers.BatchNormalization() defaults: epsilon 1e-3, momentum 0.99


def _act(name):
    return {"linear": lambda v: v, "relu": F.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}[name]


def attenuate(x, multiplier=4.0):
    """layer_blocks.py:191-198: (tanh(x * multiplier) + 1) / 2."""
    return (torch.tanh(x * multiplier) + 1.0) / 2.0


def batchnorm(x, T, prefix, training, channel_dim=1):
    """keras BatchNormalization on the channel axis: batch mean / biased variance when training, else the moving statistics
    (which a freshly built layer holds as mean 0, variance 1)."""
    shape = [1] * x.dim()
    shape[channel_dim] = -1
    dims = [d for d in range(x.dim()) if d != channel_dim]
    if training:
        mean = x.mean(dim=dims, keepdim=True)
        var = ((x - mean) ** 2).mean(dim=dims, keepdim=True)
    else:
        mean = T[prefix + ".mean"].view(shape)
        var = T[prefix + ".var"].view(shape)
    return (x - mean) / torch.sqrt(var + BN_EPS) * T[prefix + ".gamma"].view(shape) + T[prefix + ".beta"].view(shape)


def maxpool_same(x, pool, strides):
    """keras MaxPooling2D(pool, strides, 'same'): TF SAME padding (extra pad bottom / right), padded cells never win."""
    _, pt, pb = same_pads(x.shape[2], pool[0], strides[0])
    _, pl, pr = same_pads(x.shape[3], pool[1], strides[1])
    xp = F.pad(x, (pl, pr, pt, pb), value=float("-inf"))
    return F.max_pool2d(xp, kernel_size=pool, stride=strides)


# ------------------------------------------------------------------------------------------------ attention
def attention_block_t(x, T, activation="linear", kernel_size=(1, 1), prefix=""):
    """layer_blocks.py:684-728 as written.  theta_flat (B, HW, F), phi_flat permuted to (B, F, HW);
    Dot(axes=(1, 2)) contracts the PIXEL axis: S[b, i, j] = sum_p theta[b, p, i] phi[b, p, j] (F x F); Softmax() on the
    last axis; Dot(axes=(1, 2)) of (B, F, F) with g_flat (B, HW, F) contracts i: O[b, j, p] = sum_i A[b, i, j] g[b, p, i],
    shape (B, F, HW); Reshape((H, W, F)) then REINTERPRETS that buffer (no transpose)."""
    act = _act(activation)
    B, _, H, W = x.shape
    th = act(conv2d_same(x, T[prefix + "theta.w"], T[prefix + "theta.b"], (1, 1)))
    ph = act(conv2d_same(x, T[prefix + "phi.w"], T[prefix + "phi.b"], (1, 1)))
    g = act(conv2d_same(x, T[prefix + "g.w"], T[prefix + "g.b"], (1, 1)))
    Fn = th.shape[1]
    thf = th.permute(0, 2, 3, 1).reshape(B, H * W, Fn)        # Reshape((HW, F)) of the NHWC tensor
    phf = ph.permute(0, 2, 3, 1).reshape(B, H * W, Fn)
    gf = g.permute(0, 2, 3, 1).reshape(B, H * W, Fn)
    S = torch.einsum("bpi,bpj->bij", thf, phf)
    A = torch.softmax(S, dim=-1)
    O = torch.einsum("bij,bpi->bjp", A, gf)                   # (B, F, HW)
    y_nhwc = O.reshape(B, H, W, Fn)                           # Reshape(shape[1:3] + (filters,)): same memory order
    return y_nhwc.permute(0, 3, 1, 2)


def self_attention_block_t(x, T, activation="linear", kernel_size=(1, 1)):
    """layer_blocks.py:755-783: attention_block -> Conv2D(previous channels, kernel_size, activation) -> Add with the input."""
    a = attention_block_t(x, T, activation, kernel_size)
    r = _act(activation)(conv2d_same(a, T["result.w"], T["result.b"], (1, 1)))
    return r + x


# ------------------------------------------------------------------------------------------------ excite / inhibit
def spatial_mask_t(x, T, first="relu", second="sigmoid", multiplier=4.0, prefix="smask."):
    """layer_blocks.py:204-271 (add_batchnorm False): two branches Conv2D(filters, k, first) -> Conv2D(channels, 1x1, second);
    attenuate(excite - inhibit)."""
    def branch(tag):
        h = _act(first)(conv2d_same(x, T[prefix + tag + "0.w"], T[prefix + tag + "0.b"], (1, 1)))
        return _act(second)(conv2d_same(h, T[prefix + tag + "1.w"], T[prefix + tag + "1.b"], (1, 1)))
    return attenuate(branch("e") - branch("i"), multiplier)


def channel_mask_t(x, T, shared=True, first="linear", second="sigmoid", multiplier=4.0, prefix="cmask."):
    """layer_blocks.py:277-350 (add_batchnorm False): Conv2D(filters, k, first) -> GlobalMaxPool2D -> Dense(channels, second),
    one shared trunk or one per branch; attenuate(excite - inhibit): [B, C]."""
    def trunk(tag):
        h = _act(first)(conv2d_same(x, T[prefix + tag + ".w"], T[prefix + tag + ".b"], (1, 1)))
        return h.amax(dim=(2, 3))
    def dense(v, tag):
        return _act(second)(v @ T[prefix + tag + ".w"] + T[prefix + tag + ".b"])
    if shared:
        v = trunk("conv")
        return attenuate(dense(v, "de") - dense(v, "di"), multiplier)
    return attenuate(dense(trunk("conv_e"), "de") - dense(trunk("conv_i"), "di"), multiplier)


def excite_inhibit_block_t(x, T):
    """layer_blocks.py:356-412: spatial mask (relu / sigmoid) and channel mask (shared, linear / sigmoid) of the input;
    x * spatial * channel -> Conv2D(filters, k, relu) -> Conv2D(channels, 1x1, linear) -> * spatial."""
    sm = spatial_mask_t(x, T)
    cm = channel_mask_t(x, T)
    masked = x * sm * cm.view(cm.shape[0], cm.shape[1], 1, 1)
    h = F.relu(conv2d_same(masked, T["conv0.w"], T["conv0.b"], (1, 1)))
    h = conv2d_same(h, T["conv1.w"], T["conv1.b"], (1, 1))
    return sm * h


# ------------------------------------------------------------------------------------------------ resnet / mobilenetV2 variants
def resnet_block_general_t(x, T, activation="relu", strides=(1, 1), use_batchnorm=False, training=True):
    """layer_blocks.py:830-886: conv0 (stride 1, activation) -> conv1 (strides, linear); the skip input goes through
    MaxPooling2D(pool = strides + 1, 'same', strides) when strides != (1, 1) and through a 1x1 convolution when the widths
    differ; Add -> activation -> [BatchNormalization]."""
    act = _act(activation)
    h = act(conv2d_same(x, T["conv0.w"], T["conv0.b"], (1, 1)))
    h = conv2d_same(h, T["conv1.w"], T["conv1.b"], tuple(strides))
    skip = x
    if tuple(strides) != (1, 1):
        skip = maxpool_same(skip, tuple(s + 1 for s in strides), tuple(strides))
    if "skip.w" in T:
        skip = conv2d_same(skip, T["skip.w"], T["skip.b"], (1, 1))
    y = act(h + skip)
    if use_batchnorm:
        y = batchnorm(y, T, "batchnorm", training)
    return y


def mobilenetV2_block_general_t(x, T, use_batchnorm=False, training=True):
    """layer_blocks.py:500-541: conv0 1x1 linear -> depthwise 3x3 relu -> [BN] -> conv2 1x1 relu -> [BN] -> Add."""
    h = conv2d_same(x, T["conv0.w"], T["conv0.b"], (1, 1))
    h = F.relu(depthwise3x3_same(h, T["conv1.w"], T["conv1.b"]))
    if use_batchnorm:
        h = batchnorm(h, T, "batchnorm0", training)
    h = F.relu(conv2d_same(h, T["conv2.w"], T["conv2.b"], (1, 1)))
    if use_batchnorm:
        h = batchnorm(h, T, "batchnorm1", training)
    return h + x


# ------------------------------------------------------------------------------------------------ shapes + driver
def layer_param_shapes(kind, channels, filters=32, kernel_size=(3, 3), **kw):
    """name -> shape of every tensor of a block, in creation order (trainable and BatchNorm state alike)."""
    kh, kw_ = kernel_size
    c, f = int(channels), int(filters)
    P = OrderedDict()
    def conv(name, k, ci, co):
        P[name + ".w"] = (k[0], k[1], ci, co); P[name + ".b"] = (co,)
    def bn(name, n):
        for t in ("gamma", "beta", "mean", "var"):
            P[name + "." + t] = (n,)
    if kind in ("attention", "self_attention"):
        for n in ("theta", "phi", "g"):
            conv(n, kernel_size, c, f)
        if kind == "self_attention":
            conv("result", kernel_size, f, c)
    elif kind == "spatial_mask":
        ch = 1 if kw.get("flatten") else c
        for tag in ("e", "i"):
            conv("smask." + tag + "0", kernel_size, c, f)
            conv("smask." + tag + "1", (1, 1), f, ch)
    elif kind == "channel_mask":
        if kw.get("shared", True):
            conv("cmask.conv", kernel_size, c, f)
        else:
            conv("cmask.conv_e", kernel_size, c, f); conv("cmask.conv_i", kernel_size, c, f)
        for tag in ("de", "di"):
            P["cmask." + tag + ".w"] = (f, c); P["cmask." + tag + ".b"] = (c,)
    elif kind == "excite_inhibit":
        P.update(layer_param_shapes("spatial_mask", c, f, kernel_size))
        P.update(layer_param_shapes("channel_mask", c, f, kernel_size))
        conv("conv0", kernel_size, c, f)
        conv("conv1", (1, 1), f, c)
    elif kind == "resnet":
        conv("conv0", kernel_size, c, f); conv("conv1", kernel_size, f, f)
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


def layer_output_shape(kind, input_shape, filters=32, strides=(1, 1), **kw):
    """The shapes the reference's tests pin (tests/test_layer_blocks.py:42-79) and their siblings."""
    b, h, w, c = input_shape
    if kind == "attention":
        return (b, h, w, filters)
    if kind in ("self_attention", "excite_inhibit", "mnv2"):
        return (b, h, w, c)
    if kind == "spatial_mask":
        return (b, h, w, 1 if kw.get("flatten") else c)
    if kind == "channel_mask":
        return (b, c)
    if kind == "resnet":
        return (b, -(-h // strides[0]), -(-w // strides[1]), filters)
    raise ValueError(kind)


def layer_forward_backward(kind, x_nhwc, params, dy, **kw):
    """y, dx and the gradients of sum(y * dy) w.r.t. every trainable tensor (BatchNorm moving statistics excluded)."""
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    x = t(x_nhwc).permute(0, 3, 1, 2).clone().requires_grad_(True)
    state = {k for k in params if k.endswith((".mean", ".var"))}
    T = OrderedDict((k, t(v).clone().requires_grad_(k not in state)) for k, v in params.items())
    if kind == "attention":
        y = attention_block_t(x, T, kw.get("activation", "linear"))
    elif kind == "self_attention":
        y = self_attention_block_t(x, T, kw.get("activation", "linear"))
    elif kind == "spatial_mask":
        y = spatial_mask_t(x, T, kw.get("first", "relu"), kw.get("second", "sigmoid"), kw.get("multiplier", 4.0))
    elif kind == "channel_mask":
        y = channel_mask_t(x, T, kw.get("shared", True), kw.get("first", "linear"), kw.get("second", "sigmoid"),
                           kw.get("multiplier", 4.0))
    elif kind == "excite_inhibit":
        y = excite_inhibit_block_t(x, T)
    elif kind == "resnet":
        y = resnet_block_general_t(x, T, kw.get("activation", "relu"), kw.get("strides", (1, 1)), kw.get("use_batchnorm", False),
                                   kw.get("training", True))
    elif kind == "mnv2":
        y = mobilenetV2_block_general_t(x, T, kw.get("use_batchnorm", False), kw.get("training", True))
    else:
        raise ValueError(kind)
    dyt = t(dy)
    if y.dim() == 4:
        dyt = dyt.permute(0, 3, 1, 2)
    names = [k for k in T if k not in state]
    grads = torch.autograd.grad((y * dyt).sum(), [x] + [T[k] for k in names], allow_unused=True)
    G = OrderedDict((k, (g.numpy() if g is not None else np.zeros(T[k].shape))) for k, g in zip(names, grads[1:]))
    yo = y.detach()
    if yo.dim() == 4:
        yo = yo.permute(0, 2, 3, 1)
    return yo.numpy(), grads[0].permute(0, 2, 3, 1).numpy(), G
