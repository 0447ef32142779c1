"""Stand-alone Laplacian pyramid front / back end on the HIP path -- the reference's
`mvae.layer_blocks.laplacian_transform_split` / `laplacian_transform_merge` (layer_blocks.py:23-99, 107-185; SURVEY.md
section 8(f) rank 3).  Same factory signatures; the returned objects are callable like the Keras models the reference
returns (`model(x)` / `model.predict(x)`), take / return NHWC float arrays and run `mvae_laplacian_split` /
`mvae_laplacian_merge` of libmvae_hip.so on `cuda:0`.  There is no CPU fallback.

`laplacian_transform_merge(trainable=True)` (the conv-mixing variant, layer_blocks.py:147-170) is built as a FORWARD
model: it owns its Conv2D weights (glorot_normal, `get_weights()` / `set_weights()`), runs `mvae_laplacian_merge_mix`,
and has no training loop -- the reference never trains it either (nothing in the reference calls it with
trainable=True)."""
import ctypes as C

import numpy as np

from . import _abi

DEFAULT_GAUSSIAN_XY_MAX = (1, 1)          # layer_blocks.py:16
DEFAULT_GAUSSIAN_KERNEL_SIZE = (3, 3)     # layer_blocks.py:17


def gaussian_kernel(size, nsig):
    """layer_blocks.py:980-1002: exp(-d^2/2) sampled on linspace(-nsig, nsig, size), normalised to sum 1."""
    if len(nsig) != 2 or len(size) != 2:
        raise AssertionError("size and nsig must have two elements")
    k1 = [np.linspace(-abs(nsig[i]), abs(nsig[i]), size[i], endpoint=True) for i in range(2)]
    x, y = np.meshgrid(k1[0], k1[1])
    g = np.exp(-(x * x + y * y) / 2.0)
    return g / g.sum()


def _check_dims(dims, levels):
    h, w = int(dims[0]), int(dims[1])
    if levels < 1:
        raise ValueError("levels should be >= 1")
    if h % (1 << (levels - 1)) or w % (1 << (levels - 1)):
        # the reference fails here too: Subtract() of tensors whose shapes differ after MaxPool / UpSampling2D
        raise ValueError("height and width must be multiples of 2^(levels-1)")


class _HipModel:
    def __init__(self, name, device):
        self.name = name
        self._device = int(device)
        self._lib = _abi.load_library()

    def _torch(self):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("the HIP path needs a GPU (cuda:%d); there is no CPU fallback" % self._device)
        return torch, torch.device("cuda", self._device)

    def predict(self, x, batch_size=None):
        return self(x)


class LaplacianSplit(_HipModel):
    def __init__(self, input_dims, levels, name, min_value, max_value, gaussian_xy_max, gaussian_kernel_size, device=0):
        super().__init__(name, device)
        if tuple(gaussian_kernel_size) != (3, 3):
            raise ValueError("the HIP kernel implements the 3x3 Gaussian (the reference's default)")
        self.input_dims = tuple(int(d) for d in input_dims)
        self.levels = int(levels)
        _check_dims(self.input_dims, self.levels)
        self.min_value, self.max_value = float(min_value), float(max_value)
        g = np.ascontiguousarray(gaussian_kernel(gaussian_kernel_size, gaussian_xy_max), dtype=np.float32)
        self._gauss = (C.c_float * 9)(*g.ravel())

    def __call__(self, x):
        torch, dev = self._torch()
        x = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
        if x.ndim != 4 or tuple(x.shape[1:]) != self.input_dims:
            raise ValueError("expected input of shape [B, %d, %d, %d]" % self.input_dims)
        b, h, w, c = x.shape
        if b == 0:
            return [np.zeros((0, h >> i, w >> i, c), np.float32) for i in range(self.levels)]
        xd = torch.from_numpy(x).to(dev)
        outs = [torch.empty((b, h >> i, w >> i, c), dtype=torch.float32, device=dev) for i in range(self.levels)]
        work = torch.empty((b * h * w * c * 4) // 3 + 16, dtype=torch.float32, device=dev)
        ptrs = (C.c_void_p * self.levels)(*[o.data_ptr() for o in outs])
        rc = self._lib.mvae_laplacian_split(self._device, C.c_void_p(xd.data_ptr()), b, h, w, c, self.levels,
                                            self.min_value, self.max_value, self._gauss, ptrs,
                                            C.c_void_p(work.data_ptr()),
                                            C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != _abi.MVAE_OK:
            raise RuntimeError("mvae_laplacian_split failed (%d)" % rc)
        torch.cuda.synchronize(dev)
        return [o.cpu().numpy() for o in outs]


class LaplacianMerge(_HipModel):
    def __init__(self, input_dims, levels, name, min_value, max_value, device=0):
        super().__init__(name, device)
        self.levels = int(levels)
        self.input_dims = [tuple(int(d) for d in dims) for dims in input_dims][:self.levels]
        if len(self.input_dims) != self.levels:
            raise ValueError("input_dims must list one shape per level")
        _check_dims(self.input_dims[0], self.levels)
        h, w, c = self.input_dims[0]
        for i, dims in enumerate(self.input_dims):
            if dims != (h >> i, w >> i, c):
                raise ValueError("level %d should have shape %s" % (i, (h >> i, w >> i, c)))
        self.min_value, self.max_value = float(min_value), float(max_value)

    def __call__(self, xs):
        torch, dev = self._torch()
        xs = [np.ascontiguousarray(np.asarray(x, dtype=np.float32)) for x in xs]
        if len(xs) != self.levels:
            raise ValueError("expected %d inputs" % self.levels)
        b = xs[0].shape[0]
        for x, dims in zip(xs, self.input_dims):
            if x.ndim != 4 or x.shape[0] != b or tuple(x.shape[1:]) != dims:
                raise ValueError("expected inputs of shapes [B, *%s]" % (self.input_dims,))
        h, w, c = self.input_dims[0]
        if b == 0:
            return np.zeros((0, h, w, c), np.float32)
        ds = [torch.from_numpy(x).to(dev) for x in xs]
        out = torch.empty((b, h, w, c), dtype=torch.float32, device=dev)
        work = torch.empty(2 * b * h * w * c, dtype=torch.float32, device=dev)
        ptrs = (C.c_void_p * self.levels)(*[d.data_ptr() for d in ds])
        rc = self._lib.mvae_laplacian_merge(self._device, ptrs, b, h, w, c, self.levels, self.min_value,
                                            self.max_value, C.c_void_p(out.data_ptr()), C.c_void_p(work.data_ptr()),
                                            C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != _abi.MVAE_OK:
            raise RuntimeError("mvae_laplacian_merge failed (%d)" % rc)
        torch.cuda.synchronize(dev)
        return out.cpu().numpy()


class LaplacianMergeMix(LaplacianMerge):
    """laplacian_transform_merge(trainable=True): per level Concatenate([up2(out), level]) -> Conv2D(filters, 3x3, relu)
    -> Conv2D(C, 1x1, tanh, no bias) -> Add (layer_blocks.py:141-171)."""

    def __init__(self, input_dims, levels, name, min_value, max_value, filters, activation, kernel_initializer, seed=42,
                 device=0):
        super().__init__(input_dims, levels, name, min_value, max_value, device)
        if activation != "relu":
            raise ValueError("the HIP path implements the reference's default activation ('relu')")
        if kernel_initializer != "glorot_normal":
            raise ValueError("the HIP path implements the reference's default initializer ('glorot_normal')")
        self.filters = int(filters)
        if self.filters <= 0:
            raise ValueError("Filters should be > 0")
        from .initializers import truncated_normal, _TRUNC_STD
        rng = np.random.default_rng(seed)
        c = self.input_dims[0][2]
        self._weights = []
        for i in range(self.levels - 1):
            w3 = truncated_normal(rng, (3, 3, 2 * c, self.filters), np.sqrt(2.0 / (9 * 2 * c + 9 * self.filters)) / _TRUNC_STD)
            w1 = truncated_normal(rng, (1, 1, self.filters, c), np.sqrt(2.0 / (self.filters + c)) / _TRUNC_STD)
            self._weights.append({"mix.w": w3, "mix.b": np.zeros(self.filters, np.float32), "retarget.w": w1})

    def get_weights(self):
        return [dict(w) for w in self._weights]

    def set_weights(self, weights):
        if len(weights) != len(self._weights):
            raise ValueError("expected %d weight sets" % len(self._weights))
        new = []
        for w, old in zip(weights, self._weights):
            cur = {}
            for k, v in old.items():
                a = np.ascontiguousarray(np.asarray(w[k], np.float32))
                if a.shape != v.shape:
                    raise ValueError("%s has shape %s, expected %s" % (k, a.shape, v.shape))
                cur[k] = a
            new.append(cur)
        self._weights = new

    def __call__(self, xs):
        torch, dev = self._torch()
        xs = [np.ascontiguousarray(np.asarray(x, dtype=np.float32)) for x in xs]
        if len(xs) != self.levels:
            raise ValueError("expected %d inputs" % self.levels)
        b = xs[0].shape[0]
        for x, dims in zip(xs, self.input_dims):
            if x.ndim != 4 or x.shape[0] != b or tuple(x.shape[1:]) != dims:
                raise ValueError("expected inputs of shapes [B, *%s]" % (self.input_dims,))
        h, w, c = self.input_dims[0]
        if b == 0:
            return np.zeros((0, h, w, c), np.float32)
        ds = [torch.from_numpy(x).to(dev) for x in xs]
        wd = [{k: torch.from_numpy(v).to(dev) for k, v in ws.items()} for ws in self._weights]
        out = torch.empty((b, h, w, c), dtype=torch.float32, device=dev)
        work = torch.empty(b * h * w * (4 * c + self.filters) + 64, dtype=torch.float32, device=dev)
        n = max(self.levels - 1, 1)
        arr = lambda key: (C.c_void_p * n)(*([d[key].data_ptr() for d in wd] or [0]))
        ptrs = (C.c_void_p * self.levels)(*[d.data_ptr() for d in ds])
        rc = self._lib.mvae_laplacian_merge_mix(self._device, ptrs, b, h, w, c, self.levels, self.filters, arr("mix.w"),
                                                arr("mix.b"), arr("retarget.w"), self.min_value, self.max_value,
                                                C.c_void_p(out.data_ptr()), C.c_void_p(work.data_ptr()),
                                                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != _abi.MVAE_OK:
            raise RuntimeError("mvae_laplacian_merge_mix failed (%d)" % rc)
        torch.cuda.synchronize(dev)
        return out.cpu().numpy()


def laplacian_transform_split(input_dims, levels, name=None, min_value=0.0, max_value=255.0,
                              gaussian_xy_max=DEFAULT_GAUSSIAN_XY_MAX,
                              gaussian_kernel_size=DEFAULT_GAUSSIAN_KERNEL_SIZE):
    """layer_blocks.py:23-99: normalise to [-1, 1], then `levels` Laplacian levels (finest first)."""
    return LaplacianSplit(input_dims, levels, name, min_value, max_value, gaussian_xy_max, gaussian_kernel_size)


def laplacian_transform_merge(input_dims, levels, name=None, min_value=0.0, max_value=255.0, trainable=False,
                              filters=32, activation="relu", kernel_regularizer="l1",
                              kernel_initializer="glorot_normal"):
    """layer_blocks.py:107-185: upsample-and-add from the coarsest level (trainable=True: with the conv mixing of
    :147-170 in front of every Add), denormalise, clip."""
    if trainable:
        return LaplacianMergeMix(input_dims, levels, name, min_value, max_value, filters, activation, kernel_initializer)
    return LaplacianMerge(input_dims, levels, name, min_value, max_value)


# ==============================================================================
# mobilenetV2_block and resnet_block of the reference's block library as stand-alone HIP layers.
# The reference's functions take a Keras tensor and return a Keras tensor; here the factory takes the input shape and
# returns a layer object that owns its weights (glorot_normal kernels, zero biases, like the reference's defaults) and
# offers forward / backward through the C ABI (mvae_mnv2_* / mvae_resnet_*).
# ==============================================================================
class _HipBlock(_HipModel):
    def __init__(self, name, input_dims, filters, shapes, seed, device):
        super().__init__(name, device)
        self.input_dims = tuple(int(d) for d in input_dims)
        self.filters = int(filters)
        from .initializers import truncated_normal, _TRUNC_STD
        rng = np.random.default_rng(seed)
        self._weights = {}
        for k, shp in shapes.items():
            if k.endswith(".b"):
                self._weights[k] = np.zeros(shp, np.float32)
            else:
                kh, kw, ci, co = shp
                fan_in, fan_out = kh * kw * ci, kh * kw * co        # keras: receptive field x channels (depthwise: co = 1)
                self._weights[k] = truncated_normal(rng, shp, np.sqrt(2.0 / (fan_in + fan_out)) / _TRUNC_STD)
        self._saved = None

    def get_weights(self):
        return {k: v.copy() for k, v in self._weights.items()}

    def set_weights(self, weights):
        for k, v in self._weights.items():
            a = np.ascontiguousarray(np.asarray(weights[k], np.float32))
            if a.shape != v.shape:
                raise ValueError("%s has shape %s, expected %s" % (k, a.shape, v.shape))
            self._weights[k] = a

    def output_shape(self, batch):
        raise NotImplementedError

    def _dev(self, torch, dev, a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)

    def _check_x(self, x):
        x = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
        if x.ndim != 4 or tuple(x.shape[1:]) != self.input_dims:
            raise ValueError("expected input of shape [B, %d, %d, %d]" % self.input_dims)
        return x

    def __call__(self, x):
        return self.forward(x)


class MobileNetV2Block(_HipBlock):
    """mobilenetV2_block (reference layer_blocks.py:468-550), use_batchnorm=False, dropout 0."""

    def __init__(self, input_dims, filters=32, name=None, seed=42, device=0):
        if filters <= 0:
            raise ValueError("Filters should be > 0")
        c = int(input_dims[2])
        shapes = {"conv0.w": (1, 1, c, filters), "conv0.b": (filters,), "conv1.w": (3, 3, filters, 1), "conv1.b": (filters,),
                  "conv2.w": (1, 1, filters, c), "conv2.b": (c,)}
        super().__init__(name or "mobilenetV2_", input_dims, filters, shapes, seed, device)

    def forward(self, x):
        torch, dev = self._torch()
        x = self._check_x(x)
        b, h, w, c = x.shape
        f = self.filters
        xd = self._dev(torch, dev, x)
        wd = {k: self._dev(torch, dev, v) for k, v in self._weights.items()}
        t0 = torch.empty((b, h, w, f), dtype=torch.float32, device=dev); t1 = torch.empty_like(t0)
        u = torch.empty((b, h, w, c), dtype=torch.float32, device=dev); y = torch.empty_like(u)
        p = lambda t: C.c_void_p(t.data_ptr())
        rc = self._lib.mvae_mnv2_forward(self._device, p(xd), b, h, w, c, f, p(wd["conv0.w"]), p(wd["conv0.b"]),
                                         p(wd["conv1.w"]), p(wd["conv1.b"]), p(wd["conv2.w"]), p(wd["conv2.b"]), p(t0), p(t1),
                                         p(u), p(y), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != _abi.MVAE_OK:
            raise RuntimeError("mvae_mnv2_forward failed (%d)" % rc)
        torch.cuda.synchronize(dev)
        self._saved = dict(x=xd, t0=t0, t1=t1, u=u, w=wd)
        return y.cpu().numpy()

    def backward(self, dy):
        """Gradient of sum(y * dy): (dx, {name: gradient}) for the inputs of the last forward()."""
        if self._saved is None:
            raise RuntimeError("backward() needs a preceding forward()")
        torch, dev = self._torch()
        sv = self._saved
        b, h, w, c = sv["x"].shape
        f = self.filters
        dyd = self._dev(torch, dev, np.asarray(dy, np.float32).reshape(b, h, w, c))
        dx = torch.empty_like(sv["x"])
        g = {k: torch.zeros_like(v) for k, v in sv["w"].items()}
        work = torch.empty(b * h * w * (c + 2 * f) + 64, dtype=torch.float32, device=dev)
        p = lambda t: C.c_void_p(t.data_ptr())
        rc = self._lib.mvae_mnv2_backward(self._device, p(sv["x"]), p(sv["t0"]), p(sv["t1"]), p(sv["u"]), p(dyd), b, h, w, c, f,
                                          p(sv["w"]["conv0.w"]), p(sv["w"]["conv1.w"]), p(sv["w"]["conv2.w"]), p(dx),
                                          p(g["conv0.w"]), p(g["conv0.b"]), p(g["conv1.w"]), p(g["conv1.b"]), p(g["conv2.w"]),
                                          p(g["conv2.b"]), p(work), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != _abi.MVAE_OK:
            raise RuntimeError("mvae_mnv2_backward failed (%d)" % rc)
        torch.cuda.synchronize(dev)
        return dx.cpu().numpy(), {k: v.cpu().numpy() for k, v in g.items()}


class ResnetBlock(_HipBlock):
    """resnet_block (reference layer_blocks.py:789-887), strides (1, 1), use_batchnorm=False, dropout 0."""

    def __init__(self, input_dims, filters=32, kernel_size=(3, 3), strides=(1, 1), activation="relu", name=None, seed=42,
                 device=0):
        if filters <= 0:
            raise ValueError("Filters should be > 0")
        if tuple(strides) != (1, 1):
            raise ValueError("the HIP path implements the reference's default strides (1, 1)")
        if activation not in ("relu", "linear"):
            raise ValueError("the HIP path implements activation 'relu' (the reference's default) and 'linear'")
        self.kernel_size = (int(kernel_size[0]), int(kernel_size[1]))
        self.activation = activation
        c = int(input_dims[2])
        kh, kw = self.kernel_size
        shapes = {"conv0.w": (kh, kw, c, filters), "conv0.b": (filters,), "conv1.w": (kh, kw, filters, filters),
                  "conv1.b": (filters,)}
        if c != filters:
            shapes["skip.w"] = (1, 1, c, filters); shapes["skip.b"] = (filters,)
        super().__init__(name or "resnet_", input_dims, filters, shapes, seed, device)

    def forward(self, x):
        torch, dev = self._torch()
        x = self._check_x(x)
        b, h, w, c = x.shape
        f = self.filters
        kh, kw = self.kernel_size
        xd = self._dev(torch, dev, x)
        wd = {k: self._dev(torch, dev, v) for k, v in self._weights.items()}
        x0 = torch.empty((b, h, w, f), dtype=torch.float32, device=dev)
        y = torch.empty_like(x0)
        skip = torch.empty_like(x0) if c != f else None
        p = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
        rc = self._lib.mvae_resnet_forward(self._device, p(xd), b, h, w, c, f, kh, kw, 1 if self.activation == "relu" else 0,
                                           p(wd["conv0.w"]), p(wd["conv0.b"]), p(wd["conv1.w"]), p(wd["conv1.b"]),
                                           p(wd.get("skip.w")), p(wd.get("skip.b")), p(x0), p(skip), p(y),
                                           C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != _abi.MVAE_OK:
            raise RuntimeError("mvae_resnet_forward failed (%d)" % rc)
        torch.cuda.synchronize(dev)
        self._saved = dict(x=xd, x0=x0, y=y, w=wd)
        return y.cpu().numpy()

    def backward(self, dy):
        if self._saved is None:
            raise RuntimeError("backward() needs a preceding forward()")
        torch, dev = self._torch()
        sv = self._saved
        b, h, w, c = sv["x"].shape
        f = self.filters
        kh, kw = self.kernel_size
        dyd = self._dev(torch, dev, np.asarray(dy, np.float32).reshape(b, h, w, f))
        dx = torch.empty_like(sv["x"])
        g = {k: torch.zeros_like(v) for k, v in sv["w"].items()}
        work = torch.empty(b * h * w * (2 * f + c) + 64, dtype=torch.float32, device=dev)
        p = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
        rc = self._lib.mvae_resnet_backward(self._device, p(sv["x"]), p(sv["x0"]), p(sv["y"]), p(dyd), b, h, w, c, f, kh, kw,
                                            1 if self.activation == "relu" else 0, p(sv["w"]["conv0.w"]), p(sv["w"]["conv1.w"]),
                                            p(sv["w"].get("skip.w")), p(dx), p(g["conv0.w"]), p(g["conv0.b"]), p(g["conv1.w"]),
                                            p(g["conv1.b"]), p(g.get("skip.w")), p(g.get("skip.b")), p(work),
                                            C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != _abi.MVAE_OK:
            raise RuntimeError("mvae_resnet_backward failed (%d)" % rc)
        torch.cuda.synchronize(dev)
        return dx.cpu().numpy(), {k: v.cpu().numpy() for k, v in g.items()}


def mobilenetV2_block(input_dims, filters=32, dropout_ratio=0.0, use_batchnorm=False, prefix="mobilenetV2_",
                      initializer="glorot_normal", regularizer="l1", channels_index=3):
    """layer_blocks.py:468-550 with the reference's argument checks; the input is given by its shape (H, W, C)."""
    if input_dims is None:
        raise ValueError("input_layer cannot be empty")
    if filters <= 0:
        raise ValueError("Filters should be > 0")
    if dropout_ratio is not None and (dropout_ratio > 1.0 or dropout_ratio < 0.0):
        raise ValueError("Dropout ration must be [0, 1]")
    if (dropout_ratio or 0.0) > 0.0 or initializer != "glorot_normal":
        raise ValueError("the HIP path implements dropout_ratio 0 and glorot_normal (the reference's defaults)")
    if use_batchnorm:                       # :521-537: BatchNormalization behind the depthwise and the second 1x1 convolution
        from .layer_ops import MobileNetV2BlockBN
        return MobileNetV2BlockBN(input_dims, filters, name=prefix)
    return MobileNetV2Block(input_dims, filters, name=prefix)


def resnet_block(input_dims, filters=32, kernel_size=(3, 3), strides=(1, 1), activation="relu", dropout_ratio=0.0,
                 use_batchnorm=False, prefix="resnet_", initializer="glorot_normal", regularizer="l1", channels_index=3):
    """layer_blocks.py:789-887 with the reference's argument checks; the input is given by its shape (H, W, C)."""
    if input_dims is None:
        raise ValueError("input_layer cannot be empty")
    if filters <= 0:
        raise ValueError("Filters should be > 0")
    if dropout_ratio is not None and (dropout_ratio > 1.0 or dropout_ratio < 0.0):
        raise ValueError("Dropout ration must be [0, 1]")
    if (dropout_ratio or 0.0) > 0.0 or initializer != "glorot_normal":
        raise ValueError("the HIP path implements dropout_ratio 0 and glorot_normal (the reference's defaults)")
    if use_batchnorm or tuple(strides) != (1, 1) or activation not in ("relu", "linear"):
        # :847-853 (MaxPooling2D skip path of a strided block), :884-886 (BatchNormalization): the layer-operator composition
        from .layer_ops import ResnetBlockGeneral
        return ResnetBlockGeneral(input_dims, filters, kernel_size, strides, activation, use_batchnorm, name=prefix)
    return ResnetBlock(input_dims, filters, kernel_size, strides, activation, name=prefix)


# ---- the rest of the block library (layer_blocks.py:191-412, 654-783): assembled from the layer operators (layer_ops.py) ----
DEFAULT_ATTENUATION_MULTIPLIER = 4.0      # layer_blocks.py:14


def attenuate_activation(input_layer, multiplier=DEFAULT_ATTENUATION_MULTIPLIER):
    """layer_blocks.py:191-198 applied to an array on the device."""
    from .layer_ops import attenuate_activation as _att
    return _att(input_layer, multiplier)


def _dims4(input_dims, what="works only on 4d tensors"):
    if input_dims is None:
        raise ValueError("input_layer cannot be empty")
    if len(input_dims) != 3:                      # (H, W, C): the batch axis is implicit
        raise ValueError(what)
    return tuple(int(d) for d in input_dims)


def attention_block(input_dims, filters=32, kernel_size=(1, 1), activation="linear", initializer="glorot_normal", regularizer=None,
                    prefix="attention_"):
    """layer_blocks.py:654-728 with the reference's argument checks; the input is given by its shape (H, W, C)."""
    from .layer_ops import AttentionBlock
    dims = _dims4(input_dims, "only supports 4d tensors")
    if filters <= 0:
        raise ValueError("Filters should be > 0")
    if initializer != "glorot_normal":
        raise ValueError("the HIP path implements glorot_normal (the reference's default)")
    return AttentionBlock(dims, filters, kernel_size, activation, name=prefix)


def self_attention_block(input_dims, filters=32, kernel_size=(1, 1), activation="linear", initializer="glorot_normal",
                         regularizer=None, prefix="self_attention_", channels_index=3):
    """layer_blocks.py:734-783."""
    from .layer_ops import SelfAttentionBlock
    dims = _dims4(input_dims, "only supports 4d tensors")
    if filters <= 0:
        raise ValueError("Filters should be > 0")
    if initializer != "glorot_normal":
        raise ValueError("the HIP path implements glorot_normal (the reference's default)")
    return SelfAttentionBlock(dims, filters, kernel_size, activation, name=prefix)


def excite_inhibit_spatial_mask_block(input_dims, filters=32, kernel_size=(3, 3), flatten=False, add_batchnorm=False,
                                      first_level_activation="relu", second_level_activation="sigmoid",
                                      multiplier=DEFAULT_ATTENUATION_MULTIPLIER, kernel_regularizer="l1",
                                      kernel_initializer="glorot_normal", channels_index=3):
    """layer_blocks.py:204-271."""
    from .layer_ops import ExciteInhibitSpatialMask
    dims = _dims4(input_dims)
    if add_batchnorm or kernel_initializer != "glorot_normal":
        raise ValueError("the HIP path implements add_batchnorm=False and glorot_normal (the reference's defaults)")
    return ExciteInhibitSpatialMask(dims, filters, kernel_size, flatten, first_level_activation, second_level_activation, multiplier)


def excite_inhibit_channel_mask_block(input_dims, filters=32, kernel_size=(3, 3), shared=True, add_batchnorm=False,
                                      first_level_activation="linear", second_level_activation="sigmoid",
                                      multiplier=DEFAULT_ATTENUATION_MULTIPLIER, kernel_regularizer="l1",
                                      kernel_initializer="glorot_normal", channels_index=3):
    """layer_blocks.py:277-350."""
    from .layer_ops import ExciteInhibitChannelMask
    dims = _dims4(input_dims)
    if add_batchnorm or kernel_initializer != "glorot_normal":
        raise ValueError("the HIP path implements add_batchnorm=False and glorot_normal (the reference's defaults)")
    return ExciteInhibitChannelMask(dims, filters, kernel_size, shared, first_level_activation, second_level_activation, multiplier)


def excite_inhibit_block(input_dims, filters=32, kernel_size=(3, 3), kernel_regularizer="l1", kernel_initializer="glorot_normal",
                         channels_index=3):
    """layer_blocks.py:356-412."""
    from .layer_ops import ExciteInhibitBlock
    dims = _dims4(input_dims)
    if kernel_initializer != "glorot_normal":
        raise ValueError("the HIP path implements glorot_normal (the reference's default)")
    return ExciteInhibitBlock(dims, filters, kernel_size)
