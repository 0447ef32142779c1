"""
Device engine: owns one mvae_handle (C ABI, include/mvae_hip.h) plus the torch-allocated device memory it is
bound to.  torch is plumbing only (device memory, the HIP stream, torch.distributed/RCCL); every kernel on
the path is in libmvae_hip.so.  Replaces what Keras/TensorFlow did for the reference when
mvae/multiscale_vae.py:550-557 called `self._model_trainable.fit`.
"""
import ctypes as C
import os
from collections import OrderedDict

import numpy as np

from . import _abi


class MvaeError(RuntimeError):
    pass


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Engine:
    def __init__(self, input_dims, z_dims, encoder, decoder, min_value, max_value, sample_std, max_batch,
                 act_dtype="f32"):
        self.lib = _abi.load_library()
        self.act_dtype = act_dtype
        self.cfg = _abi.make_config(input_dims, z_dims, encoder, decoder, min_value, max_value, sample_std, max_batch,
                                    act_dtype)
        self.input_dims = tuple(int(v) for v in input_dims)
        self.levels = len(z_dims)
        self.max_batch = int(max_batch)
        h = C.c_void_p()
        rc = self.lib.mvae_create(C.byref(self.cfg), C.byref(h))
        if rc != _abi.MVAE_OK:
            msg = self.lib.mvae_last_error(None).decode()
            raise ValueError(msg) if rc == _abi.MVAE_E_INVALID else MvaeError(msg)
        self.h = h
        self.bound = False
        self._read_tables()

    # ------------------------------------------------------------------ tables
    def _read_tables(self):
        lib, h = self.lib, self.h
        self.P = lib.mvae_param_elems(h)
        self.S = lib.mvae_state_elems(h)
        self.Z = lib.mvae_latent_dim(h)
        self.R = lib.mvae_reduce_elems(h)
        self.metrics_off = lib.mvae_metrics_offset(h)
        self.reduce_split = lib.mvae_reduce_split(h)      # leading Dense-weight region: final after backward phase 0
        self.ws_bytes = lib.mvae_workspace_bytes(h)
        self.param_table = OrderedDict()
        name = C.create_string_buffer(_abi.MVAE_NAME_CAP)
        shape = (C.c_int64 * 4)()
        ndim, reg, off = C.c_int32(), C.c_int32(), C.c_int64()
        for i in range(lib.mvae_param_count(h)):
            lib.mvae_param_info(h, i, name, _abi.MVAE_NAME_CAP, shape, C.byref(ndim), C.byref(off), C.byref(reg))
            self.param_table[name.value.decode()] = dict(
                shape=tuple(int(shape[k]) for k in range(ndim.value)), offset=int(off.value),
                reg=_abi.REG_NAMES[reg.value])
        self.state_table = OrderedDict()
        elems = C.c_int64()
        for i in range(lib.mvae_state_count(h)):
            lib.mvae_state_info(h, i, name, _abi.MVAE_NAME_CAP, C.byref(elems), C.byref(off))
            self.state_table[name.value.decode()] = dict(shape=(int(elems.value),), offset=int(off.value))

    def _check(self, rc):
        if rc != _abi.MVAE_OK:
            msg = self.lib.mvae_last_error(self.h).decode()
            raise ValueError(msg) if rc == _abi.MVAE_E_INVALID else MvaeError("mvae error %d: %s" % (rc, msg))

    def close(self):
        if getattr(self, "h", None):
            self.lib.mvae_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ device memory
    def bind(self, device_index=None):
        import torch
        if not torch.cuda.is_available():
            raise MvaeError("no HIP device visible: the MI355X kernels are the only compute path (no CPU fallback)")
        if device_index is None:
            device_index = torch.cuda.current_device()
        self.torch = torch
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.params = torch.zeros(self.P, **f32)
        self.reduce = torch.zeros(self.R, **f32)          # [grads | BN batch statistics | metrics]
        self.accum = torch.full((self.P,), 0.1, **f32)    # keras Adagrad initial_accumulator_value
        self.state = torch.zeros(max(self.S, 1), **f32)
        self.workspace = torch.zeros(self.ws_bytes // 4 + 64, **f32)
        st = np.zeros(max(self.S, 1), np.float32)
        for k, v in self.state_table.items():              # moving_mean = 0, moving_variance = 1
            if k.endswith(".var"):
                st[v["offset"]:v["offset"] + v["shape"][0]] = 1.0
        self.state.copy_(torch.from_numpy(st))
        torch.cuda.synchronize(self.device)
        self._check(self.lib.mvae_bind(self.h, device_index, _ptr(self.params), _ptr(self.reduce), _ptr(self.accum),
                                       _ptr(self.state), _ptr(self.workspace), self.workspace.numel() * 4))
        # all kernels run on this stream: a real (non-default) HIP stream, so that the library can capture its
        # launch sequences into hipGraphs; torch ops that touch our buffers are issued under it as well
        self.stream = torch.cuda.Stream(self.device)
        self.copy_stream = None                 # pinned double-buffered H2D of datasets that do not fit in HBM
        self.comm_stream = None                 # the early all-reduce of the Dense-weight gradients (DP overlap)
        self.dataset = None
        self.bound = True
        return self

    def _stream(self):
        return C.c_void_p(self.stream.cuda_stream)

    def _enter(self):
        """Order our stream after whatever the caller enqueued on the current stream (input tensors)."""
        self.stream.wait_stream(self.torch.cuda.current_stream(self.device))

    def sync(self):
        self.stream.synchronize()

    def _pack(self, table, values, total):
        flat = np.zeros(total, np.float32)
        for k, meta in table.items():
            a = np.asarray(values[k], np.float32)
            if tuple(a.shape) != tuple(meta["shape"]):
                raise ValueError("tensor %s has shape %s, expected %s" % (k, a.shape, meta["shape"]))
            flat[meta["offset"]:meta["offset"] + a.size] = a.ravel()
        return flat

    def _unpack(self, table, flat):
        out = OrderedDict()
        for k, meta in table.items():
            n = int(np.prod(meta["shape"]))
            out[k] = flat[meta["offset"]:meta["offset"] + n].reshape(meta["shape"]).copy()
        return out

    def _upload(self, dst, flat):
        self.sync()
        dst.copy_(self.torch.from_numpy(flat))
        self.torch.cuda.synchronize(self.device)

    def _download(self, src):
        self.sync()
        return src.cpu().numpy()

    def set_params(self, values):
        self._upload(self.params, self._pack(self.param_table, values, self.P))

    def get_params(self):
        return self._unpack(self.param_table, self._download(self.params))

    def set_accum(self, values):
        self._upload(self.accum, self._pack(self.param_table, values, self.P))

    def get_accum(self):
        return self._unpack(self.param_table, self._download(self.accum))

    def get_grads(self):
        return self._unpack(self.param_table, self._download(self.reduce[:self.P]))

    def set_state(self, values):
        self._upload(self.state, self._pack(self.state_table, values, max(self.S, 1)))

    def get_state(self):
        return self._unpack(self.state_table, self._download(self.state))

    def to_device(self, a):
        t = self.torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32))
        return t.to(self.device, non_blocking=False)

    # ------------------------------------------------------------------ the hot path
    def forward(self, x, training, eps=None, noise=None, keep_mask=None, seed=0, outputs=("recon",)):
        """x: contiguous float32 CUDA tensor [B,H,W,C].  Returns the requested device tensors."""
        torch = self.torch
        B = int(x.shape[0])
        if tuple(x.shape[1:]) != self.input_dims or x.dtype != torch.float32 or not x.is_contiguous():
            raise ValueError("x must be a contiguous float32 tensor [B,%d,%d,%d]" % self.input_dims)
        io = _abi.MvaeStepIO()
        io.x, io.batch, io.training, io.seed = x.data_ptr(), B, 1 if training else 0, int(seed) & (2 ** 64 - 1)
        io.eps = eps.data_ptr() if eps is not None else None
        io.noise = noise.data_ptr() if noise is not None else None
        io.keep_mask = keep_mask.data_ptr() if keep_mask is not None else None
        out = {}
        f32 = dict(dtype=torch.float32, device=self.device)
        if "recon" in outputs:
            out["recon"] = torch.empty((B,) + self.input_dims, **f32); io.recon = out["recon"].data_ptr()
        for k in ("mu", "log_var", "z"):
            if k in outputs:
                out[k] = torch.empty((B, self.Z), **f32); setattr(io, k, out[k].data_ptr())
        if "losses" in outputs:
            out["losses"] = torch.empty((B, 3 + self.levels), **f32); io.losses = out["losses"].data_ptr()
        self._keep = (x, eps, noise, keep_mask)       # backward reads x / eps again
        self._enter()
        # The kernels run on self.stream, not on the stream the caller allocated these tensors on: tell torch's caching
        # allocator, or a tensor dropped by the caller (train() feeds a fresh batch every step and the host runs several
        # steps ahead of the device) is handed out again while this step's kernels still read it.
        for t in self._keep:
            if t is not None:
                t.record_stream(self.stream)
        self._check(self.lib.mvae_forward(self.h, C.byref(io), self._stream()))
        if out:
            self.sync()                               # the caller reads the outputs from another stream
        return out

    def backward(self, r_factor, kl_factor):
        self._check(self.lib.mvae_backward(self.h, float(r_factor), float(kl_factor), self._stream()))

    def backward_phase(self, phase, r_factor, kl_factor):
        self._check(self.lib.mvae_backward_phase(self.h, int(phase), float(r_factor), float(kl_factor), self._stream()))

    def graph_stats(self):
        """(captured launch sequences, calls that fell back to eager launches)."""
        cap, eager = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.mvae_graph_stats(self.h, C.byref(cap), C.byref(eager)))
        return cap.value, eager.value

    def fused_launch_stats(self):
        """Process-wide launch geometry of the image-resident fused kernels (mvae_fused_launch_stats)."""
        v = [C.c_int32(0) for _ in range(3)]
        self._check(self.lib.mvae_fused_launch_stats(*[C.byref(x) for x in v]))
        return {"fwd": v[0].value, "bwd": v[1].value, "max_images_per_block": v[2].value}

    def dp_overlap_active(self):
        """Opt-in (MVAE_DP_OVERLAP=1): split the gradient exchange in two messages, the leading Dense-weight region of the
        arena (137 of 144 MB for the 256x256 configurations) travelling while the encoder half of the backward pass still
        runs.  OFF by default: no multi-GPU RCCL run has yet shown it bitwise-equal to the single-message path and faster
        (it has only run over gloo on one GPU and over RCCL at world size 1, where the all-reduce is a no-op)."""
        mode = os.environ.get("MVAE_DP_OVERLAP", "")
        if mode not in ("1", "force") or self.reduce_split <= 0:
            return False
        if mode == "1" and max(self.packed_f32_hazard()) > 0:
            # RCCL's float32 reduce kernels contain v_pk_*_f32 (tools/rccl_isa_scan.py); this board returned wrong values
            # from such instructions beside this library's bf16-MFMA kernels (the bind-time self-test): no overlap here.
            if not getattr(self, "_overlap_warned", False):
                self._overlap_warned = True
                import warnings
                warnings.warn("MVAE_DP_OVERLAP=1 refused: the packed-float32 hazard self-test counted %d / %d wrong values "
                              "on this GPU and RCCL's reduce kernels use packed float32; single-message exchange instead "
                              "(MVAE_DP_OVERLAP=force overrides)" % self.packed_f32_hazard())
            return False
        return True

    def packed_f32_hazard(self):
        """(wrong packed-float32 values beside the split-bf16 kernels, beside the bf16-storage kernels); -1 = not measured."""
        a, b = C.c_int32(-1), C.c_int32(-1)
        self.lib.mvae_packed_f32_hazard(C.byref(a), C.byref(b))
        return a.value, b.value

    def apply(self, lr, clip_norm, grad_scale=1.0):
        self._check(self.lib.mvae_apply_adagrad(self.h, float(lr), float(clip_norm if clip_norm else 0.0),
                                                float(grad_scale), self._stream()))

    def collective_active(self, force=False):
        """True when train_step all-reduces: torch.distributed is up and (world > 1 or forced).  `force` (or
        MVAE_FORCE_COLLECTIVE=1) sends a world-size-1 group through the same RCCL call, which is how the
        collective branch is exercised on a one-GPU box."""
        dist = self.torch.distributed
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size() > 1 or force or os.environ.get("MVAE_FORCE_COLLECTIVE", "") == "1"

    def train_step(self, x, lr, r_factor, kl_factor, clip_norm, eps=None, noise=None, keep_mask=None, seed=0,
                   force_collective=False, timing=None):
        """forward + backward (+ one RCCL all-reduce of the reduce arena when torch.distributed is up) + Adagrad.
        timing: optional list; a (start, end) pair of torch events around the all-reduce is appended."""
        torch = self.torch
        self.forward(x, True, eps, noise, keep_mask, seed, outputs=())
        scale = 1.0
        if not self.collective_active(force_collective):
            self.backward(r_factor, kl_factor)
        elif not self.dp_overlap_active():
            self.backward(r_factor, kl_factor)
            dist = torch.distributed
            with torch.cuda.stream(self.stream):
                if timing is not None:
                    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                    e0.record(self.stream)
                dist.all_reduce(self.reduce)              # grads | BN batch statistics | metrics, one message
                if timing is not None:
                    e1.record(self.stream)
                    timing.append((e0, e1))
            scale = 1.0 / dist.get_world_size()
        else:
            # Dense-weight gradients (the leading arena region, final after phase 0) travel while the encoder halves
            # of the backward pass still run; the rest follows as a second message.  Two collectives per step, issued in
            # the same order on every rank.
            dist = torch.distributed
            D = self.reduce_split
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream(self.device)
            self.backward_phase(0, r_factor, kl_factor)
            ready = torch.cuda.Event(); ready.record(self.stream)
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ready)
                if timing is not None:
                    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                    e0.record(self.comm_stream)
                dist.all_reduce(self.reduce[:D])
                if timing is not None:
                    e1.record(self.comm_stream)
                    timing.append((e0, e1))
            self.backward_phase(1, r_factor, kl_factor)
            with torch.cuda.stream(self.stream):
                dist.all_reduce(self.reduce[D:])
            self.stream.wait_stream(self.comm_stream)
            scale = 1.0 / dist.get_world_size()
        self.apply(lr, clip_norm, scale)

    # ------------------------------------------------------------------ RCCL bound through the C ABI (mvae_comm_*)
    def comm_unique_id(self):
        """128-byte ncclUniqueId (rank 0 creates it and hands it to the other ranks out of band)."""
        buf = C.create_string_buffer(_abi.MVAE_COMM_ID_BYTES)
        self._check(self.lib.mvae_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id, rank, nranks):
        """ncclCommInitRank on the handle's device: collective over all ranks."""
        if len(unique_id) != _abi.MVAE_COMM_ID_BYTES:
            raise ValueError("unique_id must be %d bytes" % _abi.MVAE_COMM_ID_BYTES)
        self._enter()
        self._check(self.lib.mvae_comm_init(self.h, C.create_string_buffer(bytes(unique_id), _abi.MVAE_COMM_ID_BYTES), int(rank), int(nranks)))

    def comm_size(self):
        return int(self.lib.mvae_comm_size(self.h))

    def allreduce(self, offset=0, count=-1):
        self._check(self.lib.mvae_allreduce(self.h, int(offset), int(count), self._stream()))

    def train_step_dp_abi(self, x, lr, r_factor, kl_factor, clip_norm, eps=None, noise=None, keep_mask=None, seed=0):
        """forward + backward + ncclAllReduce of the reduce arena + Adagrad(grad_scale = 1 / ranks) in ONE C-ABI call."""
        io = self._step_io(x, eps, noise, keep_mask, seed)
        self._check(self.lib.mvae_train_step_dp(self.h, C.byref(io), float(r_factor), float(kl_factor), float(lr),
                                                float(clip_norm if clip_norm else 0.0), self._stream()))

    def _step_io(self, x, eps, noise, keep_mask, seed):
        io = _abi.MvaeStepIO()
        io.x, io.batch, io.training, io.seed = x.data_ptr(), int(x.shape[0]), 1, int(seed) & (2 ** 64 - 1)
        io.eps = eps.data_ptr() if eps is not None else None
        io.noise = noise.data_ptr() if noise is not None else None
        io.keep_mask = keep_mask.data_ptr() if keep_mask is not None else None
        self._keep = (x, eps, noise, keep_mask)
        self._enter()
        for t in self._keep:
            if t is not None:
                t.record_stream(self.stream)
        return io

    def train_step_abi(self, x, lr, r_factor, kl_factor, clip_norm, eps=None, noise=None, keep_mask=None, seed=0):
        """The single-device step through ONE C-ABI call, mvae_train_step (what INTEGRATION.md binds)."""
        io = _abi.MvaeStepIO()
        io.x, io.batch, io.training, io.seed = x.data_ptr(), int(x.shape[0]), 1, int(seed) & (2 ** 64 - 1)
        io.eps = eps.data_ptr() if eps is not None else None
        io.noise = noise.data_ptr() if noise is not None else None
        io.keep_mask = keep_mask.data_ptr() if keep_mask is not None else None
        self._keep = (x, eps, noise, keep_mask)
        self._enter()
        for t in self._keep:
            if t is not None:
                t.record_stream(self.stream)
        self._check(self.lib.mvae_train_step(self.h, C.byref(io), float(r_factor), float(kl_factor), float(lr),
                                             float(clip_norm if clip_norm else 0.0), self._stream()))

    # ------------------------------------------------------------------ train() input pipeline
    def load_dataset(self, x, resident_fraction=0.5):
        """Make x [N,H,W,C] float32 the source of gather_batch().  If it fits in `resident_fraction` of the free HBM
        it is uploaded once and batches are gathered on the device by a HIP kernel from a device permutation; otherwise
        it stays on the host and batches travel through two pinned buffers with asynchronous H2D copies on a copy stream
        (the gather then runs on the host into the pinned buffer)."""
        torch = self.torch
        if not (isinstance(x, np.ndarray) and x.dtype == np.float32 and x.flags.c_contiguous):
            x = np.ascontiguousarray(x, dtype=np.float32)     # (no host copy when the caller's array already fits)
        if tuple(x.shape[1:]) != self.input_dims:
            raise ValueError("dataset must be [N,%d,%d,%d]" % self.input_dims)
        free, _ = torch.cuda.mem_get_info(self.device)
        self.sync()
        ds = dict(n=len(x), row=int(np.prod(x.shape[1:])), resident=x.nbytes <= resident_fraction * free, perm=None)
        if ds["resident"]:
            with torch.cuda.stream(self.stream):
                ds["data"] = torch.empty(x.shape, dtype=torch.float32, device=self.device)
                step = max(1, (256 << 20) // max(x[0].nbytes, 1))
                for i in range(0, len(x), step):
                    ds["data"][i:i + step].copy_(torch.from_numpy(x[i:i + step]), non_blocking=False)
                ds["buf"] = [torch.empty((self.max_batch,) + self.input_dims, dtype=torch.float32, device=self.device)
                             for _ in range(2)]
        else:
            ds["host"] = x
            ds["pin"] = [torch.empty((self.max_batch,) + self.input_dims, dtype=torch.float32).pin_memory()
                         for _ in range(2)]
            ds["buf"] = [torch.empty((self.max_batch,) + self.input_dims, dtype=torch.float32, device=self.device)
                         for _ in range(2)]
            ds["copied"] = [None, None]            # H2D done (recorded on the copy stream)
            ds["consumed"] = [None, None]          # the step that read buf[k] is enqueued (recorded on self.stream)
            if self.copy_stream is None:
                self.copy_stream = torch.cuda.Stream(self.device)
        ds["tick"] = 0
        self.dataset = ds
        return ds

    def set_permutation(self, order):
        """Upload the epoch's sample order (int64 indices into the dataset)."""
        ds, torch = self.dataset, self.torch
        order = np.ascontiguousarray(order, dtype=np.int64)
        if ds["resident"]:
            with torch.cuda.stream(self.stream):
                ds["perm"] = torch.from_numpy(order).to(self.device)
        else:
            ds["perm"] = order

    def input_buffer(self, batch):
        """[batch,H,W,C] float32 view of the handle's own input buffer.  A batch written here and passed to forward() /
        train_step() is read in place: the library otherwise copies every batch into this buffer first (its captured graphs
        need a stable address), an eager launch between two graph replays."""
        p, n, dt = C.c_void_p(), C.c_int64(), C.c_int32()
        self._check_rc(self.lib.mvae_tensor_lookup2(self.h, b"xin", C.byref(p), C.byref(n), C.byref(dt)))
        off = (p.value - self.workspace.data_ptr()) // 4
        if batch > self.max_batch:
            raise ValueError("batch %d exceeds the engine's max_batch %d" % (batch, self.max_batch))
        return self.workspace[off:off + batch * n.value].view((batch,) + tuple(self.input_dims))

    def stage_input(self, x):
        """Copy a batch into the handle's input buffer once and return that view (for a caller that feeds the SAME batch to
        many steps, as a benchmark on synthetic data does; a training loop gathers every batch straight into input_buffer())."""
        v = self.input_buffer(int(x.shape[0]))
        v.copy_(x)
        return v

    def gather_batch(self, start, count):
        """Device tensor [count,H,W,C] = dataset[perm[start:start+count]], ready on self.stream."""
        ds, torch = self.dataset, self.torch
        k = ds["tick"] & 1
        ds["tick"] += 1
        out = ds["buf"][k][:count]
        if ds["resident"]:
            out = self.input_buffer(count)         # gathered straight into the handle's input buffer (same stream as the step)
            idx = ds["perm"][start:start + count]
            self._check_rc(self.lib.mvae_gather_rows(self.device.index, _ptr(ds["data"]), _ptr(idx), count, ds["row"],
                                                     _ptr(out), self._stream()))
            return out
        if ds["copied"][k] is not None:
            ds["copied"][k].synchronize()          # the pinned buffer is free again
        pin = ds["pin"][k][:count]
        np.take(ds["host"], ds["perm"][start:start + count], axis=0, out=pin.numpy())
        if ds["consumed"][k] is not None:
            self.copy_stream.wait_event(ds["consumed"][k])
        with torch.cuda.stream(self.copy_stream):
            out.copy_(pin, non_blocking=True)
            ds["copied"][k] = torch.cuda.Event(); ds["copied"][k].record(self.copy_stream)
        self.stream.wait_event(ds["copied"][k])
        return out

    def batch_consumed(self):
        """Call after the step that read the last gather_batch() result has been enqueued."""
        ds = self.dataset
        if ds is not None and not ds["resident"]:
            k = (ds["tick"] - 1) & 1
            ds["consumed"][k] = self.torch.cuda.Event(); ds["consumed"][k].record(self.stream)

    def _check_rc(self, rc):
        if rc != _abi.MVAE_OK:
            raise MvaeError("mvae error %d" % rc)

    def metrics(self):
        """{count, vae_r_loss, r_exp, vae_kl_loss, kl_scale_i} means of the last forward (synchronises)."""
        m = self._download(self.reduce[self.metrics_off:self.metrics_off + 4 + self.levels]).astype(np.float64)
        n = max(m[0], 1.0)
        out = dict(count=m[0], vae_r_loss=m[1] / n, r_exp=m[2] / n, vae_kl_loss=m[3] / n)
        for s in range(self.levels):
            out["kl_scale_%d" % s] = m[4 + s] / n
        return out

    def reg_loss(self):
        t = self.torch.zeros(1, dtype=self.torch.float32, device=self.device)
        self._enter()
        self._check(self.lib.mvae_reg_loss(self.h, _ptr(t), self._stream()))
        self.sync()
        return float(t.item())

    def decode(self, z):
        torch = self.torch
        B = int(z.shape[0])
        if z.shape[1] != self.Z or z.dtype != torch.float32 or not z.is_contiguous():
            raise ValueError("z must be a contiguous float32 tensor [B,%d]" % self.Z)
        out = torch.empty((B,) + self.input_dims, dtype=torch.float32, device=self.device)
        self._enter()
        self._check(self.lib.mvae_decode(self.h, _ptr(z), B, _ptr(out), self._stream()))
        self.sync()
        return out

    def tensor_nosync(self, name, batch):
        """float32 view of a saved tensor (a bfloat16 tensor is converted: a copy)."""
        p, n, dt = C.c_void_p(), C.c_int64(), C.c_int32()
        rc = self.lib.mvae_tensor_lookup2(self.h, name.encode(), C.byref(p), C.byref(n), C.byref(dt))
        if rc != _abi.MVAE_OK:
            raise KeyError(name)
        off = (p.value - self.workspace.data_ptr()) // 4
        if dt.value == _abi.ACT_DTYPES["bf16"]:
            nfl = (batch * n.value + 1) // 2
            with self.torch.cuda.stream(self.stream):
                raw = self.workspace[off:off + nfl].view(self.torch.bfloat16)[:batch * n.value]
                out = raw.float().view(batch, n.value)
            self.sync()          # the conversion ran on our stream: the caller reads the copy from another one
            return out
        return self.workspace[off:off + batch * n.value].view(batch, n.value)

    def scale_dtypes(self):
        return ["bf16" if self.lib.mvae_scale_dtype(self.h, s) == 1 else "f32" for s in range(self.levels)]

    def tensor(self, name, batch):
        """Debug/parity view of a saved intermediate of the last forward: [batch, elems_per_image]."""
        self.sync()
        return self.tensor_nosync(name, batch)
