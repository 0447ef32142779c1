"""
MultiscaleVAE -- drop-in for mvae.MultiscaleVAE (reference mvae/multiscale_vae.py:11-587) whose graph runs as
hand-written HIP kernels on MI355X through the C ABI in include/mvae_hip.h.

Same constructor / compile() / train() signatures and the same `encoder`, `decoder`, `model_trainable`
properties (objects with Keras-like .predict/.fit/.summary/.to_json), plus the aliases the reference's own
callers use although the class never defined them: `model_encode`, `model_decode` (mvae/callbacks.py:77-78),
`model_predict` (notebooks/cifar10_notebook.ipynb cell 10).
"""
import json
import logging
import os
import time
from collections import OrderedDict

import numpy as np

from . import schedule as _schedule
from .engine import Engine
from .initializers import init_params, init_state

logger = logging.getLogger("mvae")          # same logger name as mvae/custom_logger.py:7-14


def _norm_block_dict(d, what):
    """basic_block argument checks, layer_blocks.py:918-924."""
    f, k, s = list(d["filters"]), list(d["kernel_size"]), list(d["strides"])
    if len(f) != len(k) or len(f) != len(s) or len(f) <= 0:
        raise ValueError("len(filters) [{0}] should be equal to len(kernel_size) [{1}] and len(strides) [{2}]".format(
            len(f), len(k), len(s)))
    for v in f:
        if v <= 0:
            raise ValueError("Filters should be > 0")          # mobilenetV3_block, layer_blocks.py:586-587
    return {"filters": [int(v) for v in f],
            "kernel_size": [(int(a[0]), int(a[1])) for a in k],
            "strides": [(int(a[0]), int(a[1])) for a in s]}


def shard_batches(order, batch_size, world=1, rank=0):
    """The rank's share of every global batch of one epoch.  `order` is the epoch's sample order; global batch b is
    order[b*batch_size:(b+1)*batch_size] (last partial batch kept, like keras); rank r takes positions
    r*per .. (r+1)*per - 1 of it, per = ceil(len/world), positions past the end wrapping around to the batch's first
    samples -- nothing is dropped, and every rank takes the same number of steps with the same local batch size.
    Returns (local, spans): the rank's indices concatenated, and (offset, count) of each batch inside `local`."""
    order = np.asarray(order, dtype=np.int64)
    local, spans, off = [], [], 0
    for start in range(0, len(order), batch_size):
        idx = order[start:start + batch_size]
        per = -(-len(idx) // world)
        take = idx[(rank * per + np.arange(per)) % len(idx)] if world > 1 else idx
        spans.append((off, len(take)))
        off += len(take)
        local.append(take)
    return (np.concatenate(local) if local else np.zeros(0, np.int64)), spans


class History:
    def __init__(self):
        self.history = OrderedDict()
        self.epoch = []


class _ModelView:
    """What the reference hands out as keras.Model objects (multiscale_vae.py:566-576)."""

    def __init__(self, vae, kind):
        self._vae, self._kind = vae, kind
        self.name = {"trainable": "model", "encoder": "model_encoder", "decoder": "model_decoder"}[kind]

    # ---- inference ---------------------------------------------------------------------------------
    def predict(self, x, batch_size=32, verbose=0, **_):
        vae = self._vae
        x = np.asarray(x, dtype=np.float32)
        outs = []
        for i in range(0, len(x), batch_size):
            xb = x[i:i + batch_size]
            eng = vae._ensure_engine(len(xb))
            if self._kind == "decoder":
                outs.append(eng.decode(eng.to_device(xb)).cpu().numpy())
            else:
                key = "recon" if self._kind == "trainable" else "z"
                seed = vae._next_seed()
                outs.append(eng.forward(eng.to_device(xb), False, seed=seed, outputs=(key,))[key].cpu().numpy())
        if not outs:
            shape = (0, vae._engine.Z) if self._kind == "encoder" else (0,) + tuple(vae._inputs_dims)
            return np.zeros(shape, np.float32)
        return np.concatenate(outs, axis=0)

    def __call__(self, x):
        return self.predict(x, batch_size=max(len(x), 1))

    # ---- training (keras.Model.fit as used at multiscale_vae.py:550-557) ------------------------------
    def fit(self, x, y=None, batch_size=32, shuffle=True, epochs=1, initial_epoch=0, callbacks=None, verbose=1,
            **_):
        if self._kind != "trainable":
            raise RuntimeError("only model_trainable can be fitted")
        return self._vae._fit(np.asarray(x, dtype=np.float32), batch_size, shuffle, epochs, initial_epoch,
                              callbacks or [], verbose)

    # ---- introspection -----------------------------------------------------------------------------
    def _tables(self):
        eng = self._vae._engine
        pt = eng.param_table
        if self._kind == "encoder":
            pt = OrderedDict((k, v) for k, v in pt.items() if k.startswith("enc"))
        elif self._kind == "decoder":
            pt = OrderedDict((k, v) for k, v in pt.items() if k.startswith("dec"))
        return pt

    def count_params(self):
        return int(sum(int(np.prod(v["shape"])) for v in self._tables().values()))

    def summary(self, print_fn=print):
        pt = self._tables()
        print_fn('Model: "%s"' % self.name)
        print_fn("%-52s %-22s %10s  %s" % ("tensor", "shape", "params", "regulariser"))
        for k, v in pt.items():
            print_fn("%-52s %-22s %10d  %s" % (k, str(v["shape"]), int(np.prod(v["shape"])), v["reg"] or "-"))
        print_fn("Total trainable params: {:,}".format(self.count_params()))

    def to_json(self):
        vae = self._vae
        return json.dumps({
            "class_name": "MultiscaleVAE/" + self.name, "backend": "mvae-mi355x-hip",
            "config": vae.get_config(),
            "tensors": [{"name": k, "shape": list(v["shape"]), "regularizer": v["reg"]}
                        for k, v in self._tables().items()]})

    def get_weights(self):
        w = self._vae.get_weights()
        return [w[k] for k in self._tables()]


class MultiscaleVAE:
    def __init__(
            self,
            input_dims,
            z_dims,
            compress_output=False,
            encoder={
                "filters": [32],
                "kernel_size": [(3, 3)],
                "strides": [(1, 1)]
            },
            decoder=None,
            min_value=0.0,
            max_value=255.0,
            sample_std=0.01,
            channels_index=2,
            *, seed=42, max_batch=None, device=None, act_dtype="f32"):
        # --- argument checking (multiscale_vae.py:34-38)
        if encoder is None:
            raise ValueError("encoder cannot be None")
        if not all(i > 0 for i in z_dims):
            raise ValueError("z_dims elements should be > 0")
        if channels_index != 2:
            raise ValueError("only channels-last HxWxC inputs (channels_index=2) are supported")
        # --- decoder is reverse encoder (:40-45)
        if decoder is None:
            decoder = {
                "filters": encoder["filters"][::-1],
                "strides": encoder["strides"][::-1],
                "kernel_size": encoder["kernel_size"][::-1]
            }
        self._name = "mvae"
        self._levels = len(z_dims)
        self._z_latent_dims = [int(z) for z in z_dims]
        self._inputs_dims = tuple(int(v) for v in input_dims)
        self._encoder_config = _norm_block_dict(encoder, "encoder")
        self._decoder_config = _norm_block_dict(decoder, "decoder")
        self._compress_output = compress_output
        self._min_value = float(min_value)
        self._max_value = float(max_value)
        self._sample_std = float(sample_std)
        self._channels_index = channels_index
        self._output_channels = self._inputs_dims[channels_index]
        self._learning_rate = None
        self._compiled = None
        self._seed = int(seed)
        self._step = 0
        self._device = device
        self._act_dtype = act_dtype            # "f32" | "bf16" activation storage (BASELINE configs 4-5 use bf16)
        self._engine = None
        logger.info("Building multiscale VAE plan")
        self._engine = self._make_engine(int(max_batch) if max_batch else 1)     # host-only: validates + tables
        self._weights = init_params(self._engine.param_table, seed=self._seed)  # glorot_normal, :61
        self._accum = None
        self._state = init_state(self._engine.state_table)
        self._model_trainable = _ModelView(self, "trainable")
        self._model_encoder = _ModelView(self, "encoder")
        self._model_decoder = _ModelView(self, "decoder")

    # ==========================================================================
    def _make_engine(self, max_batch):
        return Engine(self._inputs_dims, self._z_latent_dims, self._encoder_config, self._decoder_config,
                      self._min_value, self._max_value, self._sample_std, max_batch, act_dtype=self._act_dtype)

    def _sync_host(self):
        """Pull weights / optimiser state back from a bound engine."""
        eng = self._engine
        if eng is not None and eng.bound:
            self._weights, self._accum, self._state = eng.get_params(), eng.get_accum(), eng.get_state()

    def _ensure_engine(self, batch):
        eng = self._engine
        if eng.bound and batch <= eng.max_batch:
            return eng
        if eng.bound:
            self._sync_host()
        if eng.bound or batch > eng.max_batch:      # a bound handle cannot be re-bound: make a fresh plan
            new_max = max(batch, eng.max_batch)
            eng.close()
            eng = self._make_engine(new_max)
        eng.bind(self._device)
        eng.set_params(self._weights)
        if self._accum is not None:
            eng.set_accum(self._accum)
        eng.set_state(self._state)
        self._engine = eng
        return eng

    def _next_seed(self):
        """Device-RNG seed of the next call: a function of (seed, step, rank) -- data-parallel replicas draw independent
        epsilon / GaussianNoise / SpatialDropout for their shards (weight initialisation stays rank-independent)."""
        self._step += 1
        rank = int(getattr(self, "_rank", 0))
        return ((self._seed * 0x9E3779B97F4A7C15 + self._step) ^ (rank * 0xD1B54A32D192ED03)) & (2 ** 63 - 1)

    def get_config(self):
        return dict(input_dims=list(self._inputs_dims), z_dims=list(self._z_latent_dims),
                    encoder=self._encoder_config, decoder=self._decoder_config, min_value=self._min_value,
                    max_value=self._max_value, sample_std=self._sample_std)

    def get_weights(self):
        self._sync_host()
        return self._weights

    def set_weights(self, weights):
        self._weights = OrderedDict((k, np.asarray(weights[k], np.float32)) for k in self._engine.param_table)
        if self._engine.bound:
            self._engine.set_params(self._weights)

    # ==========================================================================
    def compile(self,
                learning_rate,
                r_loss_factor=1.0,
                kl_loss_factor=1.0,
                clip_norm=1.0):
        """multiscale_vae.py:437-504: loss = r_factor * vae_r_experimental_loss + kl_factor * KL, optimiser
        Adagrad(lr, clipnorm) -- all of it lives in the HIP backward / optimiser kernels."""
        self.learning_rate = learning_rate
        self._compiled = dict(r_loss_factor=float(r_loss_factor), kl_loss_factor=float(kl_loss_factor),
                              clip_norm=clip_norm)

    # ==========================================================================
    def train_on_batch(self, x, eps=None, noise=None, keep_mask=None):
        """One optimiser step on one (local) batch; asynchronous on the device."""
        if self._compiled is None:
            raise RuntimeError("compile() must be called before training")
        x = np.asarray(x, np.float32) if not hasattr(x, "data_ptr") else x
        eng = self._ensure_engine(int(x.shape[0]))
        xd = x if hasattr(x, "data_ptr") else eng.to_device(x)
        dev = [None if v is None else (v if hasattr(v, "data_ptr") else eng.to_device(v)) for v in (eps, noise, keep_mask)]
        c = self._compiled
        eng.train_step(xd, self._learning_rate, c["r_loss_factor"], c["kl_loss_factor"], c["clip_norm"],
                       eps=dev[0], noise=dev[1], keep_mask=dev[2], seed=self._next_seed())
        return eng

    def _fit(self, x, batch_size, shuffle, epochs, initial_epoch, callbacks, verbose):
        """keras.Model.fit as the reference calls it (multiscale_vae.py:550-557).  The dataset is made device
        resident once (Engine.load_dataset) and every batch is gathered on the device from the epoch's permutation, so
        the host only enqueues work; under data parallelism rank r takes every world-th share of each global batch, the
        tail of a batch that does not divide wraps around to the batch's first samples (nothing is dropped, a few
        samples of that batch count twice) so that every rank enqueues the same number of all-reduces."""
        if self._compiled is None:
            raise RuntimeError("compile() must be called before fit()")
        import torch
        dist = torch.distributed
        world, rank = 1, 0
        if dist.is_available() and dist.is_initialized():
            world, rank = dist.get_world_size(), dist.get_rank()
        self._rank = rank
        n = len(x)
        hist = History()
        c = self._compiled
        for cb in callbacks:
            if hasattr(cb, "set_vae"):
                cb.set_vae(self)
        if n == 0:
            return hist
        per_max = -(-min(batch_size, n) // world)
        # callbacks predict on a handful of images (the intermediate-results callback: 16) between training steps: size the
        # engine for them up front, or the first such call would re-plan and re-bind it mid-epoch (every captured graph and
        # the resident dataset dropped and rebuilt)
        need = max([per_max] + [int(getattr(cb, "predict_batch", 0)) for cb in callbacks])
        fed = None
        eng = self._ensure_engine(need)
        eng.load_dataset(x)
        for epoch in range(initial_epoch, epochs):
            for cb in callbacks:
                if hasattr(cb, "on_epoch_begin"):
                    cb.on_epoch_begin(epoch, {})
            # the shuffle of epoch e depends on (seed, e) only: a run resumed with initial_epoch=e sees the same batches
            order = np.random.default_rng([self._seed + 1, epoch]).permutation(n) if shuffle else np.arange(n)
            local, spans = shard_batches(order, batch_size, world, rank)
            t0 = time.time()
            acc, seen, fed = None, 0, None
            for bi, (off, cnt) in enumerate(spans):
                eng = self._ensure_engine(need)
                if fed is not eng:                              # first batch, or a callback re-bound the engine
                    if eng.dataset is None:
                        fed = None                              # drop the old engine (and its resident dataset) first
                        eng.load_dataset(x)
                    eng.set_permutation(local)
                    fed = eng
                xb = eng.gather_batch(off, cnt)
                eng.train_step(xb, self._learning_rate, c["r_loss_factor"], c["kl_loss_factor"], c["clip_norm"],
                               seed=self._next_seed())
                eng.batch_consumed()
                with torch.cuda.stream(eng.stream):            # stays on the device: no per-step sync
                    m = eng.reduce[eng.metrics_off:eng.metrics_off + 4 + self._levels]
                    acc = m.clone() if acc is None else acc.add_(m)
                seen += 1
                for cb in callbacks:
                    if hasattr(cb, "on_batch_end"):
                        cb.on_batch_end(bi, {})
            logs = {}
            if acc is not None:
                eng.sync()
                dt = max(time.time() - t0, 1e-9)
                a = acc.cpu().numpy().astype(np.float64)
                cnt = max(a[0], 1.0)
                logs = {"vae_r_loss": a[1] / cnt, "vae_kl_loss": a[3] / cnt}
                reg = eng.reg_loss()
                logs["loss"] = c["r_loss_factor"] * a[2] / cnt + c["kl_loss_factor"] * a[3] / cnt + reg
                logs["images_per_sec"] = a[0] / dt
                logs["samples"] = a[0]
            hist.epoch.append(epoch)
            for k, v in logs.items():
                hist.history.setdefault(k, []).append(float(v))
            if verbose and rank == 0:
                logger.info("Epoch %d/%d - %s", epoch + 1, epochs,
                            " - ".join("%s: %.4f" % kv for kv in logs.items()))
            for cb in callbacks:
                if hasattr(cb, "on_epoch_end"):
                    cb.on_epoch_end(epoch, logs)
        return hist

    def train(self,
              x_train,
              batch_size,
              epochs,
              run_folder,
              print_every_n_batches=100,
              initial_epoch=0,
              step_size=1,
              lr_decay=1,
              save_checkpoint_weights=False):
        """multiscale_vae.py:508-557: step-decay LR schedule, the intermediate-results image callback on
        x_train[0:16] (callbacks.py), optional per-epoch checkpoints, then fit(x, x, shuffle=True).
        `run_folder=None` (an extension) trains without touching the file system."""
        lr_schedule = _schedule.step_decay_schedule(
            initial_lr=self._learning_rate,
            decay_factor=lr_decay,
            step_size=step_size)
        callbacks_fns = [lr_schedule]
        if run_folder is not None:
            from .callbacks import SaveIntermediateResultsCallback
            os.makedirs(run_folder, exist_ok=True)
            callbacks_fns.append(SaveIntermediateResultsCallback(
                run_folder, print_every_n_batches, initial_epoch, np.asarray(x_train[0:16], np.float32), self))
            weights_path = os.path.join(run_folder, "weights")
            os.makedirs(weights_path, exist_ok=True)
            if save_checkpoint_weights:
                callbacks_fns.append(_Checkpoint(self, weights_path))
        return self._model_trainable.fit(
            x_train,
            x_train,
            batch_size=batch_size,
            shuffle=True,
            epochs=epochs,
            initial_epoch=initial_epoch,
            callbacks=callbacks_fns)

    fit = train          # BASELINE.json speaks of a fit()/predict() surface

    def predict(self, x, batch_size=32):
        return self._model_trainable.predict(x, batch_size=batch_size)

    # ==========================================================================
    def save_weights(self, filename):
        self._sync_host()
        blob = {"w/" + k: v for k, v in self._weights.items()}
        blob.update({"s/" + k: v for k, v in self._state.items()})
        if self._accum is not None:
            blob.update({"a/" + k: v for k, v in self._accum.items()})
        blob["meta/step"] = np.asarray(self._step, dtype=np.int64)      # position of the device RNG stream
        np.savez(filename, **blob)

    def load_weights(self, filename):
        """The reference's load_weights is an empty stub (multiscale_vae.py:561-562); here an .npz written by
        save_weights is restored, anything else is ignored like the reference does."""
        if not (isinstance(filename, str) and filename.endswith(".npz") and os.path.exists(filename)):
            return
        with np.load(filename) as f:
            self._weights = OrderedDict((k, f["w/" + k]) for k in self._engine.param_table)
            self._state = OrderedDict((k, f["s/" + k]) for k in self._engine.state_table)
            if all(("a/" + k) in f for k in self._engine.param_table):
                self._accum = OrderedDict((k, f["a/" + k]) for k in self._engine.param_table)
            if "meta/step" in f:
                self._step = int(f["meta/step"])
        if self._engine.bound:
            self._engine.set_params(self._weights)
            self._engine.set_state(self._state)
            if self._accum is not None:
                self._engine.set_accum(self._accum)

    # ==========================================================================
    @property
    def encoder(self):
        return self._model_encoder

    @property
    def decoder(self):
        return self._model_decoder

    @property
    def model_trainable(self):
        return self._model_trainable

    model_encode = encoder
    model_decode = decoder
    model_predict = model_trainable

    @property
    def learning_rate(self):
        return self._learning_rate

    @learning_rate.setter
    def learning_rate(self, value):
        self._learning_rate = value

    def normalize(self, v):
        return (v - self._min_value) / (self._max_value - self._min_value)


class _Checkpoint:
    """keras.callbacks.ModelCheckpoint(save_weights_only=True) equivalent (multiscale_vae.py:531-548), .npz."""

    def __init__(self, vae, path):
        self.vae, self.path = vae, path

    def on_epoch_end(self, epoch, logs=None):
        loss = (logs or {}).get("loss", float("nan"))
        self.vae.save_weights(os.path.join(self.path, "weights-{:03d}-{:.2f}.npz".format(epoch + 1, loss)))
        self.vae.save_weights(os.path.join(self.path, "weights.npz"))
