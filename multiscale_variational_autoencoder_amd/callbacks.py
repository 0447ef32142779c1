"""Intermediate-result image dumps during training: the callback `MultiscaleVAE.train` installs in the reference
(mvae/multiscale_vae.py:518-523, mvae/callbacks.py:16-138).  In the reference it cannot run (it imports a `collage`
helper that does not exist and calls `model_encode` / `model_decode`, names the class never defines); this is a
working restatement of what it is written to do, on top of the HIP inference path (`encoder.predict` = mvae_forward
with training = 0, `decoder.predict` = mvae_decode).  Every `print_every_n_batches` batches it writes three PNG
collages of the 16 monitored images into <run_folder>/images:

    img_<epoch>_<batch>.png             decode(encode(images))
    samples_<epoch>_<batch>.png         decode(z), z ~ N(mean(encodings), std(encodings)) (statistics over all elements)
    interpolations_<epoch>_<batch>.png  decode of linear blends between consecutive encodings

No third-party imaging library is needed: nearest-neighbour resize and a zlib PNG writer are a few lines each."""
import logging
import math
import os
import struct
import zlib

import numpy as np

logger = logging.getLogger("mvae")


def collage(images):
    """[N, H, W, C] -> one [gh*H, gw*W, C] image, row-major on a near-square grid (C == 1 gives a 2-D image)."""
    x = np.asarray(images)
    if x.ndim == 3:
        x = x[..., None]
    n, h, w, c = x.shape
    gw = int(math.ceil(math.sqrt(max(n, 1))))
    gh = int(math.ceil(n / gw)) if n else 1
    out = np.zeros((gh * h, gw * w, c), x.dtype)
    for k in range(n):
        r, q = divmod(k, gw)
        out[r * h:(r + 1) * h, q * w:(q + 1) * w] = x[k]
    return out[..., 0] if c == 1 else out


def resize_nearest(img, shape):
    """order-0 resize (skimage.transform.resize(..., order=0) in the reference): nearest source pixel centre."""
    h, w = img.shape[:2]
    oh, ow = int(shape[0]), int(shape[1])
    ys = np.minimum(((np.arange(oh) + 0.5) * h / oh).astype(np.int64), h - 1)
    xs = np.minimum(((np.arange(ow) + 0.5) * w / ow).astype(np.int64), w - 1)
    return img[ys][:, xs]


def save_png(path, img01):
    """img01: float image in [0, 1], [H, W] (written inverted, matplotlib's "gray_r") or [H, W, 3]."""
    a = np.clip(np.asarray(img01, np.float64), 0.0, 1.0)
    if a.ndim == 2:
        a = 1.0 - a
    u8 = np.rint(a * 255.0).astype(np.uint8)
    if u8.ndim == 3 and u8.shape[2] not in (1, 3):
        u8 = u8[..., :3] if u8.shape[2] > 3 else np.repeat(u8[..., :1], 3, axis=2)
    if u8.ndim == 3 and u8.shape[2] == 1:
        u8 = u8[..., 0]
    h, w = u8.shape[:2]
    color_type = 0 if u8.ndim == 2 else 2
    raw = b"".join(b"\x00" + u8[r].tobytes() for r in range(h))

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)

    png = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0)) +
           chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
    with open(path, "wb") as f:
        f.write(png)


class SaveIntermediateResultsCallback:
    """Same constructor and hooks as the reference's keras Callback (mvae/callbacks.py:18-43, 66, 137)."""

    def __init__(self, run_folder, print_every_n_batches, initial_epoch, images, vae, resize_shape=(256, 256)):
        self._vae = vae
        self._images = np.asarray(images, np.float32)
        self.predict_batch = int(self._images.shape[0])     # fit() sizes the engine for this (no mid-epoch re-plan)
        self._epoch = initial_epoch
        self._run_folder = run_folder
        self._resize_shape = resize_shape
        self._print_every_n_batches = max(int(print_every_n_batches), 1)
        self._images_path = os.path.join(self._run_folder, "images")
        os.makedirs(self._images_path, exist_ok=True)

    def save_collage(self, samples, batch, prefix):
        x = self._vae.normalize(np.asarray(samples, np.float32))          # to [0, 1]
        x = resize_nearest(collage(x), self._resize_shape)
        path = os.path.join(self._images_path, "%s_%s_%d.png" % (prefix, str(self._epoch).zfill(3), batch))
        save_png(path, x)
        return path

    def on_batch_end(self, batch, logs=None):
        if batch % self._print_every_n_batches != 0 or len(self._images) == 0:
            return
        n = self._images.shape[0]
        # ---- encode -> decode
        encodings = self._vae.model_encode.predict(self._images, batch_size=n)
        decodings = self._vae.model_decode.predict(encodings, batch_size=n)
        self.save_collage(decodings, batch, "img")
        # ---- decode random latents drawn around the encodings' overall mean / spread
        mean, std = float(np.mean(encodings)), float(np.std(encodings))
        logger.info("encodings_mean: %.4g, encodings_std: %.4g", mean, std)
        rng = np.random.default_rng([int(getattr(self._vae, "_seed", 0)), int(self._epoch), int(batch)])
        z = rng.normal(mean, std, size=encodings.shape).astype(np.float32)
        self.save_collage(self._vae.model_decode.predict(z, batch_size=n), batch, "samples")
        # ---- decode linear blends between consecutive encodings: row j walks from encoding j to encoding j+1
        inter = np.zeros_like(encodings)
        side = int(round(math.sqrt(n)))
        for j in range(side):
            if j + 1 >= n:
                break
            a, b = encodings[j], encodings[j + 1]
            for i in range(side):
                k = j * side + i
                if k >= n:
                    continue
                t = float(i) / float(side - 1) if side > 1 else 0.0
                inter[k] = a * (1.0 - t) + b * t
        self.save_collage(self._vae.model_decode.predict(inter, batch_size=n), batch, "interpolations")

    def on_epoch_begin(self, epoch, logs=None):
        self._epoch += 1
