#!/usr/bin/env python3
"""images/s of MultiscaleVAE.train() (the reference's own entry point: device-resident dataset, shuffled batches gathered on the
device, step-decay schedule, metrics) next to bench.py's bare train step:  python tools/fit_rate.py [batch] [n_images] [epochs]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS
from mvae import MultiscaleVAE
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 51200
E = int(sys.argv[3]) if len(sys.argv) > 3 else 3
w = WORKLOADS["c32nb"]
x = np.random.default_rng(0).uniform(0, 255, (N,) + tuple(w["input_dims"])).astype(np.float32)
vae = MultiscaleVAE(input_dims=tuple(w["input_dims"]), z_dims=w["z_dims"], encoder=w["encoder"], decoder=w["decoder"])
vae.compile(learning_rate=1e-3, r_loss_factor=1000, kl_loss_factor=10)
h = vae.train(x, batch_size=B, epochs=E, run_folder=None)
print("train(): batch %d, %d images, images/s per epoch:" % (B, N), ["%.0f" % v for v in h.history["images_per_sec"]],
      "loss", ["%.3f" % v for v in h.history["loss"]])
