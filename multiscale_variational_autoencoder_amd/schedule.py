"""Learning-rate step decay of the reference (mvae/schedule.py:7-21), as a plain host-side scalar."""
import numpy as np


def step_decay(initial_lr, decay_factor=0.5, step_size=1):
    """Returns schedule(epoch) = initial_lr * decay_factor ** floor(epoch / step_size)  (schedule.py:17-19)."""
    def schedule(epoch):
        return initial_lr * (decay_factor ** np.floor(epoch / step_size))
    return schedule


class LearningRateScheduler:
    """Minimal stand-in for keras.callbacks.LearningRateScheduler: sets model.learning_rate at epoch begin."""
    def __init__(self, schedule):
        self.schedule = schedule
        self.vae = None

    def set_vae(self, vae):
        self.vae = vae

    def on_epoch_begin(self, epoch, logs=None):
        self.vae.learning_rate = float(self.schedule(epoch))


def step_decay_schedule(initial_lr, decay_factor=0.5, step_size=1):
    """Same name/arguments as the reference's wrapper (schedule.py:7-21)."""
    return LearningRateScheduler(step_decay(initial_lr, decay_factor, step_size))
