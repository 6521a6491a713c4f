"""StandardScaler used by TD3+BC at action-selection time (reference: offlinerlkit/utils/scaler.py:6-60)."""
import os.path as path

import numpy as np


class StandardScaler:
    def __init__(self, mu=None, std=None):
        self.mu = mu
        self.std = std

    def fit(self, data):
        self.mu = np.mean(data, axis=0, keepdims=True)
        self.std = np.std(data, axis=0, keepdims=True)
        self.std[self.std < 1e-12] = 1.0

    def transform(self, data):
        return (data - self.mu) / self.std

    def inverse_transform(self, data):
        return self.std * data + self.mu

    def save_scaler(self, save_path):
        np.save(path.join(save_path, "mu.npy"), self.mu)
        np.save(path.join(save_path, "std.npy"), self.std)

    def load_scaler(self, load_path):
        self.mu = np.load(path.join(load_path, "mu.npy"))
        self.std = np.load(path.join(load_path, "std.npy"))

    def transform_tensor(self, data):
        import torch
        return (data - torch.as_tensor(self.mu, device=data.device)) / torch.as_tensor(self.std, device=data.device)
