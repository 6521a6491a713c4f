"""Exploration noise (reference: offlinerlkit/utils/noise.py:4-13); not used inside learn()."""
import numpy as np


class GaussianNoise:
    def __init__(self, mu=0.0, sigma=1.0):
        self._mu = mu
        self._sigma = sigma

    def __call__(self, size):
        return np.random.normal(self._mu, self._sigma, size)
