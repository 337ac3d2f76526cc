"""Beta-Bernoulli model and the predictive evaluation score — host-side float64, mirrors
nanomotif/model.py:11-92 and find_motifs_bin.py:1360-1379 (same names and argument meaning).
Scores never run on the GPU: they need digamma in float64 and are O(1) per candidate."""
from __future__ import annotations

import ctypes as _C

import numpy as np

_native_psi = None
_scipy_psi = None


def psi(x):
    """digamma.  The model only ever asks for it at positive integers (alpha, beta, alpha + beta = prior + counts): those
    go to libnmscan's nm_psi_posint — scipy.special.psi's value bit for bit (tests/test_native_search.py checks the two
    against each other) without importing SciPy at CLI start-up (0.1-0.3 s); anything else falls back to SciPy."""
    global _native_psi, _scipy_psi
    if isinstance(x, (int, np.integer)) and x >= 1:
        if _native_psi is None:
            from . import _lib
            _native_psi = _lib.load().nm_psi_posint
        out = _C.c_double(0.0)
        if _native_psi(int(x), _C.byref(out)) == 0:
            return out.value
    if _scipy_psi is None:
        from scipy.special import psi as _p
        _scipy_psi = _p
    return _scipy_psi(x)

DEFAULT_PRIOR_ALPHA = 5
DEFAULT_PRIOR_BETA = 5


class BetaBernoulliModel:
    __slots__ = ("_alpha", "_beta", "_alpha_prior", "_beta_prior")

    def __init__(self, alpha=DEFAULT_PRIOR_ALPHA, beta=DEFAULT_PRIOR_BETA):
        self._alpha = alpha
        self._beta = beta
        self._alpha_prior = alpha
        self._beta_prior = beta

    @classmethod
    def from_counts(cls, n_mod, n_nomod):
        m = cls()
        m.update(int(n_mod), int(n_nomod))
        return m

    def __getstate__(self):
        return {k: getattr(self, k) for k in self.__slots__}

    def __setstate__(self, state):
        for k in self.__slots__:
            setattr(self, k, state[k])

    def get_raw_counts(self):
        return self._alpha - self._alpha_prior, self._beta - self._beta_prior

    def update(self, n_positives, n_negatives):
        self._alpha += n_positives
        self._beta += n_negatives

    def reset(self):
        self._alpha, self._beta = self._alpha_prior, self._beta_prior

    def mean(self):
        return self._alpha / (self._alpha + self._beta)

    def variance(self):
        a, b = self._alpha, self._beta
        return (a * b) / ((a + b) ** 2 * (a + b + 1))

    def standard_deviation(self):
        return np.sqrt(self.variance())

    def posterior_predictive(self, n_positives, n_negatives):
        if n_positives + n_negatives == 0:
            return 0.0
        both = psi(self._alpha + self._beta)
        return n_positives * (psi(self._alpha) - both) + n_negatives * (psi(self._beta) - both)

    def posterior_predictive_per_obs(self, n_positives, n_negatives):
        n_new = n_positives + n_negatives
        if n_new == 0:
            return 0.0
        return self.posterior_predictive(n_positives, n_negatives) / n_new

    def __repr__(self):
        return f"BetaBernoulliModel(alpha={self._alpha}, beta={self._beta})"

    __str__ = __repr__


def predictive_evaluation_score(next_model, current_model) -> float:
    """(mean_next / mean_cur) * (ppo_next(alpha_n, beta_n) - ppo_next(alpha_c - alpha_n, beta_c - beta_n))."""
    extra_pos = current_model._alpha - next_model._alpha
    extra_neg = current_model._beta - next_model._beta
    pp_next = next_model.posterior_predictive_per_obs(next_model._alpha, next_model._beta)
    pp_extra = next_model.posterior_predictive_per_obs(extra_pos, extra_neg)
    return (next_model.mean() / current_model.mean()) * (pp_next - pp_extra)
