"""Oracle restatement of the Beta-Bernoulli model and the predictive score
(reference: nanomotif/model.py:8-92, find_motifs_bin.py:1360-1379).  Test infrastructure only."""
from __future__ import annotations

from scipy.special import psi

PRIOR_ALPHA = 5   # model.py:8-9
PRIOR_BETA = 5


class BetaBernoulliModel:
    def __init__(self, alpha=PRIOR_ALPHA, beta=PRIOR_BETA):
        self._alpha, self._beta = alpha, beta
        self._alpha_prior, self._beta_prior = alpha, beta

    def update(self, n_positives, n_negatives):            # model.py:37-39
        self._alpha += n_positives
        self._beta += n_negatives

    def get_raw_counts(self):                              # model.py:34-35
        return self._alpha - self._alpha_prior, self._beta - self._beta_prior

    def mean(self):                                        # model.py:48-49
        return self._alpha / (self._alpha + self._beta)

    def posterior_predictive(self, n_pos, n_neg):          # model.py:78-85
        if n_pos + n_neg == 0:
            return 0.0
        e_log_p = psi(self._alpha) - psi(self._alpha + self._beta)
        e_log_1mp = psi(self._beta) - psi(self._alpha + self._beta)
        return n_pos * e_log_p + n_neg * e_log_1mp

    def posterior_predictive_per_obs(self, n_pos, n_neg):  # model.py:87-92
        n = n_pos + n_neg
        if n == 0:
            return 0.0
        return self.posterior_predictive(n_pos, n_neg) / n

    def __repr__(self):
        return f"BetaBernoulliModel(alpha={self._alpha}, beta={self._beta})"


def predictive_evaluation_score(next_model, current_model) -> float:
    """find_motifs_bin.py:1360-1379."""
    extra_pos = current_model._alpha - next_model._alpha
    extra_neg = current_model._beta - next_model._beta
    pp_next = next_model.posterior_predictive_per_obs(next_model._alpha, next_model._beta)
    pp_extra = next_model.posterior_predictive_per_obs(extra_pos, extra_neg)
    return (next_model.mean() / current_model.mean()) * (pp_next - pp_extra)
