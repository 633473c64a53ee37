"""Parameter estimators used by ``BaseReducedBasis.set`` (reference: src/lib/Estimators.py:24-37).

O(n*k) host arithmetic on the tiny (n_basis, M) coefficient matrices -- out of scope of the GPU hot
path; only the two estimators the basis objects instantiate exist here.
"""
import numpy as np


class Estimator:
    def __init__(self, a_values_base):
        self.a_values_base = a_values_base

    def fit(self, c_values, a_values):
        return self

    def estimate_parameter(self, c_values):
        raise Exception("Not implemented.")


class EstimatorLinear(Estimator):
    """a_est = sum_b c[b, i] * a_base[b]  (:24-27)."""

    def estimate_parameter(self, c_values):
        return np.tensordot(np.asarray(c_values).T, np.asarray(self.a_values_base), axes=(1, 0))


class EstimatorInv(Estimator):
    """1 / a_est = sum_b c[b, i] / a_base[b]  (:30-37)."""

    def __init__(self, a_values_base):
        super().__init__(a_values_base)
        self.inv_a_values_base = 1.0 / np.array(self.a_values_base)

    def estimate_parameter(self, c_values):
        return 1.0 / np.tensordot(np.asarray(c_values).T, self.inv_a_values_base, axes=(1, 0))
