"""
Least-squares backend on top of the batched residual kernel (BASELINE.json config 5, SURVEY.md
section 8 row f2).  The reference has no such backend -- its "least-squares fit" is the swarm
minimising the RMSE (README.md:3, nmrfit/utils.py:176) -- so this is a build-defined extension
that reuses the hot path: one ``nmrfit_residual_batch`` launch returns the residual vectors of
the D+1 parameter rows a forward-difference Jacobian needs.

    fun(x) = weights*(V_data - V_fit) / sqrt(N)        so that ||fun(x)||_2 == objective(x)
    jac(x) = [fun(x + h_i e_i) - fun(x)] / h_i         D+1 rows, one launch

``least_squares`` hands both to scipy.optimize.least_squares (trust-region reflective, box
bounds); ``polish`` refines a swarm result.  scipy runs on the host; every residual and
Jacobian column comes from the GPU.
"""
import numpy as np

from . import _cabi

_SQRT_EPS = float(np.sqrt(np.finfo(np.float64).eps))


class ResidualModel:
    """fun / jac callables over an ``equations.Evaluator``."""

    def __init__(self, evaluator, lower=None, upper=None, rel_step=_SQRT_EPS):
        self.ev = evaluator
        self.N = evaluator.N
        self.lower = None if lower is None else _cabi.f64(lower)
        self.upper = None if upper is None else _cabi.f64(upper)
        self.rel_step = rel_step
        self.n_fun = 0
        self.n_jac = 0
        self._scale = 1.0 / np.sqrt(self.N)

    def fun(self, x):
        self.n_fun += 1
        return self.ev.residual_batch(np.asarray(x, dtype=np.float64))[0] * self._scale

    def steps(self, x):
        """Forward steps h_i = rel_step*max(1,|x_i|), flipped where x_i + h_i would leave the
        box (scipy's '2-point' rule)."""
        x = np.asarray(x, dtype=np.float64)
        h = self.rel_step * np.maximum(1.0, np.abs(x))
        if self.upper is not None:
            flip = x + h > self.upper
            if self.lower is not None:
                flip &= (x - h >= self.lower)
            h = np.where(flip, -h, h)
        return h

    def rows(self, x):
        x = np.asarray(x, dtype=np.float64)
        h = self.steps(x)
        rows = np.tile(x, (x.size + 1, 1))
        idx = np.arange(x.size)
        rows[idx + 1, idx] += h
        # the step actually taken after rounding
        return rows, rows[idx + 1, idx] - x

    def jac(self, x):
        self.n_jac += 1
        rows, h = self.rows(x)
        R = self.ev.residual_batch(rows)                    # one launch: (D+1) x N
        J = (R[1:] - R[0]) * (self._scale / h[:, None])    # D x N
        return np.ascontiguousarray(J.T)                    # N x D

    def objective(self, x):
        return float(np.linalg.norm(self.fun(x)))


def least_squares(evaluator, x0, lower, upper, **kwargs):
    """scipy.optimize.least_squares with GPU residuals / Jacobian.  Returns the scipy result;
    ``result.cost`` is 0.5*objective**2 and ``result.objective`` the RMSE the swarm minimises."""
    from scipy.optimize import least_squares as _ls
    lower, upper = _cabi.f64(lower), _cabi.f64(upper)
    model = ResidualModel(evaluator, lower, upper)
    x0 = np.clip(np.asarray(x0, dtype=np.float64), lower, upper)
    kwargs.setdefault("method", "trf")
    kwargs.setdefault("x_scale", np.maximum(upper - lower, 1e-12))
    res = _ls(model.fun, x0, jac=model.jac, bounds=(lower, upper), **kwargs)
    res.objective = float(np.sqrt(2.0 * res.cost))
    res.n_residual_launches = model.n_fun + model.n_jac
    return res


def polish(evaluator, x_swarm, lower, upper, fit_im=False, **kwargs):
    """Refine a swarm result; keeps it if the least-squares step does not improve on it.

    The residual rows are the REAL-part residual only.  ``fit_im`` is the mode the swarm
    minimised: acceptance and the returned value use that same objective
    (``objective_batch(x, fit_im=...)``), so a step that lowers the real-part RMSE but raises the
    imaginary term is rejected and the meaning of the returned error never changes."""
    x_swarm = np.asarray(x_swarm, dtype=np.float64)
    res = least_squares(evaluator, x_swarm, lower, upper, **kwargs)
    f0, f1 = (float(f) for f in evaluator.objective_batch(np.stack([x_swarm, res.x]), fit_im=fit_im))
    if f1 <= f0:
        return res.x, f1, res
    return x_swarm, f0, res
