"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  **parity unpinned.**

The reference's gradient, restated: `Zygote.gradient` through
`solve(prob, ROCK4(); saveat=t_train, reltol=1f-3, sensealg=InterpolatingAdjoint(autojacvec=ZygoteVJP()))`
(wind_mixing/src/NDE_training.jl:291,304,327-333; free convection: Tsit5/ROCK4 at reltol 1e-4 with
`InterpolatingAdjoint(checkpointing=true)`, free_convection/src/solve.jl:1-6) is a *continuous* adjoint on an *adaptive*
solve — DiffEqSensitivity 6.45.0, pinned in wind_mixing/Manifest.toml, absent from /root/reference; SURVEY Appendix B
states its published algorithm:

    forward :  x(t) by an adaptive embedded Runge-Kutta pair, dense output; sol[n] = x(t_n) at the `saveat` times
    backward:  dλ/dt = -(∂f/∂x)ᵀ λ ,  dμ/dt = -(∂f/∂p)ᵀ λ   from t_end to t_0, x(t) read from the forward interpolant,
               λ += ∂L/∂sol[n] at every `saveat` time t_n (a jump);   ∇_p L = μ(t_0)
    tolerances of the backward solve = those of the forward solve (abstol 1e-6 default, reltol as given)

This module restates that with SciPy's `solve_ivp` (RK45 = Dormand-Prince 5(4) standing in for ROCK4/Tsit5: SciPy has no
stabilised Chebyshev method; every embedded pair at the same tolerance tracks the same continuous problem to that tolerance)
so that the gap between the product's *discrete* RK4 adjoint and "the reference CPU DiffEqFlux solve" can be MEASURED instead
of assumed (tests/test_oracle.py::test_discrete_vs_continuous_adjoint_gradient_gap).  One column (simulation)
at a time, as the reference's list comprehension over simulations does (NDE_training.jl:291).

Only `tests/` and tools that feed DESIGN.md import this module.
"""
from __future__ import annotations

import numpy as np
from scipy.integrate import solve_ivp

from . import nde_oracle as O


def _flat_grads(grads):
    return O.pack_grads(grads)


def loss_and_grad_continuous(cfg, x0, bcs, theta, truth, scalings, rtol=1e-3, atol=1e-6, method="RK45", max_step=np.inf):
    """(total, scaled terms[6], grad[n_params], sol[n_col, n_save, n_state], stats) by the continuous interpolating adjoint."""
    m = O.Model(cfg, np.float64)
    nets = m.unpack(theta)
    x0 = np.asarray(x0, np.float64)
    bcs = np.asarray(bcs, np.float64)
    truth = np.asarray(truth, np.float64)
    scalings = np.asarray(scalings, np.float64)
    ts = np.asarray(cfg.save_times, np.float64)
    n_col, n_state, n_save = x0.shape[0], x0.shape[1], len(ts)
    n_par = len(theta)

    sol = np.zeros((n_col, n_save, n_state))
    dense = []
    nfev_f = 0
    for c in range(n_col):
        bc = bcs[c:c + 1]
        f = lambda t, x: m.rhs(x[None], bc, nets, t)[0]
        r = solve_ivp(f, (ts[0], ts[-1]), x0[c], method=method, t_eval=ts, rtol=rtol, atol=atol, dense_output=True, max_step=max_step)
        assert r.success, r.message
        sol[c] = r.y.T
        dense.append(r.sol)
        nfev_f += r.nfev
    total, terms = O.loss(cfg, sol, truth, scalings)

    grad = np.zeros(n_par)
    nfev_b = 0
    for c in range(n_col):
        bc = bcs[c:c + 1]
        xs = dense[c]

        def aug(t, y):
            lam = y[:n_state]
            _, vjp = m.rhs(xs(t)[None], bc, nets, t, True)
            xbar, g = vjp(lam[None])
            return np.concatenate([-xbar[0], -_flat_grads(g)])

        y = np.zeros(n_state + n_par)
        for n in range(n_save - 1, 0, -1):
            y[:n_state] += O._loss_injection(cfg, sol[c:c + 1, n], truth[c:c + 1, n], scalings, n_col, n_save)[0]
            r = solve_ivp(aug, (ts[n], ts[n - 1]), y, method=method, rtol=rtol, atol=atol, max_step=max_step)
            assert r.success, r.message
            y = r.y[:, -1]
            nfev_b += r.nfev
        grad += y[n_state:]
    return total, terms, grad, sol, dict(nfev_forward=nfev_f, nfev_backward=nfev_b)
