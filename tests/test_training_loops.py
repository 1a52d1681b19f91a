"""Host-side optimiser loops against literal restatements of the third-party loops the reference calls (no GPU: the
"problem" is an analytic stand-in with the same `grad_loss` interface).

GalacticOptim 1.2.0 `__solve(prob, opt::Flux.Optimise.AbstractOptimiser; maxiters, cb, save_best=true)` and Flux 0.11.6
`ADAM` / `Flux.train!` are pinned in wind_mixing/Manifest.toml / free_convection/Manifest.toml and absent from
/root/reference; they are restated here from their published source (call sites: NDE_training.jl:340-372, training.jl:71)."""
import numpy as np

from colnde.flux_compat import ADAM
from colnde.wind_mixing import train_NDE
from colnde.free_convection import train_neural_differential_equation


class Quadratic:
    """loss(θ) = ½ (θ-c)ᵀ A (θ-c) with an ill-conditioned A: ADAM overshoots, so best-loss and last iterate differ."""
    loss_scalings = np.ones(6)

    def __init__(self, n=12, seed=0):
        r = np.random.default_rng(seed)
        self.A = np.diag(np.logspace(-1, 2, n))
        self.c = r.standard_normal(n)

    def grad_loss(self, theta):
        d = np.asarray(theta, np.float64) - self.c
        return float(0.5 * d @ self.A @ d), dict(u=0.0), (self.A @ d)


class FluxADAM:
    """Flux 0.11.6 `ADAM`: `state::IdDict`; `apply!(o, x, Δ)`: `mt, vt, βp = get!(o.state, x, (zero(x), zero(x), o.beta))`,
    `mt = β1 mt + (1-β1)Δ; vt = β2 vt + (1-β2)Δ²; Δ = mt/(1-βp1)/(√(vt/(1-βp2))+ϵ)·η; o.state[x] = (mt, vt, βp.*β)`."""

    def __init__(self, eta, beta=(0.9, 0.999)):
        self.eta, self.beta, self.state = eta, beta, {}

    def apply(self, x, delta):
        mt, vt, bp = self.state.get(id(x), (np.zeros_like(x), np.zeros_like(x), self.beta))
        mt = self.beta[0] * mt + (1 - self.beta[0]) * delta
        vt = self.beta[1] * vt + (1 - self.beta[1]) * delta * delta
        out = mt / (1 - bp[0]) / (np.sqrt(vt / (1 - bp[1])) + 1e-8) * self.eta
        self.state[id(x)] = (mt, vt, (bp[0] * self.beta[0], bp[1] * self.beta[1]))
        return out


def galactic_solve(f, u0, opt, maxiters, keep=None, cb=None):
    """GalacticOptim 1.2.0 `__solve` for a Flux optimiser, save_best = true."""
    theta = np.array(u0, dtype=np.float64)              # θ = copy(prob.u0): a NEW array ⇒ a new IdDict key
    if keep is not None:
        keep.append(theta)                              # (keeps ids unique for the lifetime of the test)
    min_err, min_theta = np.inf, None
    for i in range(1, maxiters + 1):
        x, _, g = f(theta)
        if cb is not None and cb(theta, x):             # cb_call = cb(θ, x...); elseif cb_call break
            break
        theta -= opt.apply(theta, g)                    # update!(opt, θ, g)
        if x < min_err:
            min_err, min_theta = x, theta.copy()        # min_θ = copy(θ): after the update
        if i == maxiters:                               # "Last iteration, revert to best."
            theta = min_theta
            if cb is not None:
                cb(theta, min_err)                      # cb(θ, x...) once more, on the reverted point; then break
            break
    return theta


def test_train_NDE_follows_galacticoptim_over_optimizers_and_epochs():
    prob = Quadratic()
    w0 = np.zeros(12, np.float32)
    # literal: for i in optimizers, epoch in 1:epochs: res = solve(prob_loss, opt, maxiters); weights .= res.minimizer
    lit_opts = [FluxADAM(0.3), FluxADAM(0.05)]
    w, keep = w0.astype(np.float64), []
    for opt in lit_opts:
        for _ in range(2):
            w = galactic_solve(prob.grad_loss, w, opt, 15, keep)
    res = train_NDE(prob, w0, [ADAM(0.3), ADAM(0.05)], epochs=2, maxiters=15)
    np.testing.assert_allclose(res.weights, w, rtol=2e-5, atol=1e-6)
    assert len(res.history) == 2 * 2 * 15
    # the reset matters: carrying the moments across solves gives a different trajectory
    res_c = train_NDE(prob, w0, [ADAM(0.3), ADAM(0.05)], epochs=2, maxiters=15, continue_state=True)
    assert np.abs(res_c.weights - w).max() > 1e-3
    # and so does where the best θ is copied: the argmin-of-loss iterate itself is a different point
    assert prob.grad_loss(res.weights)[0] != min(h["total"] for h in res.history)


def test_callback_stops_and_sees_reference_arguments():
    prob = Quadratic()
    seen = []

    def cb(theta, total, losses, scalings):
        seen.append((theta.copy(), total, losses, scalings))
        return len(seen) >= 4
    train_NDE(prob, np.zeros(12, np.float32), [ADAM(0.1)], epochs=1, maxiters=50, cb=cb)      # halted: no extra callback
    assert len(seen) == 4 and seen[0][1] == prob.grad_loss(np.zeros(12))[0] and seen[0][3] is prob.loss_scalings


def test_flux_train_keeps_adam_state_across_calls():
    """`Flux.train!(nde_loss, Flux.params(NN), repeated((), epochs), opt)` (training.jl:71): the parameter arrays persist, so one
    optimiser's moments carry over from call to call — two calls of 5 epochs equal one call of 10."""
    class P:
        q = Quadratic(8, 1)

        def nde_loss_and_grad(self, th):
            v, _, g = self.q.grad_loss(th)
            return v, g
    a, ha = train_neural_differential_equation(P(), np.zeros(8, np.float32), ADAM(0.05), 10)
    opt = ADAM(0.05)
    b, h1 = train_neural_differential_equation(P(), np.zeros(8, np.float32), opt, 5)
    b, h2 = train_neural_differential_equation(P(), b, opt, 5)
    np.testing.assert_allclose(b, a, rtol=1e-6)
    np.testing.assert_allclose(h1 + h2, ha, rtol=1e-6)
    # literal Flux loop
    lit, th = FluxADAM(0.05), np.zeros(8)
    for _ in range(10):
        th -= lit.apply(th, P.q.grad_loss(th)[2])
    np.testing.assert_allclose(a, th, rtol=2e-5, atol=1e-6)


def test_solve_issues_maxiters_plus_one_callbacks_the_last_on_the_reverted_best():
    """GalacticOptim's `__solve` calls `cb(min_θ, min_err...)` once more at `i == maxiters`: an un-halted solve ISSUES maxiters + 1
    callbacks, the last one on the reverted best θ with the best loss.  (What the reference's own callback does with that extra call — nothing:
    `if iter <= maxiters`, NDE_training.jl:344 — is the next test's subject)"""
    prob = Quadratic()
    lit_calls, calls = [], []
    w = galactic_solve(prob.grad_loss, np.zeros(12), FluxADAM(0.3), 15, cb=lambda th, x: lit_calls.append((th.copy(), x)) and False)

    def cb(theta, total, losses, scalings):
        calls.append((theta.copy(), total))
        return False
    res = train_NDE(prob, np.zeros(12, np.float32), [ADAM(0.3)], epochs=1, maxiters=15, cb=cb)
    assert len(lit_calls) == len(calls) == 16
    assert len(res.history) == 15                                     # the extra callback evaluates nothing new
    np.testing.assert_allclose(calls[-1][0], lit_calls[-1][0], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(calls[-1][0], res.weights, rtol=0, atol=0)
    assert calls[-1][1] == min(c[1] for c in calls[:-1]) and np.isclose(calls[-1][1], lit_calls[-1][1], rtol=1e-6)
    # two epochs: 2 x (maxiters + 1); a halted solve gets no extra call
    calls.clear()
    train_NDE(prob, np.zeros(12, np.float32), [ADAM(0.3)], epochs=2, maxiters=5, cb=cb)
    assert len(calls) == 12


def test_reference_logging_callback_writes_maxiters_records_per_solve(tmp_path):
    """the reference's `cb` is guarded by `if iter <= maxiters` (NDE_training.jl:343-368), so the extra callback GalacticOptim issues on
    the reverted best point neither prints nor writes: a reference log holds exactly `maxiters` records per (optimizer, epoch) solve.
    `reference_logging_callback` reproduces the closure; wired into `train_NDE` as the docstring says, the log has maxiters records per solve,
    their losses are the per-iteration ones (not the duplicated minimum), and `extract_NN` indexes them as in a reference log."""
    from colnde import checkpoint as ck
    from colnde.config import NDEConfig
    from colnde.wind_mixing import reference_logging_callback
    cfg = NDEConfig(layer_sizes=(96, 2, 31), activations=("mish", "identity"), save_times=(0.0, 1.0))
    n = cfg.n_params

    class Prob:                        # a quadratic in the full weight vector with the six-term loss bookkeeping of WindMixingNDE.grad_loss
        loss_scalings = np.array([1, 1, 1, 5e-3, 5e-3, 5e-3])

        def grad_loss(self, th):
            th = np.asarray(th, np.float64)
            tot = float(0.5 * np.sum((th - 0.3) ** 2))
            return tot, dict(zip(("u", "v", "T", "dudz", "dvdz", "dTdz"), [tot / 6] * 6)), (th - 0.3).astype(np.float32)

    path = str(tmp_path / "log.tree")
    maxiters, calls = 7, []
    w = np.zeros(n, np.float32)
    nets0 = [ck.network_record(w[k * (n // 3):(k + 1) * (n // 3)], cfg.layer_sizes, cfg.activations) for k in range(3)]
    opts = [ADAM(0.05), ADAM(0.02)]
    ck.write_metadata_NDE_training(path, ["f"], [1], [range(1, 3)], {"Pr": 1.0}, [opts], *nets0)
    for opt in opts:                                       # one cb per solve, as train_NDE's loop builds them (the stage is the same: one training stage)
        cb = reference_logging_callback(path, cfg, 1, opt, maxiters, log=lambda it, tot, losses: calls.append((it, tot)))
        res = train_NDE(Prob(), w, [opt], epochs=1, maxiters=maxiters, cb=cb)
        w = res.weights
    assert len(calls) == 2 * maxiters and [c[0] for c in calls[:maxiters]] == list(range(1, maxiters + 1))
    with ck.GroupFile(path) as f:
        keys = f.keys("training_data/loss/total/1")
        assert len(keys) == 2 * maxiters                    # maxiters per solve — not maxiters + 1
        logged = [float(f["training_data/loss/total/1/%d" % (i + 1)]) for i in range(2 * maxiters)]
    np.testing.assert_allclose(logged, [c[1] for c in calls], rtol=1e-6)
    assert len(set(np.round(logged[:maxiters], 12))) == maxiters      # strictly decreasing quadratic: no duplicated minimum record
