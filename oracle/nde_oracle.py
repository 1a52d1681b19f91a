"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  **parity unpinned.**

A NumPy restatement (float64 by default) of the reference's NDE column-model hot path: the RHS
variants, a fixed-step classical RK4 solve, the profile/gradient MSE losses and the *discrete*
adjoint (back-propagation through the RK4 steps) giving ∂loss/∂weights.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this
module; the product path (`climateparameterizations.jl_amd`) never does.

"parity unpinned": the reference (pure Julia; OrdinaryDiffEq 5.55.1 / DiffEqSensitivity 6.45.0 /
Flux 0.11.6 / Zygote 0.6.11 pinned in wind_mixing/Manifest.toml) cannot run in the build container
(no `julia`), and its own tests hold no golden vector, known-answer test or fixture for this path
(SURVEY §4, §8c).  What pins this oracle instead: analytic known-answer cases, a literal
dense-matrix torch-autograd restatement, finite differences, and SciPy adaptive solves (tests/).

Each function cites the reference lines it restates (paths relative to /root/reference).
Index convention here is 0-based: cells k = 0..Nz-1 (0 = deepest), faces f = 0..Nz (0 = bottom).
"""
from __future__ import annotations

import numpy as np

WIND_MIXING, FREE_CONVECTION, CONVECTIVE_ADJUSTMENT_NDE = 0, 1, 2


# ----------------------------------------------------------------------------------------------
# operators  (src/differentiation_operators.jl, wind_mixing/src/filtering_operators.jl)
# ----------------------------------------------------------------------------------------------
def Dc(N, delta):
    """`Dᶜ(N, Δ)`: N×(N+1), face→cell  (src/differentiation_operators.jl:6-14)."""
    D = np.zeros((N, N + 1))
    for k in range(N):
        D[k, k] = -1.0
        D[k, k + 1] = 1.0
    return D / delta


def Df(N, delta):
    """`Dᶠ(N, Δ)`: (N+1)×N, cell→face, first and last rows zero  (src/differentiation_operators.jl:21-29)."""
    D = np.zeros((N + 1, N))
    for k in range(1, N):
        D[k, k - 1] = -1.0
        D[k, k] = 1.0
    return D / delta


def smoothing_filter(N, filter_width=3):
    """`smoothing_filter(N, w)`: moving average with shortened end windows (filtering_operators.jl:1-14)."""
    assert N >= filter_width and filter_width % 2 == 1
    F = np.zeros((N, N))
    hw = (filter_width - 1) // 2
    for i in range(1, hw + 1):                       # Julia 1-based i
        F[i - 1, 0:hw + i] = 1.0 / (hw + i)
        F[N - i, N - (hw + i):N] = 1.0 / (hw + i)
    for i in range(hw + 1, N - hw + 1):
        F[i - 1, i - hw - 1:i + hw] = 1.0 / filter_width
    return F


def _face_grad(q, Nz):
    """(Dᶠ q) with Δ = 1/Nz, batched over the leading axis: [n, Nz] → [n, Nz+1]."""
    g = np.zeros(q.shape[:-1] + (Nz + 1,), dtype=q.dtype)
    g[..., 1:Nz] = (q[..., 1:] - q[..., :-1]) * Nz
    return g


def _face_grad_T(gbar, Nz):
    """transpose of `_face_grad`."""
    qbar = np.zeros(gbar.shape[:-1] + (Nz,), dtype=gbar.dtype)
    qbar[..., 1:] += gbar[..., 1:Nz] * Nz
    qbar[..., :-1] -= gbar[..., 1:Nz] * Nz
    return qbar


# ----------------------------------------------------------------------------------------------
# activations (NNlib 0.7: relu, mish, swish, leakyrelu, tanh)
# ----------------------------------------------------------------------------------------------
def _softplus(x):
    return np.where(x > 0, x + np.log1p(np.exp(-np.abs(x))), np.log1p(np.exp(-np.abs(x))))


def _sigmoid(x):
    e = np.exp(-np.abs(x))
    return np.where(x >= 0, 1.0 / (1.0 + e), e / (1.0 + e))


def act(name, z):
    if name == "identity":
        return z
    if name == "relu":
        return np.maximum(z, 0)
    if name == "mish":
        return z * np.tanh(_softplus(z))
    if name == "swish":
        return z * _sigmoid(z)
    if name == "tanh":
        return np.tanh(z)
    if name == "leakyrelu":
        return np.maximum(0.01 * z, z)
    raise ValueError(name)


def act_grad(name, z):
    if name == "identity":
        return np.ones_like(z)
    if name == "relu":
        return (z > 0).astype(z.dtype)
    if name == "mish":
        t = np.tanh(_softplus(z))
        return t + z * (1 - t * t) * _sigmoid(z)
    if name == "swish":
        s = _sigmoid(z)
        return s + z * s * (1 - s)
    if name == "tanh":
        t = np.tanh(z)
        return 1 - t * t
    if name == "leakyrelu":
        return np.where(z > 0, 1.0, 0.01).astype(z.dtype)
    raise ValueError(name)


# ----------------------------------------------------------------------------------------------
# parameter packing (Flux.destructure order; NDE_training.jl:11-21,37,58-64)
# ----------------------------------------------------------------------------------------------
def unpack(theta, layer_sizes, n_nets):
    nets, o = [], 0
    for _ in range(n_nets):
        layers = []
        for i in range(len(layer_sizes) - 1):
            n_in, n_out = layer_sizes[i], layer_sizes[i + 1]
            W = theta[o:o + n_in * n_out].reshape((n_out, n_in), order="F")
            o += n_in * n_out
            b = theta[o:o + n_out]
            o += n_out
            layers.append((W, b))
        nets.append(layers)
    assert o == len(theta), (o, len(theta))
    return nets


def pack_grads(grads):
    parts = []
    for layers in grads:
        for Wb, bb in layers:
            parts.append(Wb.reshape(-1, order="F"))
            parts.append(bb.reshape(-1))
    return np.concatenate(parts)


def mlp_forward(layers, acts, x):
    """`Chain(Dense…)(x)`: σ.(W*x .+ b) per layer; batched x [n, in].  Returns output and the tape."""
    a, tape = x, []
    for (W, b), name in zip(layers, acts):
        z = a @ W.T + b
        tape.append((a, z))
        a = act(name, z)
    return a, tape


def mlp_vjp(layers, acts, tape, ybar):
    """Pullback of `mlp_forward`: returns xbar and [(Wbar, bbar)] summed over the batch."""
    grads = [None] * len(layers)
    abar = ybar
    for li in range(len(layers) - 1, -1, -1):
        W, _ = layers[li]
        a_prev, z = tape[li]
        zbar = abar * act_grad(acts[li], z)
        grads[li] = (zbar.T @ a_prev, zbar.sum(axis=0))
        abar = zbar @ W
    return abar, grads


# ----------------------------------------------------------------------------------------------
# RHS variants
# ----------------------------------------------------------------------------------------------
class Model:
    """Bundles cfg-derived constants; `cfg` is any object with NDEConfig's attribute names."""

    def __init__(self, cfg, dtype=np.float64):
        self.cfg, self.dtype = cfg, dtype
        self.Nz = cfg.Nz
        self.acts = tuple(cfg.activations)
        self.n_nets = 3 if cfg.model == WIND_MIXING else 1
        mu, sg = cfg.mu, cfg.sigma
        self.mu_u, self.mu_v, self.mu_T, self.mu_uw, self.mu_vw, self.mu_wT = [dtype(m) for m in mu]
        self.s_u, self.s_v, self.s_T, self.s_uw, self.s_vw, self.s_wT = [dtype(s) for s in sg]
        if cfg.smooth_NN:
            self.F_int = smoothing_filter(cfg.Nz - 1, 3).astype(dtype)
        if cfg.smooth_Ri:
            self.F_face = smoothing_filter(cfg.Nz + 1, 3).astype(dtype)

    def unpack(self, theta):
        return unpack(np.asarray(theta, dtype=self.dtype), self.cfg.layer_sizes, self.n_nets)

    # -- wind mixing: NDE / predict_flux / predict_NDE  (NDE_training.jl:56-165; in-place twin
    #    training_postprocessing.jl:105-153 when cfg.inplace_variant) -----------------------------
    def wm_top_flux(self, bcs, t):
        """wT_top; diurnal: `scalings.wT(Qᵇ sin(2π/86400 · t τ)/(α g))` (NDE_training.jl:73, data_containers.jl:135)."""
        c = self.cfg
        if not c.diurnal:
            return bcs[:, 5]
        Q = bcs[:, 5]
        return (Q * np.sin(2 * np.pi / 86400.0 * (t * c.tau)) / (c.alpha * c.g) - self.mu_wT) / self.s_wT

    def wm_rhs(self, x, bcs, nets, t=0.0, want_vjp=False, sw=None, want_flux=False):
        c, Nz = self.cfg, self.Nz
        H, tau, f = c.H, c.tau, c.f
        u, v, T = x[:, :Nz], x[:, Nz:2 * Nz], x[:, 2 * Nz:]
        s_u, s_v, s_T, s_uw, s_vw, s_wT = self.s_u, self.s_v, self.s_T, self.s_uw, self.s_vw, self.s_wT
        n = x.shape[0]
        # (1) three MLPs on the full 3Nz vector (NDE_training.jl:94-96)
        outs, tapes = [], []
        for k in range(3):
            o, tp = mlp_forward(nets[k], self.acts, x)
            if c.smooth_NN:                               # :98-102
                o = o @ self.F_int.T
            outs.append(o)
            tapes.append(tp)
        s0 = (-self.mu_uw / s_uw, -self.mu_vw / s_vw, -self.mu_wT / s_wT)   # scalings.φ(0f0)
        bc_b = (bcs[:, 0], bcs[:, 2], bcs[:, 4])
        bc_t = (bcs[:, 1], bcs[:, 3], self.wm_top_flux(bcs, t))
        # (2) face vectors (:104-112)
        F = []
        for k in range(3):
            Fk = np.zeros((n, Nz + 1), dtype=x.dtype)
            Fk[:, 1:Nz] = outs[k]
            if not c.zero_weights:
                Fk[:, 0], Fk[:, Nz] = bc_b[k], bc_t[k]
            F.append(Fk)
        cs = (s_u / s_uw / H, s_v / s_vw / H, s_T / s_wT / H)
        save = {}
        if c.modified_pacanowski_philander:               # :114-139
            eps = 0.0 if c.inplace_variant else c.eps
            gu, gv, gT = _face_grad(u, Nz), _face_grad(v, Nz), _face_grad(T, Nz)
            with np.errstate(divide="ignore", invalid="ignore"):
                S2 = (s_u * (gu + eps)) ** 2 + (s_v * (gv + eps)) ** 2
                B = H * c.g * c.alpha * s_T
                Ri = B * (gT + eps) / S2                  # local_richardson :46-52
            Ri_s = Ri @ self.F_face.T if c.smooth_Ri else Ri      # :121-123
            y = (Ri_s - c.Ric) / c.dRi
            th = np.tanh(y)
            nu = c.nu0 + c.nu_minus * (1 - th) / 2        # tanh_step :54, :125
            nu_T = nu / c.Pr
            if c.inplace_variant and c.convective_adjustment:
                # training_postprocessing.jl:118-121 — tests ∂u∂z (sic), not ∂T∂z
                nu_T = np.where(gu > 0, nu / c.Pr, c.kappa)
            D = [cs[0] * nu * gu, cs[1] * nu * gv, cs[2] * nu_T * gT]
            for k in range(3):
                if c.zero_weights:                        # :129-132
                    top = bc_t[k]
                    if c.inplace_variant and c.diurnal and k == 2:
                        # training_postprocessing.jl:142-144 overwrites wT[end] without the -scaling(0)
                        F[k][:, 0] += bc_b[k] - s0[k]
                        F[k][:, Nz] += top
                    else:
                        F[k][:, 0] += bc_b[k] - s0[k]
                        F[k][:, Nz] += top - s0[k]
                F[k][:, 1:Nz] -= D[k][:, 1:Nz]
            save.update(gu=gu, gv=gv, gT=gT, S2=S2, Ri=Ri, th=th, nu=nu, B=B, eps=eps)
        elif c.convective_adjustment:                     # :140-143 (reference leaves κ unbound; constants.κ meant)
            gT = _face_grad(T, Nz)
            F[2] = F[2] - cs[2] * c.kappa * np.minimum(0.0, gT)
            save.update(gT=gT)
        # (3) tendencies (predict_NDE :160-162)
        A = (tau / H * s_uw / s_u * Nz, tau / H * s_vw / s_v * Nz, tau / H * s_wT / s_T * Nz)
        du = -A[0] * (F[0][:, 1:] - F[0][:, :-1]) + f * tau / s_u * (s_v * v + self.mu_v)
        dv = -A[1] * (F[1][:, 1:] - F[1][:, :-1]) - f * tau / s_v * (s_u * u + self.mu_u)
        dT = -A[2] * (F[2][:, 1:] - F[2][:, :-1])
        dx = np.concatenate([du, dv, dT], axis=1)
        if want_flux:                                     # what predict_flux returns (:139, :143, :145): the three face vectors
            return np.stack(F, axis=1)
        if not want_vjp:
            return dx

        def vjp(dbar):
            dub, dvb, dTb = dbar[:, :Nz], dbar[:, Nz:2 * Nz], dbar[:, 2 * Nz:]
            xb_u = -(f * tau / s_v) * s_u * dvb
            xb_v = (f * tau / s_u) * s_v * dub
            xb_T = np.zeros_like(dTb)
            Fbar = []
            for k, db in enumerate((dub, dvb, dTb)):
                Fb = np.zeros((n, Nz + 1), dtype=x.dtype)
                Fb[:, 1:] += -A[k] * db
                Fb[:, :-1] += A[k] * db
                Fbar.append(Fb)
            if c.modified_pacanowski_philander:
                gu, gv, gT, S2, Ri, th, nu = (save[q] for q in ("gu", "gv", "gT", "S2", "Ri", "th", "nu"))
                eps, B = save["eps"], save["B"]
                sl = slice(1, Nz)
                Db = [-Fbar[k] for k in range(3)]
                gub = np.zeros_like(gu); gvb = np.zeros_like(gu); gTb = np.zeros_like(gu)
                nub = np.zeros_like(gu)
                gub[:, sl] = Db[0][:, sl] * cs[0] * nu[:, sl]
                gvb[:, sl] = Db[1][:, sl] * cs[1] * nu[:, sl]
                gTb[:, sl] = Db[2][:, sl] * cs[2] * nu[:, sl] / c.Pr
                nub[:, sl] = (Db[0][:, sl] * cs[0] * gu[:, sl] + Db[1][:, sl] * cs[1] * gv[:, sl]
                              + Db[2][:, sl] * cs[2] * gT[:, sl] / c.Pr)
                Ribs = nub * (-c.nu_minus / (2 * c.dRi)) * (1 - th * th)
                Rib = Ribs @ self.F_face if c.smooth_Ri else Ribs
                gTb[:, sl] += Rib[:, sl] * B / S2[:, sl]
                gub[:, sl] += Rib[:, sl] * (-Ri[:, sl] / S2[:, sl]) * 2 * s_u ** 2 * (gu[:, sl] + eps)
                gvb[:, sl] += Rib[:, sl] * (-Ri[:, sl] / S2[:, sl]) * 2 * s_v ** 2 * (gv[:, sl] + eps)
                xb_u = xb_u + _face_grad_T(gub, Nz)
                xb_v = xb_v + _face_grad_T(gvb, Nz)
                xb_T = xb_T + _face_grad_T(gTb, Nz)
            elif c.convective_adjustment:
                gT = save["gT"]
                gTb = -Fbar[2] * cs[2] * c.kappa * ((gT < 0) if sw is None else sw)
                xb_T = xb_T + _face_grad_T(gTb, Nz)
            xbar = np.concatenate([xb_u, xb_v, xb_T], axis=1)
            grads = []
            for k in range(3):
                ob = Fbar[k][:, 1:Nz]
                if c.smooth_NN:
                    ob = ob @ self.F_int
                xb, g = mlp_vjp(nets[k], self.acts, tapes[k], ob)
                xbar = xbar + xb
                grads.append(g)
            return xbar, grads

        return dx, vjp

    # -- free convection: ∂T∂t (free_convection_nde.jl:29-38) and convective-adjustment NDE
    #    (convective_adjustment_nde.jl:33-48) ----------------------------------------------------
    def fc_rhs(self, x, bcs, nets, t=0.0, want_vjp=False, sw=None, want_flux=False):
        c, Nz = self.cfg, self.Nz
        C = (self.s_wT / self.s_T) * (c.tau / c.H)
        n = x.shape[0]
        o, tape = mlp_forward(nets[0], self.acts, x)
        w = np.zeros((n, Nz + 1), dtype=x.dtype)
        w[:, 0], w[:, Nz], w[:, 1:Nz] = bcs[:, 0], bcs[:, 1], o
        dT = -C * Nz * (w[:, 1:] - w[:, :-1])
        ca = c.model == CONVECTIVE_ADJUSTMENT_NDE
        if ca:
            gT = _face_grad(x, Nz)
            q = np.minimum(0.0, c.ca_K * gT)
            dT = dT + C * Nz * (q[:, 1:] - q[:, :-1])
        if want_flux:                                     # free_convection/src/solve.jl:32-46: wT_NN (- min(0, 10 dT/dz) for the CA-NDE), scaled
            return (w - q if ca else w)[:, None, :]
        if not want_vjp:
            return dT

        def vjp(dbar):
            wb = np.zeros((n, Nz + 1), dtype=x.dtype)
            wb[:, 1:] += -C * Nz * dbar
            wb[:, :-1] += C * Nz * dbar
            xbar, g = mlp_vjp(nets[0], self.acts, tape, wb[:, 1:Nz])
            if ca:
                qb = -wb
                gTb = qb * c.ca_K * ((gT < 0) if sw is None else sw)
                xbar = xbar + _face_grad_T(gTb, Nz)
            return xbar, [g]

        return dT, vjp

    def rhs(self, x, bcs, nets, t=0.0, want_vjp=False, sw=None):
        """`sw`: switch pattern (dT/dz < 0 per face) to use in the PULLBACK of min(0, K dT/dz) instead of this state's own."""
        if self.cfg.model == WIND_MIXING:
            return self.wm_rhs(x, bcs, nets, t, want_vjp, sw)
        return self.fc_rhs(x, bcs, nets, t, want_vjp, sw)

    def switch_pattern(self, x):
        """The faces where the convective-adjustment switch is on at state x (None for the models without one)."""
        c, Nz = self.cfg, self.Nz
        if c.model == CONVECTIVE_ADJUSTMENT_NDE:
            return _face_grad(x, Nz) < 0
        if c.model == WIND_MIXING and c.convective_adjustment and not c.modified_pacanowski_philander:
            return _face_grad(x[:, 2 * Nz:], Nz) < 0
        return None


def predict_flux(cfg, x, bcs, theta, t=0.0, dtype=np.float64):
    """`predict_flux` (wind_mixing/src/NDE_training.jl:83-147) for every column: [n][n_nets][Nz + 1] face fluxes in scaled units; T-only models: the
    flux the dataset-level `solve_nde` re-evaluates per saved step (free_convection/src/solve.jl:32-46)."""
    m = Model(cfg, dtype)
    x, bcs = np.asarray(x, dtype), np.asarray(bcs, dtype)
    nets = m.unpack(theta)
    if cfg.model == WIND_MIXING:
        return m.wm_rhs(x, bcs, nets, t, want_flux=True)
    return m.fc_rhs(x, bcs, nets, t, want_flux=True)


def loss_per_tstep(cfg, sol, truth):
    """`loss_per_tstep(a, b)` (wind_mixing/src/loss.jl:44-46) on the six profile matrices of every simulation (training_postprocessing.jl:311-316):
    [n][6][n_save]; gradient terms over the Nz + 1 faces of `Dᶠ`, zero boundary rows included (loss.jl:9).  T-only models fill terms 2 and 5."""
    sol, truth = np.asarray(sol, np.float64), np.asarray(truth, np.float64)
    n, n_save, Nz = sol.shape[0], sol.shape[1], cfg.Nz
    nv = sol.shape[2] // Nz
    out = np.zeros((n, 6, n_save))
    for v in range(nv):
        d = sol[:, :, v * Nz:(v + 1) * Nz] - truth[:, :, v * Nz:(v + 1) * Nz]
        g = np.zeros(d.shape[:2] + (Nz + 1,))
        g[:, :, 1:Nz] = (d[:, :, 1:] - d[:, :, :-1]) * Nz
        term = v if nv == 3 else 2
        out[:, term] = (d ** 2).mean(axis=2)
        out[:, 3 + term] = (g ** 2).mean(axis=2)
    return out


def error_estimate(cfg, x0, bcs, theta, dtype=np.float64, reltol=None):
    """Richardson estimate of the error of the solve at cfg.substeps from one at twice as many (what colnde_error_estimate returns):
    e = (u_S - u_2S) 2^p / (2^p - 1); per column and save point rms_i(e_i / (abstol / reltol + |u_2S,i|)), then the maximum — the adaptive
    integrator's accept test rms(err / (abstol + reltol |u|)) <= 1 with OrdinaryDiffEq's default abstol = 1e-6, divided through by reltol
    (`reltol` = the tolerance in force: cfg.reltol, 0 = the reference's 1e-3 of NDE_training.jl:291; 1e-4 at free_convection/src/solve.jl:4).
    p = the order the right-hand side lets the stepper reach: 4 (RK4), 2 (RKC2), and 1 for the switching closures (ConvectiveAdjustmentNDE, the
    wind-mixing convective-adjustment branch), which converge at about first order through their kinks."""
    from dataclasses import replace
    a = solve(cfg, x0, bcs, theta, dtype=dtype)
    b = solve(replace(cfg, substeps=2 * cfg.substeps), x0, bcs, theta, dtype=dtype)
    switching = cfg.model == CONVECTIVE_ADJUSTMENT_NDE or (cfg.model == WIND_MIXING and cfg.convective_adjustment)
    p = 1 if switching else (2 if cfg.stepper == "rkc2" else 4)
    if reltol is None:
        reltol = getattr(cfg, "reltol", 0.0) or 1e-3
    q = (a - b) / (1e-6 / reltol + np.abs(b))
    return float(np.max(np.sqrt(np.mean(q * q, axis=-1)))) * 2 ** p / (2 ** p - 1)      # rms over the state's components, max over columns and save points


def rhs(cfg, x, bcs, theta, t=0.0, dtype=np.float64):
    """One RHS evaluation — the `NDE(x,p,t)` / `NDE!(dx,x,p,t)` / `∂T∂t(T,p,t)` closures."""
    m = Model(cfg, dtype)
    return m.rhs(np.asarray(x, dtype), np.asarray(bcs, dtype), m.unpack(theta), t)


# ----------------------------------------------------------------------------------------------
# fixed-step RK4 solve with `saveat` (stand-in for solve(prob, ROCK4(); saveat=t_train): NDE_training.jl:291)
# ----------------------------------------------------------------------------------------------
def _step_times(cfg):
    ts = np.asarray(cfg.save_times, dtype=np.float64)
    S = cfg.substeps
    out = []
    for i in range(len(ts) - 1):
        dt = (ts[i + 1] - ts[i]) / S
        for s in range(S):
            out.append((ts[i] + s * dt, dt, i + 1 if s == S - 1 else -1))
    return out


# ----------------------------------------------------------------------------------------------
# stabilised explicit stepper: second-order Runge-Kutta-Chebyshev (RKC2)
# The reference integrates its stiff variants with ROCK4 (wind_mixing/train_NDE.jl:143, free_convection/test_free_convection_nde.jl:32-35
# — OrdinaryDiffEq 5.55.1, pinned in the Manifests, absent from /root/reference): an adaptive stabilised Chebyshev method whose stage
# count follows the spectral radius.  A fixed-step product needs a fixed-stage member of the same family with a discrete adjoint:
# RKC2 of van der Houwen & Sommeijer in Verwer's form (Sommeijer, Shampine, Verwer, "RKC: an explicit solver for parabolic PDEs",
# J. Comput. Appl. Math. 88 (1998), eqs. 2.1-2.8), damping eps = 2/13.  Real stability interval beta(s) ~ 0.653 s^2.
# ----------------------------------------------------------------------------------------------
RKC_EPS = 2.0 / 13.0


def _cheb(s, w0):
    """T_j(w0), T_j'(w0), T_j''(w0) for j = 0..s by the three-term recurrences."""
    T, dT, d2T = np.zeros(s + 1), np.zeros(s + 1), np.zeros(s + 1)
    T[0], T[1], dT[1] = 1.0, w0, 1.0
    for j in range(2, s + 1):
        T[j] = 2 * w0 * T[j - 1] - T[j - 2]
        dT[j] = 2 * T[j - 1] + 2 * w0 * dT[j - 1] - dT[j - 2]
        d2T[j] = 4 * dT[j - 1] + 2 * w0 * d2T[j - 1] - d2T[j - 2]
    return T, dT, d2T


def rkc_coefficients(s):
    """(mu[1..s], nu[1..s], mut[1..s], gat[1..s], c[0..s], beta) of the s-stage RKC2 (index 0 unused for the first four):
         Y_0 = y,  Y_1 = Y_0 + mut_1 h F_0,
         Y_j = (1 - mu_j - nu_j) Y_0 + mu_j Y_{j-1} + nu_j Y_{j-2} + mut_j h F_{j-1} + gat_j h F_0   (j = 2..s),   y+ = Y_s,
       with F_j = f(t + c_j h, Y_j); beta = real stability boundary."""
    assert s >= 2
    w0 = 1.0 + RKC_EPS / s ** 2
    T, dT, d2T = _cheb(s, w0)
    w1 = dT[s] / d2T[s]
    b = np.zeros(s + 1)
    for j in range(2, s + 1):
        b[j] = d2T[j] / dT[j] ** 2
    b[0] = b[1] = b[2]
    a = 1.0 - b * T
    mu, nu, mut, gat = (np.zeros(s + 1) for _ in range(4))
    mut[1] = b[1] * w1
    for j in range(2, s + 1):
        mu[j] = 2 * b[j] * w0 / b[j - 1]
        nu[j] = -b[j] / b[j - 2]
        mut[j] = 2 * b[j] * w1 / b[j - 1]
        gat[j] = -a[j - 1] * mut[j]
    c = np.zeros(s + 1)
    c[1] = mut[1]
    for j in range(2, s + 1):
        c[j] = w1 * d2T[j] / dT[j]
    beta = (w0 + 1.0) * d2T[s] / dT[s]
    return mu, nu, mut, gat, c, beta


def stiff_lambda(cfg):
    """-lambda of the stiffest diffusive mode the configuration can switch on (4 D Nz^2; csrc/api.hip `stiff_lambda`)."""
    D = 0.0
    if cfg.model == WIND_MIXING:
        k = cfg.tau / cfg.H ** 2
        if cfg.modified_pacanowski_philander:
            D = k * (cfg.nu0 + cfg.nu_minus) * max(1.0, 1.0 / cfg.Pr)
            if cfg.inplace_variant and cfg.convective_adjustment:
                D = max(D, k * cfg.kappa)
        elif cfg.convective_adjustment:
            D = k * cfg.kappa
    elif cfg.model == CONVECTIVE_ADJUSTMENT_NDE:
        D = cfg.sigma[5] / cfg.sigma[2] * cfg.tau / cfg.H * cfg.ca_K
    return 4.0 * D * cfg.Nz ** 2


RKC_SAFETY = 0.9      # use 90 % of the real stability interval (the switching diffusivities are not a constant-coefficient problem)


def rkc_stages(cfg):
    """Stage count of the RKC2 step: cfg.rkc_stages if given, else the least s >= 2 with 0.9 beta(s) >= lambda h."""
    if getattr(cfg, "rkc_stages", 0):
        return int(cfg.rkc_stages)
    ts = np.asarray(cfg.save_times, np.float64)
    h = float(np.max(np.diff(ts))) / cfg.substeps
    z = stiff_lambda(cfg) * h
    s = 2
    while RKC_SAFETY * rkc_coefficients(s)[5] < z:
        s += 1
    return s


def _rkc_step(m, nets, bcs, x, t, h, co, s, want_tape=False):
    """One RKC2 step in increment form, d_j = Y_j - Y_0 (1 - mu_j - nu_j + mu_j + nu_j = 1):
         d_1 = mu~_1 h F_0,   d_j = mu_j d_{j-1} + nu_j d_{j-2} + mu~_j h F_{j-1} + gamma~_j h F_0,   Y_j = Y_0 + d_j
    — algebraically the recurrence of `rkc_coefficients`, but the differences 2 Y_{j-1} - Y_{j-2} are taken between increments
    instead of O(1) states, which is what keeps a float32 run (the product's arithmetic) accurate."""
    mu, nu, mut, gat, c, _ = co
    dt_ = m.dtype                      # every coefficient in the working precision (a float32 run is float32 throughout)
    mu, nu, mut, gat = (a.astype(dt_) for a in (mu, nu, mut, gat))
    F0 = m.rhs(x, bcs, nets, t)
    d = [np.zeros_like(x), mut[1] * h * F0]
    Y = [x, x + d[1]]
    for j in range(2, s + 1):
        Fp = m.rhs(Y[j - 1], bcs, nets, t + c[j - 1] * h)
        d.append(mu[j] * d[j - 1] + nu[j] * d[j - 2] + mut[j] * h * Fp + gat[j] * h * F0)
        Y.append(x + d[j])
    return (Y[s], Y[:s]) if want_tape else Y[s]


def solve(cfg, x0, bcs, theta, dtype=np.float64, return_tape=False):
    """Returns sol [n_col, n_save, n_state]; save point 0 is x0 (as `saveat` includes t_train[1]).
    cfg.stepper = "rkc2": each of the `substeps` steps per save interval is one s-stage RKC2 step (tape: its s stage states)."""
    m = Model(cfg, dtype)
    if getattr(cfg, "stepper", "rk4") == "rkc2":
        nets = m.unpack(theta)
        x = np.array(x0, dtype=dtype)
        bcs = np.asarray(bcs, dtype=dtype)
        s = rkc_stages(cfg)
        co = rkc_coefficients(s)
        sol = np.zeros((x.shape[0], len(cfg.save_times), x.shape[1]), dtype=dtype)
        sol[:, 0] = x
        tape = []
        for (t, dt, save_idx) in _step_times(cfg):
            x, Ys = _rkc_step(m, nets, bcs, x, t, dtype(dt), co, s, True)
            if return_tape:
                tape.append(Ys)
            if save_idx >= 0:
                sol[:, save_idx] = x
        return (sol, tape) if return_tape else sol
    nets = m.unpack(theta)
    x = np.array(x0, dtype=dtype)
    bcs = np.asarray(bcs, dtype=dtype)
    sol = np.zeros((x.shape[0], len(cfg.save_times), x.shape[1]), dtype=dtype)
    sol[:, 0] = x
    tape = []
    for (t, dt, save_idx) in _step_times(cfg):
        dt = dtype(dt)
        if return_tape:
            tape.append(x.copy())
        k1 = m.rhs(x, bcs, nets, t)
        k2 = m.rhs(x + dt / 2 * k1, bcs, nets, t + dt / 2)
        k3 = m.rhs(x + dt / 2 * k2, bcs, nets, t + dt / 2)
        k4 = m.rhs(x + dt * k3, bcs, nets, t + dt)
        x = x + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
        if save_idx >= 0:
            sol[:, save_idx] = x
    if return_tape:
        return sol, tape
    return sol


# ----------------------------------------------------------------------------------------------
# losses (wind_mixing/src/loss.jl:1-42, NDE_training.jl:290-323; free_convection/src/training.jl:55-62)
# ----------------------------------------------------------------------------------------------
def loss_terms(cfg, sol, truth):
    """Unscaled terms.  Wind mixing: (u, v, T, ∂u∂z, ∂v∂z, ∂T∂z) each `mean over sims of Flux.mse`;
    gradient terms use Dᶠ per snapshot, i.e. Nz+1 rows *including the two all-zero ones* (loss.jl:9).
    Free convection: a single Flux.mse over the concatenated trajectories (training.jl:58-60) in slot 2."""
    sol, truth = np.asarray(sol, np.float64), np.asarray(truth, np.float64)
    Nz = cfg.Nz
    terms = np.zeros(6)
    if cfg.model != WIND_MIXING:
        terms[2] = np.mean((sol - truth) ** 2)
        return terms
    d = truth - sol
    for k in range(3):
        dk = d[:, :, k * Nz:(k + 1) * Nz]
        terms[k] = np.mean(dk ** 2)
        gk = _face_grad(dk, Nz)
        terms[3 + k] = np.mean(gk ** 2)          # mean over (Nz+1)·Nt per sim, then over sims
    return terms


def default_loss_scalings(cfg, gradient_scaling=5e-3):
    """`training_fractions === nothing` branch (NDE_training.jl:257-258); loss_NDE zeroes the gradient terms (:298)."""
    if cfg.model != WIND_MIXING:
        return np.array([0, 0, 1.0, 0, 0, 0])
    g = gradient_scaling if cfg.train_gradient else 0.0
    return np.array([1.0, 1.0, 1.0, g, g, g])


def calculate_loss_scalings(losses, fractions, train_gradient):
    """wind_mixing/src/loss.jl:11-31; `losses` = (u,v,T,∂u∂z,∂v∂z,∂T∂z), fractions = dict(T, dTdz, profile)."""
    Lu, Lv, LT, Lgu, Lgv, LgT = losses
    vel = (1 - fractions["T"]) / fractions["T"] * LT / (Lu + Lv)
    profile_loss = vel * (Lu + Lv) + LT
    if train_gradient:
        velg = (1 - fractions["dTdz"]) / fractions["dTdz"] * LgT / (Lgu + Lgv)
        gradient_loss = velg * (Lgu + Lgv) + LgT
        tot = (1 - fractions["profile"]) / fractions["profile"] * profile_loss / gradient_loss
    else:
        velg = gradient_loss = tot = 0.0
    return np.array([vel, vel, 1.0, tot * velg, tot * velg, tot])


def loss(cfg, sol, truth, scalings):
    """(total, scaled terms) = `sum(apply_loss_scalings(losses, loss_scalings))` (NDE_training.jl:319-322)."""
    t = loss_terms(cfg, sol, truth) * np.asarray(scalings, np.float64)
    return t.sum(), t


def _loss_injection(cfg, sol_n, truth_n, scalings, n_col_total, n_save):
    """∂(total loss)/∂sol[:, n, :] for one save point; n_col_total = global column count."""
    Nz = cfg.Nz
    d = sol_n - truth_n
    if cfg.model != WIND_MIXING:
        return scalings[2] * 2.0 * d / (n_col_total * n_save * Nz)
    out = np.zeros_like(d)
    for k in range(3):
        dk = d[:, k * Nz:(k + 1) * Nz]
        out[:, k * Nz:(k + 1) * Nz] = scalings[k] * 2.0 * dk / (n_col_total * n_save * Nz)
        if scalings[3 + k] != 0:
            gk = _face_grad(dk, Nz)
            out[:, k * Nz:(k + 1) * Nz] += scalings[3 + k] * 2.0 * _face_grad_T(gk, Nz) / (n_col_total * n_save * (Nz + 1))
    return out


def loss_and_grad(cfg, x0, bcs, theta, truth, scalings, dtype=np.float64, n_col_total=None):
    """Discrete adjoint of `solve` + `loss`: returns (total, scaled terms[6], grad[n_params], sol).

    Stand-in for Zygote through `solve(...; sensealg=InterpolatingAdjoint(autojacvec=ZygoteVJP()))`
    (NDE_training.jl:303-333); the reference's adjoint is continuous, so agreement is to solver
    tolerance only (SURVEY §7 "Time stepper mismatch")."""
    m = Model(cfg, dtype)
    nets = m.unpack(theta)
    bcs = np.asarray(bcs, dtype=dtype)
    truth = np.asarray(truth, dtype=dtype)
    scalings = np.asarray(scalings, dtype=np.float64)
    sol, tape = solve(cfg, x0, bcs, theta, dtype, return_tape=True)
    n_col = sol.shape[0]
    if n_col_total is None:
        n_col_total = n_col
    n_save = sol.shape[1]
    total, terms = loss(cfg, sol, truth, scalings)
    if n_col_total != n_col:
        total, terms = total * n_col / n_col_total, terms * n_col / n_col_total
    lam = np.zeros_like(sol[:, 0])
    gacc = None
    steps = _step_times(cfg)
    rkc = getattr(cfg, "stepper", "rk4") == "rkc2"
    if rkc:
        s_ = rkc_stages(cfg)
        mu, nu, mut, gat, cst, _ = rkc_coefficients(s_)
        kap = (1.0 - mu - nu).astype(dtype)
        mu, nu, mut, gat = (a.astype(dtype) for a in (mu, nu, mut, gat))
    for si in range(len(steps) - 1, -1, -1):
        t, dt, save_idx = steps[si]
        dt = dtype(dt)
        if save_idx >= 0:
            lam = lam + _loss_injection(cfg, sol[:, save_idx], truth[:, save_idx], scalings, n_col_total, n_save)
        if rkc:
            # discrete adjoint of one RKC2 step: cotangents Yb[j] of the stage states, F0b of F_0; one VJP per stage state Y_0..Y_{s-1}.
            # The convective-adjustment switch is pulled back with ONE pattern per step, that of Y_{s-1}: with per-stage patterns
            # the stage polynomials no longer cancel and the exact adjoint of the recurrence is unbounded (|grad| 1e12..1e57 on the
            # 64-level CA-NDE; tests/test_oracle.py::test_rkc2_switch_pullback).  Smooth closures are unaffected (sw is None).
            Ys = tape[si]
            sw = m.switch_pattern(Ys[s_ - 1]) if not getattr(cfg, "rkc_exact_switch_pullback", False) else None
            Yb = [np.zeros_like(lam) for _ in range(s_ + 1)]
            Yb[s_] = lam
            F0b = np.zeros_like(lam)
            gs = None
            for j in range(s_, 1, -1):
                Yb[0] = Yb[0] + kap[j] * Yb[j]
                Yb[j - 1] = Yb[j - 1] + mu[j] * Yb[j]
                Yb[j - 2] = Yb[j - 2] + nu[j] * Yb[j]
                F0b = F0b + gat[j] * dt * Yb[j]
                _, v = m.rhs(Ys[j - 1], bcs, nets, t + cst[j - 1] * dt, True, sw)
                xb, g = v(mut[j] * dt * Yb[j])
                Yb[j - 1] = Yb[j - 1] + xb
                gs = pack_grads(g) if gs is None else gs + pack_grads(g)
            Yb[0] = Yb[0] + Yb[1]
            F0b = F0b + mut[1] * dt * Yb[1]
            _, v = m.rhs(Ys[0], bcs, nets, t, True, sw)
            xb, g = v(F0b)
            lam = Yb[0] + xb
            gs = pack_grads(g) if gs is None else gs + pack_grads(g)
            gacc = gs if gacc is None else gacc + gs
            continue
        x = tape[si]
        k1, v1 = m.rhs(x, bcs, nets, t, True)
        X2 = x + dt / 2 * k1
        k2, v2 = m.rhs(X2, bcs, nets, t + dt / 2, True)
        X3 = x + dt / 2 * k2
        k3, v3 = m.rhs(X3, bcs, nets, t + dt / 2, True)
        X4 = x + dt * k3
        k4, v4 = m.rhs(X4, bcs, nets, t + dt, True)
        k4b = dt / 6 * lam
        x4b, g4 = v4(k4b)
        k3b = dt / 3 * lam + dt * x4b
        x3b, g3 = v3(k3b)
        k2b = dt / 3 * lam + dt / 2 * x3b
        x2b, g2 = v2(k2b)
        k1b = dt / 6 * lam + dt / 2 * x2b
        x1b, g1 = v1(k1b)
        lam = lam + x1b + x2b + x3b + x4b
        gs = pack_grads(g1) + pack_grads(g2) + pack_grads(g3) + pack_grads(g4)
        gacc = gs if gacc is None else gacc + gs
    return total, terms, gacc, sol


# ----------------------------------------------------------------------------------------------
# embedded inference (free_convection/double_gyre_nn.jl:149-168, :135,:140-147)
# ----------------------------------------------------------------------------------------------
def infer_forcing(cfg, T, top_flux, theta, Lz, dtype=np.float64):
    """T [n_col, Nz] in model units → forcing = −∂z wT on cell centres (Δz = Lz/Nz).
    `T̃ = 19.65 + T/20` (:156); `wT_int = unscale_wT(NN(scale_T(T̃)))` (:158-159); `wT = [0; wT_int; top]` (:160)."""
    m = Model(cfg, dtype)
    nets = m.unpack(theta)
    T = np.asarray(T, dtype)
    Tt = 19.65 + T / 20.0
    o, _ = mlp_forward(nets[0], m.acts, (Tt - m.mu_T) / m.s_T)
    wT = np.zeros((T.shape[0], cfg.Nz + 1), dtype)
    wT[:, 1:cfg.Nz] = m.s_wT * o + m.mu_wT
    wT[:, cfg.Nz] = np.asarray(top_flux, dtype)
    dz = Lz / cfg.Nz
    return -(wT[:, 1:] - wT[:, :-1]) / dz


# ----------------------------------------------------------------------------------------------
# the steps either side of the hot path
# ----------------------------------------------------------------------------------------------
def convective_adjustment(T, dt, dz, K, halo_bottom=None, halo_top=None, dtype=np.float64):
    """`convective_adjustment!(model, Δt, K)` — free_convection/double_gyre_nn.jl:27-62, 1-D twin
    free_convection/src/oceananigans_nn.jl:13-40 — restated literally: the centred `∂z(T)` (:34, mean of the two face
    gradients = (T[k+1] − T[k−1])/(2Δz) with halo cells at the ends), `κ[k] = ∂T∂z[k] < 0 ? K : 0` (:37-40),
    `ld`, `ud`, `d` (:46-53) assembled into a DENSE matrix and solved with LAPACK (`Tridiagonal \\`, :55-57).
    T [n_col, Nz], k = 0 deepest.  Halo cells default to the nearest interior value (Oceananigans fills the halos of
    flux-bounded fields with zero normal gradient; Value/Gradient boundary conditions give other halo values, which the
    caller passes in)."""
    T = np.asarray(T, dtype)
    n, Nz = T.shape
    hb = T[:, 0] if halo_bottom is None else np.asarray(halo_bottom, dtype)
    ht = T[:, -1] if halo_top is None else np.asarray(halo_top, dtype)
    Text = np.concatenate([hb[:, None], T, ht[:, None]], axis=1)
    dTdz = (Text[:, 2:] - Text[:, :-2]) / (2.0 * dz)
    kappa = np.where(dTdz < 0, dtype(K), dtype(0))
    c = dtype(dt) / dtype(dz) ** 2
    out = np.empty_like(T)
    for i in range(n):
        k = kappa[i]
        L = np.zeros((Nz, Nz), dtype)
        for r in range(1, Nz):
            L[r, r - 1] = -c * k[r]                 # ld[k] = -Δt/Δz² κ[k], k in 2:Nz
        for r in range(Nz - 1):
            L[r, r + 1] = -c * k[r + 1]             # ud[k] = -Δt/Δz² κ[k+1], k in 1:Nz-1
            L[r, r] = 1 + c * (k[r] + k[r + 1])
        L[Nz - 1, Nz - 1] = 1 + c * k[Nz - 1]
        out[i] = np.linalg.solve(L, T[i])
    return out


def modified_pacanowski_philander_step(u, v, T, dt, dz, nu0, nu_minus, dRi, Ric, Pr, alpha, g, convective_adjustment=False,
                                       halo_bottom=None, dtype=np.float64):
    """`modified_pacanowski_philander!(model, constants, Δt, p, convective_adjustment)` — wind_mixing/src/NDE_oceananigans.jl:61-101
    with `modified_pacanowski_philander_diffusivity` (:17-58), restated literally for a batch of columns [n_col, Nz] (k = 0 deepest):
    `Ri` on the Nz+1 faces = ∂z b / ((∂z u)² + (∂z v)²) with b = gαT (Oceanostics 0.3.2 `richardson_number_ccf!` on the
    (Center, Center, Face) location of a 1x1 periodic column — third-party, pinned in wind_mixing/Manifest.toml:1279, absent from
    /root/reference; its published kernel is that ratio of centred face differences), `ν[i] = ν₀ + ν₋ tanh_step((Ri[i] − Riᶜ)/ΔRi)`
    for i in 2:Nz and 0 at both ends (:45-47), `ν_T = Ri > 0 ? ν/Pr : 1` under convective adjustment, `ν/Pr` otherwise (:49-55),
    the three diagonals of :69-83 assembled DENSELY and solved with LAPACK (`Tridiagonal \\`, :88-90), then `T′[1] = T_bottom` (:94).
    halo_bottom: (u, v, T) halo cells below the deepest cell [3, n_col] (they carry the ocean model's boundary conditions; only
    Ri[1] — the ν_T switch of the bottom face — reads them) or None for the zero-gradient fill (Ri[1] = 0/0 = NaN, `NaN > 0` false)."""
    u, v, T = (np.asarray(a, dtype) for a in (u, v, T))
    n, Nz = T.shape
    if halo_bottom is None:
        hb = np.stack([u[:, 0], v[:, 0], T[:, 0]])
    else:
        hb = np.asarray(halo_bottom, dtype)
    # top halo: zero-gradient; Ri[Nz+1] enters ν_T[Nz+1] only, which no diagonal reads
    ext = lambda a, h: np.concatenate([h[:, None], a, a[:, -1:]], axis=1)
    ue, ve, Te = ext(u, hb[0]), ext(v, hb[1]), ext(T, hb[2])
    with np.errstate(divide="ignore", invalid="ignore"):
        dudz, dvdz = np.diff(ue, axis=1) / dz, np.diff(ve, axis=1) / dz
        dbdz = dtype(g) * dtype(alpha) * np.diff(Te, axis=1) / dz
        Ri = dbdz / (dudz ** 2 + dvdz ** 2)                                     # [n, Nz+1]
        nu = np.zeros((n, Nz + 1), dtype)
        nu[:, 1:Nz] = nu0 + nu_minus * (1 - np.tanh((Ri[:, 1:Nz] - Ric) / dRi)) / 2
        nu_T = np.where(Ri > 0, nu / Pr, dtype(1)) if convective_adjustment else nu / Pr
    c = dtype(dt) / dtype(dz) ** 2
    uo, vo, To = np.empty_like(u), np.empty_like(v), np.empty_like(T)

    def tri(nuf):
        L = np.zeros((Nz, Nz), dtype)
        for r in range(1, Nz):
            L[r, r - 1] = -c * nuf[r]                       # lower[i] = -Δt/Δz² ν[i], i in 2:Nz
        for r in range(Nz - 1):
            L[r, r + 1] = -c * nuf[r + 1]                   # upper[i] = -Δt/Δz² ν[i+1], i in 1:Nz-1
            L[r, r] = 1 + c * (nuf[r] + nuf[r + 1])
        L[Nz - 1, Nz - 1] = 1 + c * nuf[Nz - 1]
        return L
    for i in range(n):
        Lv, LT = tri(nu[i]), tri(nu_T[i])
        uo[i] = np.linalg.solve(Lv, u[i])
        vo[i] = np.linalg.solve(Lv, v[i])
        To[i] = np.linalg.solve(LT, T[i])
        To[i, 0] = T[i, 0]
    return uo, vo, To


def adam_step(theta, grad, m, v, eta, beta, eps, beta_t):
    """Flux.Optimise.ADAM `apply!` + `update!` (Flux 0.11.6, src/optimise/optimisers.jl — third-party, pinned in
    wind_mixing/Manifest.toml; call sites NDE_training.jl:340-372, training.jl:71).  Returns (theta, m, v, beta_t) after
    one step; beta_t = running powers (β₁ᵗ, β₂ᵗ), (β₁, β₂) on the first step."""
    theta, grad, m, v = (np.asarray(a, np.float64) for a in (theta, grad, m, v))
    m = beta[0] * m + (1 - beta[0]) * grad
    v = beta[1] * v + (1 - beta[1]) * grad * grad
    delta = m / (1 - beta_t[0]) / (np.sqrt(v / (1 - beta_t[1])) + eps) * eta
    return theta - delta, m, v, (beta_t[0] * beta[0], beta_t[1] * beta[1])


def coarse_grain_center(phi, n):
    """`coarse_grain(Φ, n, Center)` — src/DataWrangling/coarse_graining.jl:8-16 (block means, Δ = N/n)."""
    phi = np.asarray(phi, np.float64)
    d = phi.shape[-1] // n
    assert d * n == phi.shape[-1]
    return phi.reshape(phi.shape[:-1] + (n, d)).mean(axis=-1)


def coarse_grain_linear_interpolation_face(phi, n):
    """`coarse_grain_linear_interpolation(Φ, n, Face)` — src/DataWrangling/coarse_graining.jl:47-62, the form
    wind_mixing/src/data_containers.jl:357 uses: end points kept, interior point i (1-based) at p = 1 + (i-1)(N-1)/(n-1)."""
    phi = np.asarray(phi, np.float64)
    N = phi.shape[-1]
    out = np.empty(phi.shape[:-1] + (n,), np.float64)
    out[..., 0], out[..., -1] = phi[..., 0], phi[..., -1]
    gap = (N - 1) / (n - 1)
    for i in range(2, n):                          # 1-based interior indices, as in the reference loop
        p = 1 + (i - 1) * gap
        f = int(np.floor(p))
        out[..., i - 1] = (f + 1 - p) * phi[..., f - 1] + (p - f) * phi[..., min(f, N - 1)]
    return out


def zero_mean_unit_variance(data):
    """`ZeroMeanUnitVarianceScaling(data)` — src/DataWrangling/feature_scaling.jl:17-23: (μ, σ = std with n-1), scaled."""
    data = np.asarray(data, np.float64)
    mu, sigma = data.mean(), data.std(ddof=1)
    return (data - mu) / sigma, mu, sigma


# ----------------------------------------------------------------------------------------------
# flux-MLP pre-training: predict_uw / predict_vw / predict_wT and train_NN
# (wind_mixing/src/NN_training.jl:25-169, 207-249; free convection: train_free_convection_nde.jl:124-128, 186-216)
# ----------------------------------------------------------------------------------------------
def predict_single_flux(cfg, k, x, bcs, layers, want_tape=False, dtype=np.float64):
    """Face vector of ONE flux net (k = 0 uw, 1 vw, 2 wT) for profiles x [n, n_state], as predict_uw/vw/wT build it: NN output on the
    interior faces, boundary faces 0 (zero_weights) or the BCs, minus the Richardson-number diffusive flux (MPP; with zero_weights
    the end faces carry -(BC - scaling(0))) or, for wT, the convective-adjustment flux.  T-only models: [bottom; NN(T); top]."""
    m = Model(cfg, dtype)
    Nz = cfg.Nz
    x = np.asarray(x, dtype)
    bcs = np.asarray(bcs, dtype)
    o, tape = mlp_forward(layers, m.acts, x)
    if cfg.smooth_NN:
        o = o @ m.F_int.T
    n = x.shape[0]
    F = np.zeros((n, Nz + 1), dtype)
    F[:, 1:Nz] = o
    if cfg.model != WIND_MIXING:
        F[:, 0], F[:, Nz] = bcs[:, 0], bcs[:, 1]
        return (F, tape) if want_tape else F
    sg, mu = cfg.sigma, cfg.mu
    s0 = -mu[3 + k] / sg[3 + k]
    if not cfg.zero_weights:
        F[:, 0], F[:, Nz] = bcs[:, 2 * k], bcs[:, 2 * k + 1]
    u, v, T = x[:, :Nz], x[:, Nz:2 * Nz], x[:, 2 * Nz:]
    cs = sg[k] / sg[3 + k] / cfg.H
    if cfg.modified_pacanowski_philander:
        gu, gv, gT = _face_grad(u, Nz), _face_grad(v, Nz), _face_grad(T, Nz)
        eps = cfg.eps
        Ri = cfg.H * cfg.g * cfg.alpha * sg[2] * (gT + eps) / ((sg[0] * (gu + eps)) ** 2 + (sg[1] * (gv + eps)) ** 2)
        if cfg.smooth_Ri:
            Ri = Ri @ m.F_face.T
        nu = cfg.nu0 + cfg.nu_minus * (1 - np.tanh((Ri - cfg.Ric) / cfg.dRi)) / 2
        D = cs * (nu / cfg.Pr if k == 2 else nu) * (gu, gv, gT)[k]
        if cfg.zero_weights:
            D[:, 0] = -(bcs[:, 2 * k] - s0)
            D[:, Nz] = -(bcs[:, 2 * k + 1] - s0)
        F = F - D
    elif cfg.convective_adjustment and k == 2:
        F = F - cs * cfg.kappa * np.minimum(0.0, _face_grad(T, Nz))
    return (F, tape) if want_tape else F


def nn_pretrain_loss_and_grad(cfg, k, x, bcs, layers, flux, gradient_scaling):
    """`NN_loss(input, output)` of train_NN (:218-229) for one sample batch of size 1..n (mean over the batch is NOT taken: one value
    per sample) and its gradient w.r.t. the net's parameters, per sample: returns (loss [n], [per-sample packed gradients])."""
    Nz = cfg.Nz
    F, tape = predict_single_flux(cfg, k, x, bcs, layers, True)
    y = np.asarray(flux, np.float64)
    r = F - y
    dg = ((F[:, 1:] - F[:, :-1]) - (y[:, 1:] - y[:, :-1])) * Nz
    loss = (r ** 2).mean(axis=1) + gradient_scaling * (dg ** 2).mean(axis=1)
    Fb = 2.0 / (Nz + 1) * r
    Fb[:, 1:] += gradient_scaling * 2.0 / Nz * dg * Nz
    Fb[:, :-1] -= gradient_scaling * 2.0 / Nz * dg * Nz
    ob = Fb[:, 1:Nz]
    if cfg.smooth_NN:
        ob = ob @ Model(cfg).F_int
    grads = []
    acts = tuple(cfg.activations)
    for i in range(x.shape[0]):
        tp = [(a[i:i + 1], z[i:i + 1]) for a, z in tape]
        _, g = mlp_vjp(layers, acts, tp, ob[i:i + 1])
        grads.append(pack_grads([g]))
    return loss, grads


def train_NN(cfg, k, theta_net, profiles, bcs, fluxes, order, eta, epochs, gradient_scaling=1e-4, beta=(0.9, 0.999), eps=1e-8):
    """One optimiser of train_NN (:238-246): `epochs` calls of `Flux.train!(NN_loss, Flux.params(NN), training_data, opt)` — one ADAM
    update per sample in `order` — each followed by `total_loss(training_data)`.  Returns (theta, [total loss after each epoch])."""
    th = np.array(theta_net, np.float64)
    mm, vv, bt = np.zeros_like(th), np.zeros_like(th), (beta[0], beta[1])
    hist = []
    for _ in range(epochs):
        for i in order:
            layers = unpack(th, cfg.layer_sizes, 1)[0]
            _, g = nn_pretrain_loss_and_grad(cfg, k, profiles[i:i + 1], bcs[i:i + 1], layers, fluxes[i:i + 1], gradient_scaling)
            th, mm, vv, bt = adam_step(th, g[0], mm, vv, eta, beta, eps, bt)
        layers = unpack(th, cfg.layer_sizes, 1)[0]
        hist.append(float(nn_pretrain_loss_and_grad(cfg, k, profiles, bcs, layers, fluxes, gradient_scaling)[0].mean()))
    return th, hist
