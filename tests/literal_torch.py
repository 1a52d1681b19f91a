"""An *independent*, literal restatement of the reference RHS in torch float64, written the way the
Julia code is written — dense `Dᶜ`/`Dᶠ`/filter matrices, vector concatenations, broadcasting —
instead of the stencil form the oracle uses.  Used only by tests to pin oracle/nde_oracle.py
(values through the forward, gradients through torch autograd).  One column at a time.

Follows wind_mixing/src/NDE_training.jl:46-165, free_convection/src/free_convection_nde.jl:29-38,
free_convection/src/convective_adjustment_nde.jl:33-48, wind_mixing/src/loss.jl:1-9.
"""
import numpy as np
import torch

from oracle import nde_oracle as O

DT = torch.float64


def _act(name, z):
    if name == "identity":
        return z
    if name == "relu":
        return torch.relu(z)
    if name == "mish":
        return z * torch.tanh(torch.nn.functional.softplus(z))
    if name == "swish":
        return z * torch.sigmoid(z)
    if name == "tanh":
        return torch.tanh(z)
    if name == "leakyrelu":
        return torch.maximum(0.01 * z, z)
    raise ValueError(name)


def chains(theta, cfg):
    """re(θ): list of nets, each a list of (W out×in, b), column-major vec (Flux.destructure)."""
    n_nets = 3 if cfg.model == O.WIND_MIXING else 1
    nets, o = [], 0
    s = cfg.layer_sizes
    for _ in range(n_nets):
        layers = []
        for i in range(len(s) - 1):
            W = theta[o:o + s[i] * s[i + 1]].reshape(s[i], s[i + 1]).T   # column-major out×in
            o += s[i] * s[i + 1]
            b = theta[o:o + s[i + 1]]
            o += s[i + 1]
            layers.append((W, b))
        nets.append(layers)
    return nets


def chain_call(layers, acts, x):
    for (W, b), a in zip(layers, acts):
        x = _act(a, W @ x + b)
    return x


def wm_rhs(cfg, x, p_bcs, theta, t=0.0):
    Nz, H, tau, f = cfg.Nz, cfg.H, cfg.tau, cfg.f
    mu_u, mu_v, mu_T, mu_uw, mu_vw, mu_wT = cfg.mu
    s_u, s_v, s_T, s_uw, s_vw, s_wT = cfg.sigma
    D_cell = torch.tensor(O.Dc(Nz, 1 / Nz), dtype=DT)
    D_face = torch.tensor(O.Df(Nz, 1 / Nz), dtype=DT)
    nets = chains(theta, cfg)
    uw_b, uw_t, vw_b, vw_t, wT_b, wT_t = [p_bcs[i] for i in range(6)]
    if cfg.diurnal:
        Q = p_bcs[5]
        wT_t = (Q * np.sin(2 * np.pi / (24 * 60 ** 2) * (t * tau)) / (cfg.alpha * cfg.g) - mu_wT) / s_wT
    u, v, T = x[:Nz], x[Nz:2 * Nz], x[2 * Nz:]
    uw_i, vw_i, wT_i = (chain_call(nets[k], cfg.activations, x) for k in range(3))
    if cfg.smooth_NN:
        Fi = torch.tensor(O.smoothing_filter(Nz - 1, 3), dtype=DT)
        uw_i, vw_i, wT_i = Fi @ uw_i, Fi @ vw_i, Fi @ wT_i
    z = torch.zeros(1, dtype=DT)
    one = lambda s: torch.as_tensor(s, dtype=DT).reshape(1)
    if cfg.zero_weights:
        uw, vw, wT = (torch.cat([z, a, z]) for a in (uw_i, vw_i, wT_i))
    else:
        uw = torch.cat([one(uw_b), uw_i, one(uw_t)])
        vw = torch.cat([one(vw_b), vw_i, one(vw_t)])
        wT = torch.cat([one(wT_b), wT_i, one(wT_t)])
    if cfg.modified_pacanowski_philander:
        eps = cfg.eps
        dudz, dvdz, dTdz = D_face @ u, D_face @ v, D_face @ T
        Bz = H * cfg.g * cfg.alpha * s_T * (dTdz + eps)
        S2 = (s_u * (dudz + eps)) ** 2 + (s_v * (dvdz + eps)) ** 2
        Ri = Bz / S2
        if cfg.smooth_Ri:
            Ri = torch.tensor(O.smoothing_filter(Nz + 1, 3), dtype=DT) @ Ri
        nu = cfg.nu0 + cfg.nu_minus * (1 - torch.tanh((Ri - cfg.Ric) / cfg.dRi)) / 2
        s0 = (-mu_uw / s_uw, -mu_vw / s_vw, -mu_wT / s_wT)
        if cfg.zero_weights:
            nudu = torch.cat([one(-(uw_b - s0[0])), s_u / s_uw / H * nu[1:-1] * dudz[1:-1], one(-(uw_t - s0[0]))])
            nudv = torch.cat([one(-(vw_b - s0[1])), s_v / s_vw / H * nu[1:-1] * dvdz[1:-1], one(-(vw_t - s0[1]))])
            nudT = torch.cat([one(-(wT_b - s0[2])), s_T / s_wT / H * nu[1:-1] / cfg.Pr * dTdz[1:-1], one(-(wT_t - s0[2]))])
        else:
            nudu = s_u / s_uw / H * nu * dudz
            nudv = s_v / s_vw / H * nu * dvdz
            nudT = s_T / s_wT / H * nu * dTdz / cfg.Pr
        uw, vw, wT = uw - nudu, vw - nudv, wT - nudT
    elif cfg.convective_adjustment:
        dTdz = D_face @ T
        wT = wT - s_T / s_wT / H * cfg.kappa * torch.minimum(torch.zeros_like(dTdz), dTdz)
    dudt = -tau / H * s_uw / s_u * (D_cell @ uw) + f * tau / s_u * (s_v * v + mu_v)
    dvdt = -tau / H * s_vw / s_v * (D_cell @ vw) - f * tau / s_v * (s_u * u + mu_u)
    dTdt = -tau / H * s_wT / s_T * (D_cell @ wT)
    return torch.cat([dudt, dvdt, dTdt])


def wm_rhs_inplace(cfg, x, p_bcs, theta, t=0.0):
    """The in-place evaluation RHS `NDE!` with its `predict_flux!`, written as training_postprocessing.jl:55-153 is: face
    vectors whose end entries are preset ONCE to `BC - scaling(0)` (:82-93), interiors overwritten by the nets (:105-107) and
    decremented by the diffusive flux (:126-128), `Ri` WITHOUT the ϵ of the training RHS (:114), the `ν_T` switch that tests
    ∂u∂z (:118-121), and a diurnal top flux refreshed every call WITHOUT the `- scaling(0)` offset (:142-144)."""
    Nz, H, tau, f = cfg.Nz, cfg.H, cfg.tau, cfg.f
    mu_u, mu_v, mu_T, mu_uw, mu_vw, mu_wT = cfg.mu
    s_u, s_v, s_T, s_uw, s_vw, s_wT = cfg.sigma
    D_cell = torch.tensor(O.Dc(Nz, 1 / Nz), dtype=DT)
    D_face = torch.tensor(O.Df(Nz, 1 / Nz), dtype=DT)
    nets = chains(theta, cfg)
    uw_b, uw_t, vw_b, vw_t, wT_b, wT_t = [float(p_bcs[i]) for i in range(6)]
    scal0 = lambda mu, sg: (0.0 - mu) / sg                                   # scalings.φ(0f0)
    if cfg.diurnal:
        Q = float(p_bcs[5])
        top_fn = lambda tt: (Q * np.sin(2 * np.pi / (24 * 60 ** 2) * tt) / (cfg.alpha * cfg.g) - mu_wT) / s_wT
    uw, vw, wT = (torch.zeros(Nz + 1, dtype=DT) for _ in range(3))
    uw[0] = uw_b - scal0(mu_uw, s_uw)
    vw[0] = vw_b - scal0(mu_vw, s_vw)
    wT[0] = wT_b - scal0(mu_wT, s_wT)
    uw[-1] = uw_t - scal0(mu_uw, s_uw)
    vw[-1] = vw_t - scal0(mu_vw, s_vw)
    wT[-1] = (top_fn(0.0) if cfg.diurnal else wT_t) - scal0(mu_wT, s_wT)
    u, v, T = x[:Nz], x[Nz:2 * Nz], x[2 * Nz:]
    # predict_flux!
    uw[1:-1] = chain_call(nets[0], cfg.activations, x)
    vw[1:-1] = chain_call(nets[1], cfg.activations, x)
    wT[1:-1] = chain_call(nets[2], cfg.activations, x)
    dudz, dvdz, dTdz = D_face @ u, D_face @ v, D_face @ T
    Bz = H * cfg.g * cfg.alpha * s_T * dTdz
    S2 = (s_u * dudz) ** 2 + (s_v * dvdz) ** 2
    Ri = Bz / S2                                                              # end faces 0/0: never read
    nu = cfg.nu0 + cfg.nu_minus * (1 - torch.tanh((Ri - cfg.Ric) / cfg.dRi)) / 2
    nu_T = torch.zeros(Nz + 1, dtype=DT)
    if cfg.convective_adjustment:
        for i in range(1, Nz):
            nu_T[i] = nu[i] / cfg.Pr if dudz[i] > 0 else cfg.kappa
    else:
        nu_T = nu / cfg.Pr
    uw[1:-1] = uw[1:-1] - s_u / s_uw / H * nu[1:-1] * dudz[1:-1]
    vw[1:-1] = vw[1:-1] - s_v / s_vw / H * nu[1:-1] * dvdz[1:-1]
    wT[1:-1] = wT[1:-1] - s_T / s_wT / H * nu_T[1:-1] * dTdz[1:-1]
    # NDE!
    if cfg.diurnal:
        wT[-1] = top_fn(t * tau)
    dudt = -tau / H * s_uw / s_u * (D_cell @ uw) + f * tau / s_u * (s_v * v + mu_v)
    dvdt = -tau / H * s_vw / s_v * (D_cell @ vw) - f * tau / s_v * (s_u * u + mu_u)
    dTdt = -tau / H * s_wT / s_T * (D_cell @ wT)
    return torch.cat([dudt, dvdt, dTdt])


def fc_rhs(cfg, T, p_bcs, theta, t=0.0):
    Nz = cfg.Nz
    s_T, s_wT = cfg.sigma[2], cfg.sigma[5]
    Dzc = torch.tensor(O.Dc(Nz, 1 / Nz), dtype=DT)
    nets = chains(theta, cfg)
    wT_i = chain_call(nets[0], cfg.activations, T)
    wT = torch.cat([p_bcs[0].reshape(1), wT_i, p_bcs[1].reshape(1)])
    if cfg.model == O.FREE_CONVECTION:
        return -(Dzc * s_wT / s_T * cfg.tau / cfg.H) @ wT
    Dzf = torch.tensor(O.Df(Nz, 1 / Nz), dtype=DT)
    dz_wT = Dzc @ wT
    dTdz = Dzf @ T
    dzK = Dzc @ torch.minimum(torch.zeros_like(dTdz), cfg.ca_K * dTdz)
    return s_wT / s_T * cfg.tau / cfg.H * (-dz_wT + dzK)


def rhs(cfg, x, bcs, theta, t=0.0):
    if cfg.model == O.WIND_MIXING:
        return wm_rhs_inplace(cfg, x, bcs, theta, t) if cfg.inplace_variant else wm_rhs(cfg, x, bcs, theta, t)
    return fc_rhs(cfg, x, bcs, theta, t)


def solve_rk4(cfg, x0, bcs, theta):
    """sol [n_save, n_state] for one column, same fixed-step RK4 as the oracle."""
    ts = cfg.save_times
    x = x0
    out = [x]
    for i in range(len(ts) - 1):
        dt = (ts[i + 1] - ts[i]) / cfg.substeps
        for s in range(cfg.substeps):
            t = ts[i] + s * dt
            k1 = rhs(cfg, x, bcs, theta, t)
            k2 = rhs(cfg, x + dt / 2 * k1, bcs, theta, t + dt / 2)
            k3 = rhs(cfg, x + dt / 2 * k2, bcs, theta, t + dt / 2)
            k4 = rhs(cfg, x + dt * k3, bcs, theta, t + dt)
            x = x + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
        out.append(x)
    return torch.stack(out)


def total_loss(cfg, sols, truths, scalings):
    """loss_gradient_NDE (NDE_training.jl:303-323) / nde_loss (training.jl:55-62); sols: list of [n_save, n_state]."""
    Nz = cfg.Nz
    mse = lambda a, b: torch.mean((a - b) ** 2)
    if cfg.model != O.WIND_MIXING:
        return scalings[2] * mse(torch.cat(sols), torch.cat(truths))
    D_face = torch.tensor(O.Df(Nz, 1 / Nz), dtype=DT)
    tot = 0.0
    for k in range(3):
        prof = [mse(t[:, k * Nz:(k + 1) * Nz], s[:, k * Nz:(k + 1) * Nz]) for s, t in zip(sols, truths)]
        grad = [mse(t[:, k * Nz:(k + 1) * Nz] @ D_face.T, s[:, k * Nz:(k + 1) * Nz] @ D_face.T) for s, t in zip(sols, truths)]
        tot = tot + scalings[k] * torch.stack(prof).mean() + scalings[3 + k] * torch.stack(grad).mean()
    return tot
