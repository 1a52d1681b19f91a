"""`ColumnNDE` — one handle on the HIP tile engine for a set of columns (simulations).

Thin, typed front-end of the C ABI (include/colnde.h).  NumPy arrays go through the host-pointer entry
points; torch CUDA(ROCm) tensors go through the `_dev` twins on torch's current stream (PyTorch is used
for device memory, streams and torch.distributed only).  The reference-named closures
(`NDE`, `NDE!`, `loss_NDE`, `loss_gradient_NDE`, `solve_nde`, …) live in wind_mixing.py / free_convection.py.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import numpy as np

from . import _lib
from .config import MATRIX_ARITHMETIC_NAMES, NDEConfig, matrix_arithmetic_id, to_c_config

KERNEL_IDS = {"forward": 0, "adjoint": 1, "reduce": 2, "rhs": 3, "infer": 4, "dw1": 5, "convadj": 6, "adam": 7, "impldiff": 8}
ENGINE_AUTO, ENGINE_TILE16, ENGINE_REGTILE, ENGINE_FC32 = 0, 1, 2, 3


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("expected shape %s, got %s" % (tuple(shape), tuple(a.shape)))
    return a


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def min_substeps(cfg: NDEConfig) -> int:
    """`colnde_min_substeps`: the least RK4 sub-steps per save interval inside the diffusive stability bound (no GPU needed)."""
    c, keep = to_c_config(cfg, 1, 0, 0)
    n = _lib.lib().colnde_min_substeps(ctypes.byref(c))
    if n < 0:
        raise _lib.ColndeError(_lib.lib().colnde_last_error().decode("utf-8", "replace"))
    return int(n)


def rkc_stages(cfg: NDEConfig) -> int:
    """`colnde_rkc_stages`: stages per RKC2 step this configuration runs with (no GPU needed)."""
    c, keep = to_c_config(cfg, 1, 0, 0)
    n = _lib.lib().colnde_rkc_stages(ctypes.byref(c))
    if n < 0:
        raise _lib.ColndeError(_lib.lib().colnde_last_error().decode("utf-8", "replace"))
    return int(n)


class ColumnNDE:
    def __init__(self, cfg: NDEConfig, n_columns: int, device: int = 0, engine: int = 0, matrix_arithmetic="bf16x3_exact"):
        """matrix_arithmetic: "bf16x3_exact" (default: f32 products as six bf16 MFMA products of exact three-way operand splits, f32 accumulation,
        wherever the engine has a split kernel) or "f32_mfma" (v_mfma_f32_* throughout) — include/colnde.h COLNDE_MATRIX_*."""
        cfg.validate()
        self.cfg = cfg
        self.n_columns = int(n_columns)
        self.device = int(device)
        self._h = ctypes.c_void_p()
        L = _lib.lib()
        c, keep = to_c_config(cfg, n_columns, device, engine, matrix_arithmetic)
        _lib.check(L.colnde_create(ctypes.byref(c), ctypes.byref(self._h)))
        self._L = L
        self.n_params = L.colnde_n_params(self._h)
        assert self.n_params == cfg.n_params
        self.n_columns_total = self.n_columns
        self.engine = L.colnde_engine(self._h)      # engine actually selected (ENGINE_TILE16 or ENGINE_REGTILE)

    # ---- lifetime -------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.colnde_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- configuration --------------------------------------------------------------------------------
    def set_stream(self, stream_ptr: int):
        _lib.check(self._L.colnde_set_stream(self._h, ctypes.c_void_p(stream_ptr)))

    def use_torch_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def set_matrix_arithmetic(self, matrix_arithmetic):
        """Switch the handle between "bf16x3_exact" and "f32_mfma" (tapes and plans do not depend on it: same-handle A/B)."""
        _lib.check(self._L.colnde_set_matrix_arithmetic(self._h, matrix_arithmetic_id(matrix_arithmetic)))

    @property
    def matrix_arithmetic(self) -> str:
        return MATRIX_ARITHMETIC_NAMES[self._L.colnde_matrix_arithmetic(self._h)]

    def set_global_columns(self, n_total: int):
        _lib.check(self._L.colnde_set_global_columns(self._h, int(n_total)))
        self.n_columns_total = int(n_total)

    def set_profiling(self, on: bool = True):
        _lib.check(self._L.colnde_set_profiling(self._h, int(bool(on))))

    def kernel_time(self, which: str):
        ms = ctypes.c_float(0)
        n = ctypes.c_int(0)
        _lib.check(self._L.colnde_kernel_time(self._h, KERNEL_IDS[which], ctypes.byref(ms), ctypes.byref(n)))
        return float(ms.value), int(n.value)

    def plan(self):
        """How the gradient path runs (valid after the first loss_grad): engine, column blocks, which tapes are in use."""
        info = (ctypes.c_int * 8)()
        _lib.check(self._L.colnde_plan(self._h, info))
        return dict(engine=info[0], block_columns=info[1], n_blocks=info[2], z1_taped=bool(info[3]) and info[0] == ENGINE_REGTILE,
                    time_segments=info[3] if info[0] == ENGINE_FC32 else 0,
                    dw_taped=bool(info[4]), dw_slices=info[5], split_forward=bool(info[6] & 1), split_adjoint=bool(info[6] & 2), split_rich_tape=bool(info[6] & 4),
                    approximate_gradient=bool(info[7] & 1),
                    # which kernel families run the exact three-way bf16 split (the others: f32 MFMA)
                    bf16x3_forward=bool(info[7] & 2), bf16x3_adjoint=bool(info[7] & 4), bf16x3_dw=bool(info[7] & 8),
                    matrix_arithmetic=self.matrix_arithmetic)

    def describe(self) -> str:
        """`colnde_describe`: the plan spelled out, plus every COLNDE_* tuning switch set in this process that the library reads."""
        need = self._L.colnde_describe(self._h, None, 0)
        if need < 0:
            _lib.check(1)
        buf = ctypes.create_string_buffer(need)
        self._L.colnde_describe(self._h, buf, need)
        return buf.value.decode()

    def pretrain_flux(self, flux_type: int, theta, m, v, profiles, bcs, fluxes, order, gradient_scaling: float, opt, update: bool = True):
        """`colnde_pretrain_flux_dev`: one `Flux.train!` pass (one ADAM update per sample, in `order`) over device tensors; `opt` is a
        flux_compat.ADAM whose running powers are advanced.  Returns the mean per-sample loss (update=False: at fixed weights)."""
        self._chk_dev(theta, (self.n_params,))
        n = int(profiles.shape[0])
        self._chk_dev(profiles, (n, self.cfg.n_state))
        self._chk_dev(bcs, (n, self.cfg.n_bc))
        self._chk_dev(fluxes, (n, self.cfg.Nz + 1))
        if update:
            self._chk_dev(m, (self.n_params,))
            self._chk_dev(v, (self.n_params,))
        import torch
        if order is not None and (order.dtype != torch.int32 or not order.is_cuda or order.numel() != n):
            raise ValueError("order must be an int32 device tensor of n entries")
        self.use_torch_stream()
        bt = (ctypes.c_double * 2)(*opt.beta_t)
        loss = ctypes.c_float(0)
        _lib.check(self._L.colnde_pretrain_flux_dev(self._h, int(flux_type), theta.data_ptr(), m.data_ptr() if m is not None else None,
                                                    v.data_ptr() if v is not None else None, profiles.data_ptr(), bcs.data_ptr(),
                                                    fluxes.data_ptr(), order.data_ptr() if order is not None else None, n,
                                                    float(gradient_scaling), opt.eta, opt.beta[0], opt.beta[1], opt.eps, bt, int(bool(update)),
                                                    ctypes.byref(loss)))
        if update:
            opt.beta_t = [bt[0], bt[1]]
        return float(loss.value)

    def reset_kernel_times(self):
        _lib.check(self._L.colnde_reset_kernel_times(self._h))

    # ---- problem data ---------------------------------------------------------------------------------
    def set_problem(self, x0, bcs, truth=None):
        c = self.cfg
        if _is_torch(x0):
            self._chk_dev(x0, (self.n_columns, c.n_state))
            self._chk_dev(bcs, (self.n_columns, c.n_bc))
            if truth is not None:
                self._chk_dev(truth, (self.n_columns, c.n_save, c.n_state))
            self.use_torch_stream()
            _lib.check(self._L.colnde_set_problem_dev(self._h, x0.data_ptr(), bcs.data_ptr(),
                                                      truth.data_ptr() if truth is not None else None))
            return
        x0 = _f32(x0, (self.n_columns, c.n_state))
        bcs = _f32(bcs, (self.n_columns, c.n_bc))
        tr = _f32(truth, (self.n_columns, c.n_save, c.n_state)) if truth is not None else None
        _lib.check(self._L.colnde_set_problem(self._h, _ptr(x0), _ptr(bcs), _ptr(tr)))

    def _chk_dev(self, t, shape):
        import torch
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == tuple(shape)):
            raise ValueError("expected a contiguous float32 device tensor of shape %s" % (tuple(shape),))
        if t.device.index != self.device:
            raise ValueError("tensor on device %s, handle on device %d" % (t.device, self.device))

    # ---- one RHS evaluation ---------------------------------------------------------------------------
    def rhs(self, x, weights, bcs, t: float = 0.0):
        c = self.cfg
        if _is_torch(x):
            import torch
            n = x.shape[0]
            self._chk_dev(x, (n, c.n_state))
            self._chk_dev(weights, (self.n_params,))
            self._chk_dev(bcs, (n, c.n_bc))
            dx = torch.empty_like(x)
            self.use_torch_stream()
            _lib.check(self._L.colnde_rhs_dev(self._h, x.data_ptr(), weights.data_ptr(), bcs.data_ptr(), float(t),
                                              dx.data_ptr(), n))
            return dx
        x = _f32(x)
        n = x.shape[0]
        x = _f32(x, (n, c.n_state))
        w = _f32(weights, (self.n_params,))
        b = _f32(bcs, (n, c.n_bc))
        dx = np.empty_like(x)
        _lib.check(self._L.colnde_rhs(self._h, _ptr(x), _ptr(w), _ptr(b), float(t), _ptr(dx), n))
        return dx

    # ---- flux diagnostics -----------------------------------------------------------------------------
    def flux(self, x, weights, bcs, t: float = 0.0):
        """`predict_flux` (wind_mixing/src/NDE_training.jl:83-147): [n][n_nets][Nz + 1] face fluxes (scaled) of n states; T-only models: the wT that the
        dataset-level `solve_nde` re-evaluates per saved step (free_convection/src/solve.jl:32-46)."""
        c = self.cfg
        nn = 3 if c.n_state == 3 * c.Nz else 1
        if _is_torch(x):
            import torch
            n = x.shape[0]
            self._chk_dev(x, (n, c.n_state))
            self._chk_dev(weights, (self.n_params,))
            self._chk_dev(bcs, (n, c.n_bc))
            fl = torch.empty((n, nn, c.Nz + 1), dtype=torch.float32, device=x.device)
            self.use_torch_stream()
            _lib.check(self._L.colnde_flux_dev(self._h, x.data_ptr(), weights.data_ptr(), bcs.data_ptr(), float(t), fl.data_ptr(), n))
            return fl
        x = _f32(x)
        n = x.shape[0]
        x = _f32(x, (n, c.n_state))
        w = _f32(weights, (self.n_params,))
        b = _f32(bcs, (n, c.n_bc))
        fl = np.empty((n, nn, c.Nz + 1), dtype=np.float32)
        _lib.check(self._L.colnde_flux(self._h, _ptr(x), _ptr(w), _ptr(b), float(t), _ptr(fl), n))
        return fl

    def loss_per_tstep(self, weights):
        """`loss_per_tstep` (wind_mixing/src/loss.jl:44-46) of the six profile terms: [n_columns][6][n_save], unscaled mse per save point."""
        shape = (self.n_columns, 6, self.cfg.n_save)
        if _is_torch(weights):
            import torch
            self._chk_dev(weights, (self.n_params,))
            out = torch.empty(shape, dtype=torch.float32, device=weights.device)
            self.use_torch_stream()
            _lib.check(self._L.colnde_loss_per_tstep_dev(self._h, weights.data_ptr(), out.data_ptr()))
            return out
        w = _f32(weights, (self.n_params,))
        out = np.empty(shape, dtype=np.float32)
        _lib.check(self._L.colnde_loss_per_tstep(self._h, _ptr(w), _ptr(out)))
        return out

    # ---- error-controlled time stepping ---------------------------------------------------------------
    @property
    def substeps(self) -> int:
        """Sub-steps per save interval in use (cfg.substeps, or what the handle chose from reltol)."""
        return int(self._L.colnde_substeps(self._h))

    def set_substeps(self, substeps: int) -> None:
        """colnde_set_substeps: impose a sub-step count (the MAX over ranks of what each shard chose — colnde.distributed.agree_substeps); before the
        first loss_grad of this handle."""
        _lib.check(self._L.colnde_set_substeps(self._h, int(substeps)))

    @property
    def n_steps(self) -> int:
        """Time steps of one solve with the sub-step count IN USE (cfg.n_steps is 0 for a config created with substeps = 0)."""
        return (self.cfg.n_save - 1) * self.substeps

    def error_estimate(self, weights) -> float:
        """Richardson estimate of the solve's error at the current sub-step count against one at twice the count, in the integrator's mixed norm
        max |e| / (1e-3 + |u|) (include/colnde.h: colnde_error_estimate) — what `reltol` bounds."""
        est = ctypes.c_float(0)
        if _is_torch(weights):
            self._chk_dev(weights, (self.n_params,))
            self.use_torch_stream()
            _lib.check(self._L.colnde_error_estimate_dev(self._h, weights.data_ptr(), ctypes.byref(est)))
        else:
            w = _f32(weights, (self.n_params,))
            _lib.check(self._L.colnde_error_estimate(self._h, _ptr(w), ctypes.byref(est)))
        return float(est.value)

    def choose_substeps(self, weights, reltol: float = 0.0):
        """The least power-of-two sub-step count (not below the stability bound) whose error estimate meets `reltol` (0: cfg.reltol); the handle
        keeps it.  Returns (substeps, estimate).  Before the first loss_grad only: the count sizes the tapes."""
        w = _f32(weights.cpu().numpy() if _is_torch(weights) else weights, (self.n_params,))
        s, est = ctypes.c_int(0), ctypes.c_float(0)
        _lib.check(self._L.colnde_choose_substeps(self._h, _ptr(w), float(reltol), ctypes.byref(s), ctypes.byref(est)))
        return int(s.value), float(est.value)

    # ---- forward solve --------------------------------------------------------------------------------
    def forward(self, weights, out=None):
        c = self.cfg
        shape = (self.n_columns, c.n_save, c.n_state)
        if _is_torch(weights):
            import torch
            self._chk_dev(weights, (self.n_params,))
            sol = out if out is not None else torch.empty(shape, dtype=torch.float32, device=weights.device)
            self._chk_dev(sol, shape)
            self.use_torch_stream()
            _lib.check(self._L.colnde_forward_dev(self._h, weights.data_ptr(), sol.data_ptr()))
            return sol
        w = _f32(weights, (self.n_params,))
        sol = np.empty(shape, dtype=np.float32)
        _lib.check(self._L.colnde_forward(self._h, _ptr(w), _ptr(sol)))
        return sol

    # ---- losses ---------------------------------------------------------------------------------------
    def loss(self, weights, scalings: Sequence[float]):
        sc = (ctypes.c_float * 6)(*[float(s) for s in scalings])
        if _is_torch(weights):
            import torch
            self._chk_dev(weights, (self.n_params,))
            out = torch.empty(8, dtype=torch.float32, device=weights.device)
            self.use_torch_stream()
            _lib.check(self._L.colnde_loss_dev(self._h, weights.data_ptr(), sc, out.data_ptr()))
            return out
        w = _f32(weights, (self.n_params,))
        terms = (ctypes.c_float * 6)()
        total = ctypes.c_float(0)
        _lib.check(self._L.colnde_loss(self._h, _ptr(w), sc, terms, ctypes.byref(total)))
        return float(total.value), np.array(list(terms), dtype=np.float32)

    def loss_grad(self, weights, scalings: Sequence[float], out=None):
        """NumPy: (total, terms[6], grad[n_params]).  torch: one device tensor [grad; terms(6); total; 0]
        (the buffer to all-reduce when columns are sharded over ranks)."""
        sc = (ctypes.c_float * 6)(*[float(s) for s in scalings])
        if _is_torch(weights):
            import torch
            self._chk_dev(weights, (self.n_params,))
            if out is None:
                out = torch.empty(self.n_params + 8, dtype=torch.float32, device=weights.device)
            self._chk_dev(out, (self.n_params + 8,))
            self.use_torch_stream()
            _lib.check(self._L.colnde_loss_grad_dev(self._h, weights.data_ptr(), sc, out.data_ptr()))
            return out
        w = _f32(weights, (self.n_params,))
        terms = (ctypes.c_float * 6)()
        total = ctypes.c_float(0)
        grad = np.empty(self.n_params, dtype=np.float32)
        _lib.check(self._L.colnde_loss_grad(self._h, _ptr(w), sc, terms, ctypes.byref(total), _ptr(grad)))
        return float(total.value), np.array(list(terms), dtype=np.float32), grad

    # ---- embedded inference ---------------------------------------------------------------------------
    def infer_dz_wT(self, weights, T, top_flux, Lz: float):
        """+∂z wT, what `compute_neural_network_forcing!` stores in `params.∂z_wT_NN` (double_gyre_nn.jl:165); `infer_forcing` is its negative (:135)."""
        return self.infer_forcing(weights, T, top_flux, Lz, _dz_wT=True)

    def infer_forcing(self, weights, T, top_flux, Lz: float, _dz_wT: bool = False):
        Nz = self.cfg.Nz
        if _is_torch(T):
            import torch
            n = T.shape[0]
            self._chk_dev(T, (n, Nz))
            self._chk_dev(top_flux, (n,))
            self._chk_dev(weights, (self.n_params,))
            out = torch.empty_like(T)
            self.use_torch_stream()
            fn = self._L.colnde_infer_dz_wT_dev if _dz_wT else self._L.colnde_infer_forcing_dev
            _lib.check(fn(self._h, weights.data_ptr(), T.data_ptr(), top_flux.data_ptr(), float(Lz), out.data_ptr(), n))
            return out
        T = _f32(T)
        n = T.shape[0]
        T = _f32(T, (n, Nz))
        tf = _f32(top_flux, (n,))
        w = _f32(weights, (self.n_params,))
        out = np.empty_like(T)
        fn = self._L.colnde_infer_dz_wT if _dz_wT else self._L.colnde_infer_forcing
        _lib.check(fn(self._h, _ptr(w), _ptr(T), _ptr(tf), float(Lz), _ptr(out), n))
        return out

    # ---- the steps either side of the hot path --------------------------------------------
    def convective_adjustment(self, T, dt: float, dz: float, K: float, halo_bottom=None, halo_top=None, out=None):
        """`convective_adjustment!(model, Δt, K)` (free_convection/double_gyre_nn.jl:27-62) on [n][Nz] columns."""
        Nz = self.cfg.Nz
        if _is_torch(T):
            import torch
            n = T.shape[0]
            self._chk_dev(T, (n, Nz))
            for hl in (halo_bottom, halo_top):
                if hl is not None:
                    self._chk_dev(hl, (n,))
            if out is None:
                out = torch.empty_like(T)
            self._chk_dev(out, (n, Nz))
            self.use_torch_stream()
            _lib.check(self._L.colnde_convective_adjustment_dev(
                self._h, T.data_ptr(), halo_bottom.data_ptr() if halo_bottom is not None else None,
                halo_top.data_ptr() if halo_top is not None else None, float(dt), float(dz), float(K), out.data_ptr(), n))
            return out
        T = _f32(T)
        n = T.shape[0]
        T = _f32(T, (n, Nz))
        hb = _f32(halo_bottom, (n,)) if halo_bottom is not None else None
        ht = _f32(halo_top, (n,)) if halo_top is not None else None
        res = np.empty_like(T)
        _lib.check(self._L.colnde_convective_adjustment(self._h, _ptr(T), _ptr(hb), _ptr(ht), float(dt), float(dz), float(K),
                                                        _ptr(res), n))
        return res

    def implicit_diffusion(self, u, v, T, dt: float, dz: float, params, convective_adjustment: bool = False, halo_bottom=None, out=None):
        """`modified_pacanowski_philander!(model, constants, Δt, p, convective_adjustment)` (wind_mixing/src/NDE_oceananigans.jl:61-101)
        on [n][Nz] columns of u, v, T.  params = (ν₀, ν₋, ΔRi, Riᶜ, Pr, α, g); halo_bottom [3][n] or None.  Returns (u′, v′, T′);
        `out` = a triple of device tensors (each may be its own input: in place)."""
        Nz = self.cfg.Nz
        pr = (ctypes.c_float * 7)(*[float(x) for x in params])
        if _is_torch(T):
            import torch
            n = T.shape[0]
            for a in (u, v, T):
                self._chk_dev(a, (n, Nz))
            if halo_bottom is not None:
                self._chk_dev(halo_bottom, (3, n))
            if out is None:
                out = tuple(torch.empty_like(T) for _ in range(3))
            for a in out:
                self._chk_dev(a, (n, Nz))
            self.use_torch_stream()
            _lib.check(self._L.colnde_implicit_diffusion_dev(
                self._h, u.data_ptr(), v.data_ptr(), T.data_ptr(), halo_bottom.data_ptr() if halo_bottom is not None else None,
                float(dt), float(dz), pr, int(bool(convective_adjustment)), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), n))
            return out
        T = _f32(T)
        n = T.shape[0]
        u, v, T = _f32(u, (n, Nz)), _f32(v, (n, Nz)), _f32(T, (n, Nz))
        hb = _f32(halo_bottom, (3, n)) if halo_bottom is not None else None
        res = tuple(np.empty_like(T) for _ in range(3))
        _lib.check(self._L.colnde_implicit_diffusion(self._h, _ptr(u), _ptr(v), _ptr(T), _ptr(hb), float(dt), float(dz), pr,
                                                     int(bool(convective_adjustment)), _ptr(res[0]), _ptr(res[1]), _ptr(res[2]), n))
        return res

    def adam_step(self, weights, grad, m, v, eta: float, beta=(0.9, 0.999), eps: float = 1e-8, beta_t=None):
        """One fused `Flux.Optimise.ADAM` apply!/update! on device vectors (in place).  beta_t = running powers (β₁ᵗ, β₂ᵗ)."""
        import torch
        n = weights.numel()
        for t in (weights, m, v):
            self._chk_dev(t, (n,))
        if not (grad.is_cuda and grad.dtype == torch.float32 and grad.is_contiguous() and grad.numel() >= n):
            raise ValueError("grad must be a contiguous float32 device tensor with at least %d elements" % n)
        bt = beta if beta_t is None else beta_t
        self.use_torch_stream()
        _lib.check(self._L.colnde_adam_step_dev(self._h, weights.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), float(eta),
                                                float(beta[0]), float(beta[1]), float(eps), float(bt[0]), float(bt[1]), n))
        return weights

    # ---- data preparation on device (wind_mixing/src/data_containers.jl:343-427) -------------------------
    def coarse_grain(self, x, n: int, location: str = "center"):
        """`coarse_grain(Φ, n, Center)` / `coarse_grain_linear_interpolation(Φ, n, Face)` on the rows of a device tensor [rows, N]."""
        import torch
        if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2):
            raise ValueError("expected a contiguous float32 device tensor [rows, N]")
        out = torch.empty(x.shape[0], int(n), dtype=torch.float32, device=x.device)
        self.use_torch_stream()
        _lib.check(self._L.colnde_coarse_grain_dev(self._h, x.data_ptr(), x.shape[0], x.shape[1], int(n),
                                                   {"center": 0, "face": 1}[location], out.data_ptr()))
        return out

    def zscore(self, x):
        """`ZeroMeanUnitVarianceScaling(data)` and its application: returns (scaled, mu_sigma) — both stay on the device."""
        import torch
        if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
            raise ValueError("expected a contiguous float32 device tensor")
        ms = torch.empty(2, dtype=torch.float32, device=x.device)
        out = torch.empty_like(x)
        self.use_torch_stream()
        _lib.check(self._L.colnde_zscore_stats_dev(self._h, x.data_ptr(), x.numel(), ms.data_ptr()))
        _lib.check(self._L.colnde_scale_dev(self._h, x.data_ptr(), x.numel(), ms.data_ptr(), out.data_ptr()))
        return out, ms
