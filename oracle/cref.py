"""ctypes front-end of oracle/colnde_ref.c — TEST INFRASTRUCTURE ONLY (parity unpinned, see nde_oracle.py).
Used by tests/ and by bench.py's `cpu_baseline` leg; never by the product path."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcolnde_ref.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "colnde_ref.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "_build/libcolnde_ref.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        try:
            _lib = ctypes.CDLL(_SO)
        except OSError:
            build(force=True)
            _lib = ctypes.CDLL(_SO)
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if a is not None else None


def _cfg(cfg, n_columns):
    import colnde
    from colnde.config import to_c_config
    return to_c_config(cfg, n_columns)


def rhs(cfg, x, bcs, theta, t=0.0):
    x, bcs, theta = _f(x), _f(bcs), _f(theta)
    c, keep = _cfg(cfg, x.shape[0])
    dx = np.empty_like(x)
    rc = lib().colnde_ref_rhs(ctypes.byref(c), _p(x), _p(bcs), _p(theta), ctypes.c_float(t), _p(dx), x.shape[0])
    assert rc == 0
    return dx


def forward(cfg, x0, bcs, theta, n_threads=0):
    x0, bcs, theta = _f(x0), _f(bcs), _f(theta)
    c, keep = _cfg(cfg, x0.shape[0])
    sol = np.empty((x0.shape[0], cfg.n_save, x0.shape[1]), dtype=np.float32)
    rc = lib().colnde_ref_forward(ctypes.byref(c), _p(x0), _p(bcs), _p(theta), _p(sol), int(n_threads))
    assert rc == 0
    return sol


def loss_grad(cfg, x0, bcs, theta, truth, scalings, n_col_total=0, want_grad=True, want_sol=False, n_threads=0):
    x0, bcs, theta, truth = _f(x0), _f(bcs), _f(theta), _f(truth)
    sc = _f(scalings)
    c, keep = _cfg(cfg, x0.shape[0])
    terms = np.zeros(6, dtype=np.float32)
    total = ctypes.c_float(0)
    grad = np.zeros(cfg.n_params, dtype=np.float32) if want_grad else None
    sol = np.empty((x0.shape[0], cfg.n_save, x0.shape[1]), dtype=np.float32) if want_sol else None
    rc = lib().colnde_ref_loss_grad(ctypes.byref(c), _p(x0), _p(bcs), _p(theta), _p(truth), _p(sc),
                                    ctypes.c_long(n_col_total), _p(terms), ctypes.byref(total), _p(grad), _p(sol),
                                    int(n_threads))
    assert rc == 0
    return float(total.value), terms, grad, sol


def infer_forcing(cfg, T, top_flux, theta, Lz):
    T, top_flux, theta = _f(T), _f(top_flux), _f(theta)
    c, keep = _cfg(cfg, T.shape[0])
    out = np.empty_like(T)
    rc = lib().colnde_ref_infer_forcing(ctypes.byref(c), _p(theta), _p(T), _p(top_flux), ctypes.c_float(Lz), _p(out), T.shape[0])
    assert rc == 0
    return out
