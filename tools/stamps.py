"""Diagnostic: per-phase cycle shares of one adjoint stage (workgroup 0, wave 0) from the -DCOLNDE_STAMPS build.
Usage (GPU box): make -C climateparameterizations.jl_amd/csrc stamps && python tools/stamps.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import colnde
from colnde import _lib, synthetic
# (the net-split kernels' stamps live in the three-wave variants: the four-wave ones carry none)
os.environ.setdefault("COLNDE_T16_FWD_HELPER", "0")
FWD = "--fwd" in sys.argv      # forward-kernel stamps: build with -DCOLNDE_STAMPS -DCOLNDE_STAMPS_FWD into libcolnde_stamps_fwd.so
if FWD: sys.argv.remove("--fwd")
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libcolnde_stamps_fwd.so" if FWD else "libcolnde_stamps.so")
L = _lib.lib()
L.colnde_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
FC64 = "--fc64" in sys.argv     # free convection 64-256-256-63 (tile16, taped weight gradients)
if FC64: sys.argv.remove("--fc64")
ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 33
p = synthetic.free_convection_problem(ncol, Nz=64, n_save=frames) if FC64 else synthetic.wind_mixing_problem(ncol, n_frames=frames)
nde = colnde.ColumnNDE(p.cfg, ncol)
nde.set_problem(p.x0, p.bcs)
truth = nde.forward(p.weights_truth)
nde.set_problem(p.x0, p.bcs, truth)
for _ in range(12 if ncol >= 16384 else 1):        # back-to-back launches, so that the clock stamp sees the sustained state
    nde.loss_grad(p.weights, [0, 0, 1, 0, 0, 0] if FC64 else [1, 1, 1, 5e-3, 5e-3, 5e-3])
buf = (ctypes.c_ulonglong * 16)()
_lib.check(L.colnde_debug_stamps(nde._h, buf))
SPLIT = nde.engine == 1 and not FC64 and os.environ.get("COLNDE_T16_FWD_SPLIT", "1") != "0" and os.environ.get("COLNDE_T16_ADJ_SPLIT", "1") != "0"
if SPLIT and FWD:
    names = ["x tape store + top flux", "layer 1 (4 chains, activation, tape stores)", "layers 2, 3", "physics (flux, tendency, coefficients)",
             "RK4 bookkeeping + exchange + barrier"]
elif SPLIT and os.environ.get("COLNDE_T16_ADJ_HELPER", "1") != "0":        # the four-wave adjoint: net wave 0's segments
    names = ["tape-only work before (B): activations, x / a parts, operand fetch", "wait at (B) for the helper's dO", "W3^T and W2^T chains",
             "delta stores + bias sums", "W1^T chains + part write", "wait at (A)"]
elif SPLIT:        # net-split kernels of the latency points (wave 0 = net 0)
    names = ["prefetch issue + kbar + physics pullback", "activation pairs + x / a stores", "W3^T and W2^T chains", "delta stores + bias sums",
             "W1^T chains + exchange write", "barrier + sum of the three parts"]
elif nde.engine == 2 and FWD:
    names = ["X tape store + top flux", "layer 1 (10 chains, Z1 tape store, activation)", "layers 2, 3", "physics", "RK4 update"]
elif nde.engine == 2:
    # (stamp 6 sits behind the net loop: the W1^T chains of nets 0 and 1 are charged to the next net's stamp 1)
    names = ["kbar + physics pullback + dO park", "net 0 activation pairs + W1^T chains of nets 0, 1", "L2 chain (3 nets)", "dW3 + W3^T + dZ2",
             "dW2 (transposes + outer)", "W2^T + dZ1 + tape2 store", "W1^T chains of net 2", "(net 0's Z1 loads: issue -> landed)"]
else:
    names = ["tape load + kbar", "mlp_forward", "physics_vjp", "mlp_backward", "dW tiles", "bias + xbar sum + barrier"]
v = np.array(list(buf)[:len(names)], dtype=np.float64)
nstage = p.cfg.n_steps * 4
for n, x in zip(names, v):
    print("%-28s %10.0f cycles/stage  %5.1f %%" % (n, x / nstage, 100 * x / v.sum()))
print("total %.0f cycles/stage" % (v.sum() / nstage))
if FWD and not SPLIT:
    print("whole kernel (workgroup 0, wave 0): %d s_memtime ticks in %.3f ms (s_memrealtime, 100 MHz) -> %.3f ticks/ns; stage loop share %.1f %%"
          % (buf[6], buf[7] / 1e5, buf[6] / (buf[7] * 10.0), 100.0 * v.sum() / buf[6]))
if nde.engine == 2 or SPLIT:
    # clock the stamped kernel ran at (MI355X_MICROARCH 'DVFS give-back' item 6); meaningful on launches of >= 10 ms (pass e.g. 32768 289)
    if buf[9]:
        print("whole kernel (workgroup 0, wave 0): %d shader ticks in %.3f ms of the 100 MHz reference -> in-kernel clock %.3f GHz"
              % (buf[8], buf[9] / 1e5, buf[8] / (buf[9] * 10.0)))
    raise SystemExit(0)
fine = np.array(list(buf)[8:13], dtype=np.float64)
for n, x in zip(["layer setup", "job prologue (bias)", "MFMA chain", "epilogue (act + store)", "barrier"], fine):
    print("  mlp_forward/%-24s %9.0f cycles/stage (all kernels' forward passes of wave 0)" % (n, x / nstage))
