"""Differential fuzz of the free-convection gradient paths: the fc32 engine (both tile widths, column blocks, time segments, both matrix arithmetics)
against tile16 on random column counts, level counts, save points, sub-steps, models (FreeConvectionNDE / ConvectiveAdjustmentNDE) and steppers; prints the
worst disagreement.  Usage (GPU box): python tools/fuzz_fc.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import colnde
from colnde import synthetic
from colnde.nde import ENGINE_FC32, ENGINE_TILE16

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = dict(sol=0.0, loss=0.0, grad=0.0)
KEYS = ("COLNDE_FC_CW", "COLNDE_FC_BLOCK", "COLNDE_FC_SEG")
for c in range(cases):
    Nz = int(rng.choice([32, 64]))
    ncol = int(rng.choice([1, 5, 16, 31, 32, 33, 70, 130, 300]))
    n_save = int(rng.choice([2, 3, 5, 9]))
    ca = bool(rng.integers(2))
    rkc = ca and bool(rng.integers(2))
    sub = 1 if rkc else int(rng.choice([1, 2, 3]))
    t_end = float(rng.choice([0.005, 0.01, 0.02]))
    p = synthetic.free_convection_problem(ncol, Nz=Nz, n_save=n_save, substeps=sub, t_end=t_end, convective_adjustment=ca)
    cfg = p.cfg.with_(stepper="rkc2") if rkc else p.cfg
    if not rkc:
        need = colnde.min_substeps(cfg)
        if need > cfg.substeps:
            cfg = cfg.with_(substeps=int(need))
    env = {"COLNDE_FC_CW": str(int(rng.choice([16, 32, 32])))}
    arith = str(rng.choice(["bf16x3_exact", "bf16x3_exact", "f32_mfma"]))        # fc32's matrix arithmetic (tile16, the reference here, runs f32 MFMA)
    mode = int(rng.integers(3))
    if mode == 1:
        env["COLNDE_FC_BLOCK"] = "32"
    elif mode == 2 and n_save > 2:
        env["COLNDE_FC_SEG"] = str(int(rng.integers(1, n_save - 1)))
    sc = [0, 0, 1, 0, 0, 0]
    res = {}
    truth = None
    msg = ""
    for label, eng in (("tile16", ENGINE_TILE16), ("fc32", 0)):
        for k in KEYS:
            os.environ.pop(k, None)
        if label == "fc32":
            os.environ.update(env)
        try:
            with colnde.ColumnNDE(cfg, ncol, engine=eng, matrix_arithmetic=arith if label == "fc32" else "f32_mfma") as nde:
                nde.set_problem(p.x0, p.bcs)
                if truth is None:
                    truth = nde.forward(p.weights_truth)
                nde.set_problem(p.x0, p.bcs, truth)
                sol = nde.forward(p.weights)
                tot, terms, g = nde.loss_grad(p.weights, sc)
                res[label] = (sol, tot, g.astype(np.float64), nde.plan(), nde.engine)
        except colnde.ColndeError as e:
            res[label] = None
            msg = str(e)[:100]
    for k in KEYS:
        os.environ.pop(k, None)
    tag = "case %2d Nz %d ncol %3d saves %d sub %d %s%s env %s %s" % (c, Nz, ncol, n_save, cfg.substeps, "CA" if ca else "FC", "+RKC2" if rkc else "", env, arith)
    if res["tile16"] is None or res["fc32"] is None:
        print(tag + ": refused (%s)" % msg, flush=True)
        continue
    a, b = res["fc32"], res["tile16"]
    assert a[4] == ENGINE_FC32, a[4]
    ds = np.abs(a[0] - b[0]).max()
    dl = abs(a[1] - b[1]) / max(abs(b[1]), 1e-30)
    dg = np.linalg.norm(a[2] - b[2]) / max(np.linalg.norm(b[2]), 1e-30)
    worst["sol"] = max(worst["sol"], ds); worst["loss"] = max(worst["loss"], dl); worst["grad"] = max(worst["grad"], dg)
    flag = "  <-- LOOK" if (ds > 2e-5 or dg > (5e-2 if rkc else 1e-3) or not np.isfinite(dg)) else ""
    print(tag + ": sol %.1e loss %.1e grad %.1e segs %s blocks %s bf16 %d%d%d%s" % (ds, dl, dg, a[3].get("time_segments"), a[3].get("n_blocks"), a[3].get("bf16x3_forward", 0), a[3].get("bf16x3_adjoint", 0), a[3].get("bf16x3_dw", 0), flag), flush=True)
print("worst disagreement fc32 vs tile16 over %d cases: sol %.2e, loss %.2e (relative), gradient %.2e (relative L2)" % (cases, worst["sol"], worst["loss"], worst["grad"]))
