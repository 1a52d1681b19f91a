"""fc32: iteration time by tile width and matrix arithmetic (COLNDE_FC_CW = 16: 16-column tiles on f32 MFMA or, exact split, on v_mfma_f32_16x16x32_bf16;
32: the exact-split kernels of engine_fc_split.hip) over column counts — where should engine AUTO hand over?   usage: fc_tile_width.py [Nz ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import colnde
from colnde import synthetic
dev = torch.device("cuda", 0)
for Nz in [int(a) for a in sys.argv[1:]] or [64, 32]:
    for ncol in (8, 64, 256, 512, 1024, 2048, 4096, 8192):
        row = []
        for cw, ma in (("16", "f32_mfma"), ("16", "bf16x3_exact"), ("32", "bf16x3_exact")):
            os.environ["COLNDE_FC_CW"] = cw
            p = synthetic.free_convection_problem(ncol, Nz=Nz)
            h = colnde.ColumnNDE(p.cfg, ncol, matrix_arithmetic=ma)
            x0, bcs, w, wt = (torch.from_numpy(a).to(dev) for a in (p.x0, p.bcs, p.weights, p.weights_truth))
            h.set_problem(x0, bcs)
            truth = h.forward(wt)
            h.set_problem(x0, bcs, truth)
            out = torch.empty(p.cfg.n_params + 8, device=dev)
            sc = [0, 0, 1, 0, 0, 0]
            h.loss_grad(w, sc, out=out); torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(3): h.loss_grad(w, sc, out=out)
            torch.cuda.synchronize()
            row.append((time.time() - t0) / 3 * 1e3)
            pl = h.plan()
            h.close()
        print("Nz %d columns %5d: tile width 16, f32 MFMA %8.2f ms | tile width 16, bf16x3 %8.2f ms | tile width 32, bf16x3 (%d%d%d) %8.2f ms" % (Nz, ncol, row[0], row[1], pl["bf16x3_forward"], pl["bf16x3_adjoint"], pl["bf16x3_dw"], row[2]), flush=True)
os.environ.pop("COLNDE_FC_CW", None)
