/* abi_smoke.c — the drop-in boundary exercised from plain C, with nothing but include/colnde.h: fill colnde_config by hand,
 * create -> set_problem -> forward -> loss_grad on the committed golden vectors (tests/golden/wind_mixing_mpp.npz dumped to raw
 * float32 files by tests/test_abi_c_harness.py), compare with the expected outputs.  Built with gcc and run by pytest on the GPU
 * box; it is what a C / Julia `ccall` host sees (no Python, no torch in the process).
 *
 *   abi_smoke <dir with x0.f32 bcs.f32 weights.f32 truth.f32 scalings.f32 sol.f32 grad.f32 total.f32> */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "colnde.h"

static float* slurp(const char* dir, const char* name, size_t n) {
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    float* p = (float*)malloc(n * sizeof(float));
    if (fread(p, sizeof(float), n, f) != n) { fprintf(stderr, "%s: short read (want %zu floats)\n", path, n); exit(2); }
    fclose(f);
    return p;
}

#define CHECK(call) do { if ((call) != 0) { fprintf(stderr, "%s failed: %s\n", #call, colnde_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: abi_smoke <fixture dir>\n"); return 2; }
    const char* dir = argv[1];
    enum { NCOL = 4, NZ = 32, NSAVE = 9, NS = 96 };
    float times[NSAVE];
    for (int i = 0; i < NSAVE; i++) times[i] = (float)((double)i / 288.0);

    colnde_config c;
    memset(&c, 0, sizeof c);
    c.model = COLNDE_MODEL_WIND_MIXING;
    c.Nz = NZ;
    c.n_layers = 3;
    c.layer_sizes[0] = 96; c.layer_sizes[1] = 50; c.layer_sizes[2] = 20; c.layer_sizes[3] = 31;
    c.activations[0] = COLNDE_ACT_MISH; c.activations[1] = COLNDE_ACT_MISH; c.activations[2] = COLNDE_ACT_IDENTITY;
    c.modified_pacanowski_philander = 1;
    c.zero_weights = 1;
    c.train_gradient = 1;
    c.H = 256.0f; c.tau = 172800.0f; c.f = 1e-4f; c.g = 9.81f; c.alpha = 1.67e-4f;
    c.nu0 = 1e-4f; c.nu_minus = 1e-1f; c.Ric = 0.25f; c.dRi = 1.0f; c.Pr = 1.0f; c.kappa = 10.0f; c.eps = 1e-7f;
    const float sigma[6] = {0.05f, 0.05f, 0.3f, 2e-4f, 2e-4f, 1e-5f};
    const float mu[6] = {0.0f, 0.0f, 19.5f, -1e-4f, -1e-4f, -5e-6f};
    for (int i = 0; i < 6; i++) { c.mu[i] = mu[i]; c.sigma[i] = sigma[i]; }
    c.ca_K = 10.0f;
    c.n_save = NSAVE;
    c.substeps = 2;
    c.save_times = times;
    c.n_columns = NCOL;
    c.device = 0;
    c.engine = argc > 2 ? atoi(argv[2]) : COLNDE_ENGINE_AUTO;
    c.stepper = COLNDE_STEPPER_RK4;

    if (colnde_version() != COLNDE_VERSION) { fprintf(stderr, "header %d != library %d\n", COLNDE_VERSION, colnde_version()); return 1; }
    if (colnde_min_substeps(&c) != 2) { fprintf(stderr, "min_substeps = %d, expected 2\n", colnde_min_substeps(&c)); return 1; }
    colnde_handle* h = NULL;
    CHECK(colnde_create(&c, &h));
    const int np = colnde_n_params(h);
    if (np != 3 * 6521) { fprintf(stderr, "n_params = %d\n", np); return 1; }

    float* x0 = slurp(dir, "x0.f32", (size_t)NCOL * NS);
    float* bcs = slurp(dir, "bcs.f32", (size_t)NCOL * 6);
    float* w = slurp(dir, "weights.f32", (size_t)np);
    float* truth = slurp(dir, "truth.f32", (size_t)NCOL * NSAVE * NS);
    float* scal = slurp(dir, "scalings.f32", 6);
    float* sol_ref = slurp(dir, "sol.f32", (size_t)NCOL * NSAVE * NS);
    float* grad_ref = slurp(dir, "grad.f32", (size_t)np);
    float* total_ref = slurp(dir, "total.f32", 1);

    CHECK(colnde_set_problem(h, x0, bcs, truth));
    float* sol = (float*)malloc(sizeof(float) * NCOL * NSAVE * NS);
    CHECK(colnde_forward(h, w, sol));
    double esol = 0.0;
    for (size_t i = 0; i < (size_t)NCOL * NSAVE * NS; i++) esol = fmax(esol, fabs((double)sol[i] - sol_ref[i]));

    float terms[6], total = 0.0f;
    float* grad = (float*)malloc(sizeof(float) * np);
    CHECK(colnde_loss_grad(h, w, scal, terms, &total, grad));
    double num = 0.0, den = 0.0, tsum = 0.0;
    for (int i = 0; i < np; i++) { const double d = (double)grad[i] - grad_ref[i]; num += d * d; den += (double)grad_ref[i] * grad_ref[i]; }
    for (int q = 0; q < 6; q++) tsum += terms[q];
    const double egrad = sqrt(num / den), etot = fabs((double)total - total_ref[0]) / total_ref[0];

    /* round-4 entry points, from plain C: loss_per_tstep averages to loss_NDE's terms (NDE_training.jl:308-317); predict_flux's faces difference to the
     * tendencies of colnde_rhs for T (dT/dt = -(tau/H)(sigma_wT/sigma_T) Dc wT, NDE_training.jl:162); the Richardson estimate is finite and small; the handle describes itself */
    double eterms = 0.0, eflux = 0.0;
    float est = -1.0f;
    char desc[1024];
    {
        float* lpt = (float*)malloc(sizeof(float) * NCOL * 6 * NSAVE);
        CHECK(colnde_loss_per_tstep(h, w, lpt));
        for (int q = 0; q < 6; q++) {
            double m = 0.0;
            for (int col = 0; col < NCOL; col++)
                for (int s = 0; s < NSAVE; s++) m += lpt[((size_t)col * 6 + q) * NSAVE + s];
            m = m / (NCOL * NSAVE) * scal[q];
            if (terms[q] != 0.0f) eterms = fmax(eterms, fabs(m - terms[q]) / fabs(terms[q]));
        }
        float* flux = (float*)malloc(sizeof(float) * NCOL * 3 * (NZ + 1));
        float* dx = (float*)malloc(sizeof(float) * NCOL * NS);
        CHECK(colnde_flux(h, x0, w, bcs, 0.0f, flux, NCOL));
        CHECK(colnde_rhs(h, x0, w, bcs, 0.0f, dx, NCOL));
        const double cT = (double)c.tau / c.H * c.sigma[5] / c.sigma[2];
        for (int col = 0; col < NCOL; col++)
            for (int k = 0; k < NZ; k++) {
                const float* wT = flux + ((size_t)col * 3 + 2) * (NZ + 1);
                const double want = -cT * ((double)wT[k + 1] - wT[k]) * NZ, got = dx[(size_t)col * NS + 2 * NZ + k];
                eflux = fmax(eflux, fabs(want - got) / (1e-3 + fabs(got)));
            }
        CHECK(colnde_error_estimate(h, w, &est));
        if (colnde_describe(h, desc, (int)sizeof desc) <= 0) { fprintf(stderr, "colnde_describe failed: %s\n", colnde_last_error()); return 1; }
        free(lpt); free(flux); free(dx);
    }
    printf("abi_smoke: loss_per_tstep vs terms %.2e, Dc(flux) vs rhs %.2e, error estimate %.2e, substeps %d\n  %s\n", eterms, eflux, est, colnde_substeps(h), desc);
    if (!(eterms < 5e-4) || !(eflux < 1e-4) || !(est > 0.0f && est < 1e-2f) || strstr(desc, "engine=") == NULL) return 1;

    /* a config the library must refuse, with a message */
    colnde_config bad = c;
    bad.layer_sizes[3] = 30;
    colnde_handle* hb = NULL;
    const int rc_bad = colnde_create(&bad, &hb);
    printf("abi_smoke: engine %d, max|sol - golden| = %.3e, rel loss err = %.3e, rel grad err = %.3e, sum(terms)/total = %.7f, bad config rc = %d (%s)\n",
           colnde_engine(h), esol, etot, egrad, tsum / total, rc_bad, rc_bad ? colnde_last_error() : "accepted?!");
    /* the implicit diffusion step (modified_pacanowski_philander!, NDE_oceananigans.jl:61-101) on a known answer: constant diffusivity
     * (Ric -> +inf: tanh_step = 1), one backward-Euler step damps the discrete cosine mode m by 1/(1 + c nu lambda_m) */
    double ediff = 0.0;
    {
        enum { NC = 3 };
        float u[NC * NZ], v[NC * NZ], T[NC * NZ];
        const float prm[7] = {0.0f, 2e-2f, 1.0f, 1e30f, 1.0f, 1.67e-4f, 9.81f};
        const double PI = 3.14159265358979323846, dt = 60.0, dz = 8.0, lam = 4.0 * pow(sin(PI * 3 / (2.0 * NZ)), 2);
        for (int cidx = 0; cidx < NC; cidx++)
            for (int k = 0; k < NZ; k++) {
                u[cidx * NZ + k] = (float)(0.05 * (cidx + 1) * cos(PI * 3 * (k + 0.5) / NZ));
                v[cidx * NZ + k] = 0.02f;
                T[cidx * NZ + k] = 19.6f + 0.01f * k;
            }
        float u0[NC * NZ];
        memcpy(u0, u, sizeof u);
        CHECK(colnde_implicit_diffusion(h, u, v, T, NULL, (float)dt, (float)dz, prm, 0, u, v, T, NC));
        for (int i = 0; i < NC * NZ; i++) ediff = fmax(ediff, fabs((double)u[i] - u0[i] / (1.0 + dt / (dz * dz) * 2e-2 * lam)));
        for (int i = 0; i < NC * NZ; i++) ediff = fmax(ediff, fabs((double)v[i] - 0.02));
    }
    printf("abi_smoke: implicit diffusion, max error against the damped cosine mode = %.3e\n", ediff);
    if (!(ediff < 5e-7)) return 1;
    colnde_destroy(h);
    /* tolerances of tests/test_gpu_parity.py (wind mixing, short horizon); the golden file stores float32 roundings of the float64 oracle */
    if (!(esol < 2e-5) || !(etot < 8e-5) || !(egrad < 2e-4) || fabs(tsum / total - 1.0) > 1e-5 || rc_bad == 0 || hb != NULL) return 1;
    printf("abi_smoke: OK\n");
    return 0;
}
