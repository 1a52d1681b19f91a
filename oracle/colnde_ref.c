/* CPU ORACLE (C, float32) — TEST INFRASTRUCTURE ONLY.  **parity unpinned.**
 *
 * A scalar float32 restatement of the reference's NDE column-model hot path, one column at a time,
 * OpenMP over columns.  It mirrors oracle/nde_oracle.py (which the tests pin with analytic cases, a
 * literal torch restatement and finite differences) and is what bench.py times as the `cpu_baseline`
 * ("port").  Nothing in the product path links or loads this file.
 *
 * "parity unpinned": the Julia reference cannot run in the build container and its tests hold no golden
 * vectors for this path.
 *
 * Reference lines restated (relative to /root/reference):
 *   RHS, wind mixing     wind_mixing/src/NDE_training.jl:46-165  (in-place twin training_postprocessing.jl:105-153)
 *   RHS, free convection free_convection/src/free_convection_nde.jl:29-38, convective_adjustment_nde.jl:33-48
 *   operators            src/differentiation_operators.jl:6-29, wind_mixing/src/filtering_operators.jl:1-14
 *   losses               wind_mixing/src/loss.jl:1-42, wind_mixing/src/NDE_training.jl:290-323, free_convection/src/training.jl:55-62
 *   inference            free_convection/double_gyre_nn.jl:149-168
 * Index convention: 0-based; cells k = 0..Nz-1 (0 deepest), faces f = 0..Nz.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/colnde.h"

#define MAXNZ 128

typedef struct {
    const colnde_config* c;
    int Nz, ns, n_nets, n_bc, net_size, n_params, act_total;
    float cs[3], A[3], s0[3], B, cor_u, cor_v, C_fc;
} model_t;

static void model_init(model_t* m, const colnde_config* c) {
    m->c = c;
    m->Nz = c->Nz;
    int wm = c->model == COLNDE_MODEL_WIND_MIXING;
    m->n_nets = wm ? 3 : 1;
    m->ns = wm ? 3 * c->Nz : c->Nz;
    m->n_bc = wm ? 6 : 2;
    int sz = 0, at = 0;
    for (int l = 0; l < c->n_layers; l++) {
        sz += c->layer_sizes[l] * c->layer_sizes[l + 1] + c->layer_sizes[l + 1];
        at += c->layer_sizes[l + 1];
    }
    m->net_size = sz;
    m->n_params = sz * m->n_nets;
    m->act_total = at;
    const float* sg = c->sigma;
    const float* mu = c->mu;
    m->cs[0] = sg[0] / sg[3] / c->H;
    m->cs[1] = sg[1] / sg[4] / c->H;
    m->cs[2] = sg[2] / sg[5] / c->H;
    m->A[0] = c->tau / c->H * sg[3] / sg[0] * (float)c->Nz;
    m->A[1] = c->tau / c->H * sg[4] / sg[1] * (float)c->Nz;
    m->A[2] = c->tau / c->H * sg[5] / sg[2] * (float)c->Nz;
    for (int k = 0; k < 3; k++) m->s0[k] = -mu[3 + k] / sg[3 + k];
    m->B = c->H * c->g * c->alpha * sg[2];
    m->cor_u = c->f * c->tau / sg[0];
    m->cor_v = c->f * c->tau / sg[1];
    m->C_fc = (sg[5] / sg[2]) * (c->tau / c->H);
}

/* ---- activations (NNlib 0.7) ---- */
static inline float softplusf(float x) { return x > 0 ? x + log1pf(expf(-x)) : log1pf(expf(x)); }
static inline float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }
static inline float actf(int a, float z) {
    switch (a) {
        case COLNDE_ACT_RELU: return z > 0 ? z : 0.0f;
        case COLNDE_ACT_MISH: return z * tanhf(softplusf(z));
        case COLNDE_ACT_SWISH: return z * sigmoidf(z);
        case COLNDE_ACT_TANH: return tanhf(z);
        case COLNDE_ACT_LEAKYRELU: return z > 0 ? z : 0.01f * z;
        default: return z;
    }
}
static inline float actgf(int a, float z) {
    switch (a) {
        case COLNDE_ACT_RELU: return z > 0 ? 1.0f : 0.0f;
        case COLNDE_ACT_MISH: { float t = tanhf(softplusf(z)); return t + z * (1.0f - t * t) * sigmoidf(z); }
        case COLNDE_ACT_SWISH: { float s = sigmoidf(z); return s + z * s * (1.0f - s); }
        case COLNDE_ACT_TANH: { float t = tanhf(z); return 1.0f - t * t; }
        case COLNDE_ACT_LEAKYRELU: return z > 0 ? 1.0f : 0.01f;
        default: return 1.0f;
    }
}

/* 3-point smoothing filter rows (filtering_operators.jl:1-14): y = F x and y = F^T x */
static void smooth3(const float* x, float* y, int N) {
    y[0] = 0.5f * (x[0] + x[1]);
    y[N - 1] = 0.5f * (x[N - 2] + x[N - 1]);
    for (int i = 1; i < N - 1; i++) y[i] = (x[i - 1] + x[i] + x[i + 1]) * (1.0f / 3.0f);
}
static void smooth3_T(const float* x, float* y, int N) {
    for (int j = 0; j < N; j++) y[j] = 0.0f;
    y[0] += 0.5f * x[0]; y[1] += 0.5f * x[0];
    y[N - 2] += 0.5f * x[N - 1]; y[N - 1] += 0.5f * x[N - 1];
    for (int i = 1; i < N - 1; i++) { float v = x[i] * (1.0f / 3.0f); y[i - 1] += v; y[i] += v; y[i + 1] += v; }
}

/* per-RHS tape of one column */
typedef struct {
    float* x;                 /* [ns] */
    float* z;                 /* [n_nets][act_total] pre-activations */
    float* a;                 /* [n_nets][act_total] activations */
    float gu[MAXNZ + 1], gv[MAXNZ + 1], gT[MAXNZ + 1], S2[MAXNZ + 1], Ri[MAXNZ + 1], th[MAXNZ + 1], nu[MAXNZ + 1];
} tape_t;

/* Chain(Dense...)(x): theta in Flux.destructure order, W column-major out x in */
static void mlp_fwd(const model_t* m, const float* th, const float* x, float* z, float* a) {
    const colnde_config* c = m->c;
    const float* in = x;
    int off = 0;
    for (int l = 0; l < c->n_layers; l++) {
        int ni = c->layer_sizes[l], no = c->layer_sizes[l + 1];
        const float* W = th;
        const float* b = th + ni * no;
        for (int j = 0; j < no; j++) z[off + j] = b[j];
        for (int i = 0; i < ni; i++) {
            float xi = in[i];
            const float* Wc = W + (size_t)i * no;
            for (int j = 0; j < no; j++) z[off + j] += Wc[j] * xi;
        }
        for (int j = 0; j < no; j++) a[off + j] = actf(c->activations[l], z[off + j]);
        in = a + off;
        off += no;
        th += ni * no + no;
    }
}

/* pullback: ybar [out] -> xbar += ..., g += ... (g in destructure order; NULL to skip) */
static void mlp_vjp(const model_t* m, const float* th, const float* x, const float* z, const float* a,
                    const float* ybar, float* xbar, float* g) {
    const colnde_config* c = m->c;
    int L = c->n_layers;
    int offs[COLNDE_MAX_LAYERS + 1], poff[COLNDE_MAX_LAYERS + 1];
    offs[0] = 0; poff[0] = 0;
    for (int l = 0; l < L; l++) {
        offs[l + 1] = offs[l] + c->layer_sizes[l + 1];
        poff[l + 1] = poff[l] + c->layer_sizes[l] * c->layer_sizes[l + 1] + c->layer_sizes[l + 1];
    }
    float cur[1024], nxt[1024];
    int no = c->layer_sizes[L];
    for (int j = 0; j < no; j++) cur[j] = ybar[j];
    for (int l = L - 1; l >= 0; l--) {
        int ni = c->layer_sizes[l];
        no = c->layer_sizes[l + 1];
        const float* W = th + poff[l];
        const float* ain = l == 0 ? x : a + offs[l - 1];
        for (int j = 0; j < no; j++) cur[j] *= actgf(c->activations[l], z[offs[l] + j]);
        if (g) {
            float* gW = g + poff[l];
            float* gb = gW + ni * no;
            for (int i = 0; i < ni; i++) {
                float ai = ain[i];
                for (int j = 0; j < no; j++) gW[(size_t)i * no + j] += cur[j] * ai;
            }
            for (int j = 0; j < no; j++) gb[j] += cur[j];
        }
        for (int i = 0; i < ni; i++) {
            const float* Wc = W + (size_t)i * no;
            float s = 0.0f;
            for (int j = 0; j < no; j++) s += Wc[j] * cur[j];
            nxt[i] = s;
        }
        if (l == 0) { for (int i = 0; i < ni; i++) xbar[i] += nxt[i]; }
        else memcpy(cur, nxt, sizeof(float) * ni);
    }
}

static inline float wm_top(const model_t* m, const float* bc, float t) {
    const colnde_config* c = m->c;
    if (!c->diurnal) return bc[5];
    /* scalings.wT(Q sin(2π/86400 · tτ)/(αg)) — NDE_training.jl:73, data_containers.jl:135 */
    float w = bc[5] * sinf(6.283185307179586f / 86400.0f * (t * c->tau)) / (c->alpha * c->g);
    return (w - c->mu[5]) / c->sigma[5];
}

/* wind mixing RHS (NDE_training.jl:56-165) */
static void wm_rhs(const model_t* m, const float* theta, const float* x, const float* bc, float t, float* dx, tape_t* tp) {
    const colnde_config* c = m->c;
    int Nz = m->Nz, nout = Nz - 1, at = m->act_total;
    const float* u = x; const float* v = x + Nz; const float* T = x + 2 * Nz;
    float F[3][MAXNZ + 1];
    float tmp[MAXNZ];
    float bcb[3] = { bc[0], bc[2], bc[4] };
    float bct[3] = { bc[1], bc[3], wm_top(m, bc, t) };
    for (int k = 0; k < 3; k++) {
        mlp_fwd(m, theta + (size_t)k * m->net_size, x, tp->z + k * at, tp->a + k * at);
        const float* o = tp->a + k * at + (at - nout);
        if (c->smooth_NN) { smooth3(o, tmp, nout); o = tmp; }
        for (int f = 1; f < Nz; f++) F[k][f] = o[f - 1];
        F[k][0] = c->zero_weights ? 0.0f : bcb[k];
        F[k][Nz] = c->zero_weights ? 0.0f : bct[k];
    }
    if (c->modified_pacanowski_philander) {
        float eps = c->inplace_variant ? 0.0f : c->eps;
        float Ri[MAXNZ + 1];
        for (int f = 0; f <= Nz; f++) {
            int in = f >= 1 && f < Nz;
            tp->gu[f] = in ? (u[f] - u[f - 1]) * (float)Nz : 0.0f;
            tp->gv[f] = in ? (v[f] - v[f - 1]) * (float)Nz : 0.0f;
            tp->gT[f] = in ? (T[f] - T[f - 1]) * (float)Nz : 0.0f;
            float a1 = c->sigma[0] * (tp->gu[f] + eps), a2 = c->sigma[1] * (tp->gv[f] + eps);
            tp->S2[f] = a1 * a1 + a2 * a2;
            tp->Ri[f] = m->B * (tp->gT[f] + eps) / tp->S2[f];
        }
        if (c->smooth_Ri) smooth3(tp->Ri, Ri, Nz + 1); else memcpy(Ri, tp->Ri, sizeof(float) * (Nz + 1));
        for (int f = 1; f < Nz; f++) {
            float th = tanhf((Ri[f] - c->Ric) / c->dRi);
            tp->th[f] = th;
            float nu = c->nu0 + c->nu_minus * (1.0f - th) * 0.5f;
            tp->nu[f] = nu;
            float nuT = nu / c->Pr;
            if (c->inplace_variant && c->convective_adjustment) nuT = tp->gu[f] > 0 ? nu / c->Pr : c->kappa;
            F[0][f] -= m->cs[0] * nu * tp->gu[f];
            F[1][f] -= m->cs[1] * nu * tp->gv[f];
            F[2][f] -= m->cs[2] * nuT * tp->gT[f];
        }
        if (c->zero_weights) {
            for (int k = 0; k < 3; k++) {
                F[k][0] += bcb[k] - m->s0[k];
                F[k][Nz] += (c->inplace_variant && c->diurnal && k == 2) ? bct[k] : bct[k] - m->s0[k];
            }
        }
    } else if (c->convective_adjustment) {
        for (int f = 0; f <= Nz; f++) {
            int in = f >= 1 && f < Nz;
            tp->gT[f] = in ? (T[f] - T[f - 1]) * (float)Nz : 0.0f;
            F[2][f] -= m->cs[2] * c->kappa * fminf(0.0f, tp->gT[f]);
        }
    }
    for (int k = 0; k < Nz; k++) {
        dx[k] = -m->A[0] * (F[0][k + 1] - F[0][k]) + m->cor_u * (c->sigma[1] * v[k] + c->mu[1]);
        dx[Nz + k] = -m->A[1] * (F[1][k + 1] - F[1][k]) - m->cor_v * (c->sigma[0] * u[k] + c->mu[0]);
        dx[2 * Nz + k] = -m->A[2] * (F[2][k + 1] - F[2][k]);
    }
    memcpy(tp->x, x, sizeof(float) * m->ns);
}

static void wm_vjp(const model_t* m, const float* theta, const tape_t* tp, const float* dbar, float* xbar, float* g) {
    const colnde_config* c = m->c;
    int Nz = m->Nz, nout = Nz - 1, at = m->act_total;
    const float* dub = dbar; const float* dvb = dbar + Nz; const float* dTb = dbar + 2 * Nz;
    float Fb[3][MAXNZ + 1];
    for (int k = 0; k < 3; k++) {
        const float* db = dbar + k * Nz;
        for (int f = 0; f <= Nz; f++) {
            float s = 0.0f;
            if (f >= 1) s += -m->A[k] * db[f - 1];
            if (f < Nz) s += m->A[k] * db[f];
            Fb[k][f] = s;
        }
    }
    for (int k = 0; k < Nz; k++) {
        xbar[k] = -m->cor_v * c->sigma[0] * dvb[k];
        xbar[Nz + k] = m->cor_u * c->sigma[1] * dub[k];
        xbar[2 * Nz + k] = 0.0f;
    }
    (void)dTb;
    float gb[3][MAXNZ + 1];
    memset(gb, 0, sizeof(gb));
    if (c->modified_pacanowski_philander) {
        float eps = c->inplace_variant ? 0.0f : c->eps;
        float Ribs[MAXNZ + 1], Rib[MAXNZ + 1];
        memset(Ribs, 0, sizeof(Ribs));
        for (int f = 1; f < Nz; f++) {
            float nu = tp->nu[f], th = tp->th[f];
            float D0 = -Fb[0][f], D1 = -Fb[1][f], D2 = -Fb[2][f];
            gb[0][f] = D0 * m->cs[0] * nu;
            gb[1][f] = D1 * m->cs[1] * nu;
            gb[2][f] = D2 * m->cs[2] * nu / c->Pr;
            float nub = D0 * m->cs[0] * tp->gu[f] + D1 * m->cs[1] * tp->gv[f] + D2 * m->cs[2] * tp->gT[f] / c->Pr;
            Ribs[f] = nub * (-c->nu_minus / (2.0f * c->dRi)) * (1.0f - th * th);
        }
        if (c->smooth_Ri) smooth3_T(Ribs, Rib, Nz + 1); else memcpy(Rib, Ribs, sizeof(float) * (Nz + 1));
        for (int f = 1; f < Nz; f++) {
            float S2 = tp->S2[f];
            gb[2][f] += Rib[f] * m->B / S2;
            float q = Rib[f] * (-tp->Ri[f] / S2) * 2.0f;
            gb[0][f] += q * c->sigma[0] * c->sigma[0] * (tp->gu[f] + eps);
            gb[1][f] += q * c->sigma[1] * c->sigma[1] * (tp->gv[f] + eps);
        }
    } else if (c->convective_adjustment) {
        for (int f = 1; f < Nz; f++) gb[2][f] = tp->gT[f] < 0 ? -Fb[2][f] * m->cs[2] * c->kappa : 0.0f;
    }
    for (int k = 0; k < 3; k++)
        for (int f = 1; f < Nz; f++) {
            xbar[k * Nz + f] += gb[k][f] * (float)Nz;
            xbar[k * Nz + f - 1] -= gb[k][f] * (float)Nz;
        }
    float tmp[MAXNZ];
    for (int k = 0; k < 3; k++) {
        const float* ob = &Fb[k][1];
        if (c->smooth_NN) { smooth3_T(ob, tmp, nout); ob = tmp; }
        mlp_vjp(m, theta + (size_t)k * m->net_size, tp->x, tp->z + k * at, tp->a + k * at, ob, xbar,
                g ? g + (size_t)k * m->net_size : NULL);
    }
}

/* free convection RHS and the convective-adjustment NDE */
static void fc_rhs(const model_t* m, const float* theta, const float* x, const float* bc, float t, float* dx, tape_t* tp) {
    (void)t;
    const colnde_config* c = m->c;
    int Nz = m->Nz, nout = Nz - 1, at = m->act_total;
    float w[MAXNZ + 1], q[MAXNZ + 1];
    mlp_fwd(m, theta, x, tp->z, tp->a);
    const float* o = tp->a + (at - nout);
    w[0] = bc[0]; w[Nz] = bc[1];
    for (int f = 1; f < Nz; f++) w[f] = o[f - 1];
    int ca = c->model == COLNDE_MODEL_CONV_ADJ_NDE;
    for (int f = 0; f <= Nz; f++) {
        int in = f >= 1 && f < Nz;
        tp->gT[f] = (ca && in) ? (x[f] - x[f - 1]) * (float)Nz : 0.0f;
        q[f] = ca ? fminf(0.0f, c->ca_K * tp->gT[f]) : 0.0f;
    }
    float CN = m->C_fc * (float)Nz;
    for (int k = 0; k < Nz; k++) dx[k] = -CN * (w[k + 1] - w[k]) + CN * (q[k + 1] - q[k]);
    memcpy(tp->x, x, sizeof(float) * m->ns);
}

static void fc_vjp(const model_t* m, const float* theta, const tape_t* tp, const float* dbar, float* xbar, float* g) {
    const colnde_config* c = m->c;
    int Nz = m->Nz;
    float wb[MAXNZ + 1];
    float CN = m->C_fc * (float)Nz;
    for (int f = 0; f <= Nz; f++) {
        float s = 0.0f;
        if (f >= 1) s += -CN * dbar[f - 1];
        if (f < Nz) s += CN * dbar[f];
        wb[f] = s;
    }
    for (int k = 0; k < Nz; k++) xbar[k] = 0.0f;
    if (c->model == COLNDE_MODEL_CONV_ADJ_NDE)
        for (int f = 1; f < Nz; f++) {
            float gTb = tp->gT[f] < 0 ? -wb[f] * c->ca_K : 0.0f;
            xbar[f] += gTb * (float)Nz;
            xbar[f - 1] -= gTb * (float)Nz;
        }
    mlp_vjp(m, theta, tp->x, tp->z, tp->a, &wb[1], xbar, g);
}

static void rhs_any(const model_t* m, const float* theta, const float* x, const float* bc, float t, float* dx, tape_t* tp) {
    if (m->c->model == COLNDE_MODEL_WIND_MIXING) wm_rhs(m, theta, x, bc, t, dx, tp);
    else fc_rhs(m, theta, x, bc, t, dx, tp);
}
static void vjp_any(const model_t* m, const float* theta, const tape_t* tp, const float* dbar, float* xbar, float* g) {
    if (m->c->model == COLNDE_MODEL_WIND_MIXING) wm_vjp(m, theta, tp, dbar, xbar, g);
    else fc_vjp(m, theta, tp, dbar, xbar, g);
}

static tape_t* tape_new(const model_t* m) {
    tape_t* tp = (tape_t*)calloc(1, sizeof(tape_t));
    tp->x = (float*)calloc(m->ns, sizeof(float));
    tp->z = (float*)calloc((size_t)m->n_nets * m->act_total, sizeof(float));
    tp->a = (float*)calloc((size_t)m->n_nets * m->act_total, sizeof(float));
    return tp;
}
static void tape_free(tape_t* tp) { free(tp->x); free(tp->z); free(tp->a); free(tp); }

/* one classical RK4 step; optionally keeps the four stage tapes */
static void rk4_step(const model_t* m, const float* theta, const float* bc, float t, float dt, const float* x,
                     float* xn, tape_t** tp, float* k /* [4][ns] */, float* xs /* [ns] */) {
    int ns = m->ns;
    const float ca[4] = { 0.0f, 0.5f, 0.5f, 1.0f };
    for (int s = 0; s < 4; s++) {
        if (s == 0) memcpy(xs, x, sizeof(float) * ns);
        else for (int i = 0; i < ns; i++) xs[i] = x[i] + ca[s] * dt * k[(s - 1) * ns + i];
        rhs_any(m, theta, xs, bc, t + ca[s] * dt, k + s * ns, tp[s]);
    }
    for (int i = 0; i < ns; i++)
        xn[i] = x[i] + dt / 6.0f * (k[i] + 2.0f * k[ns + i] + 2.0f * k[2 * ns + i] + k[3 * ns + i]);
}

int colnde_ref_n_params(const colnde_config* c) { model_t m; model_init(&m, c); return m.n_params; }

int colnde_ref_rhs(const colnde_config* c, const float* x, const float* bcs, const float* theta, float t, float* dx, int n_col) {
    model_t m; model_init(&m, c);
    if (c->Nz > MAXNZ) return 1;
    tape_t* tp = tape_new(&m);
    for (int col = 0; col < n_col; col++)
        rhs_any(&m, theta, x + (size_t)col * m.ns, bcs + (size_t)col * m.n_bc, t, dx + (size_t)col * m.ns, tp);
    tape_free(tp);
    return 0;
}

/* sol [n_col][n_save][ns] */
int colnde_ref_forward(const colnde_config* c, const float* x0, const float* bcs, const float* theta, float* sol, int n_threads) {
    model_t m; model_init(&m, c);
    if (c->Nz > MAXNZ) return 1;
    int ns = m.ns, n_col = c->n_columns;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel
    {
        tape_t* tp[4];
        for (int s = 0; s < 4; s++) tp[s] = tape_new(&m);
        float* k = (float*)malloc(sizeof(float) * 4 * ns);
        float* xs = (float*)malloc(sizeof(float) * ns);
        float* x = (float*)malloc(sizeof(float) * ns);
        float* xn = (float*)malloc(sizeof(float) * ns);
#pragma omp for schedule(static)
        for (int col = 0; col < n_col; col++) {
            const float* bc = bcs + (size_t)col * m.n_bc;
            float* so = sol + (size_t)col * c->n_save * ns;
            memcpy(x, x0 + (size_t)col * ns, sizeof(float) * ns);
            memcpy(so, x, sizeof(float) * ns);
            for (int iv = 0; iv < c->n_save - 1; iv++) {
                float t0 = c->save_times[iv];
                float dt = (c->save_times[iv + 1] - t0) / (float)c->substeps;
                for (int s = 0; s < c->substeps; s++) {
                    rk4_step(&m, theta, bc, t0 + (float)s * dt, dt, x, xn, tp, k, xs);
                    memcpy(x, xn, sizeof(float) * ns);
                }
                memcpy(so + (size_t)(iv + 1) * ns, x, sizeof(float) * ns);
            }
        }
        for (int s = 0; s < 4; s++) tape_free(tp[s]);
        free(k); free(xs); free(x); free(xn);
    }
    return 0;
}

/* loss injection at one save point; also accumulates the six raw sums of squares (double) */
static void inject(const model_t* m, const float* sol_n, const float* truth_n, const float* w6 /* scaling/(N) per term */,
                   float* lam, double* sums) {
    const colnde_config* c = m->c;
    int Nz = m->Nz;
    if (c->model != COLNDE_MODEL_WIND_MIXING) {
        for (int i = 0; i < Nz; i++) {
            float d = sol_n[i] - truth_n[i];
            sums[2] += (double)d * d;
            lam[i] += 2.0f * w6[2] * d;
        }
        return;
    }
    for (int k = 0; k < 3; k++) {
        const float* s = sol_n + k * Nz; const float* y = truth_n + k * Nz;
        float gprev = 0.0f;
        for (int i = 0; i < Nz; i++) {
            float d = s[i] - y[i];
            sums[k] += (double)d * d;
            lam[k * Nz + i] += 2.0f * w6[k] * d;
        }
        (void)gprev;
        for (int f = 1; f < Nz; f++) {
            float gd = ((s[f] - y[f]) - (s[f - 1] - y[f - 1])) * (float)Nz;
            sums[3 + k] += (double)gd * gd;
            float q = 2.0f * w6[3 + k] * gd * (float)Nz;
            lam[k * Nz + f] += q;
            lam[k * Nz + f - 1] -= q;
        }
    }
}

/* Forward + discrete adjoint.  terms[6] = scaled loss terms, *total = their sum, grad [n_params] (nullable),
 * sol [n_col][n_save][ns] (nullable).  Losses/gradients are normalised with n_col_total (<=0: n_columns). */
int colnde_ref_loss_grad(const colnde_config* c, const float* x0, const float* bcs, const float* theta,
                         const float* truth, const float* scalings, long n_col_total,
                         float* terms, float* total, float* grad, float* sol, int n_threads) {
    model_t m; model_init(&m, c);
    if (c->Nz > MAXNZ) return 1;
    int ns = m.ns, n_col = c->n_columns, Nz = m.Nz;
    int n_steps = (c->n_save - 1) * c->substeps;
    if (n_col_total <= 0) n_col_total = n_col;
    double cnt_p = (double)n_col_total * c->n_save * Nz, cnt_g = (double)n_col_total * c->n_save * (Nz + 1);
    float w6[6];
    for (int k = 0; k < 3; k++) { w6[k] = (float)(scalings[k] / cnt_p); w6[3 + k] = (float)(scalings[3 + k] / cnt_g); }
    double sums_all[6] = { 0, 0, 0, 0, 0, 0 };
    if (grad) memset(grad, 0, sizeof(float) * m.n_params);
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel
    {
        tape_t* tp[4];
        for (int s = 0; s < 4; s++) tp[s] = tape_new(&m);
        float* k = (float*)malloc(sizeof(float) * 4 * ns);
        float* xs = (float*)malloc(sizeof(float) * ns);
        float* xn = (float*)malloc(sizeof(float) * ns);
        float* steps = (float*)malloc(sizeof(float) * (size_t)(n_steps + 1) * ns);   /* x at every step start */
        float* lam = (float*)malloc(sizeof(float) * ns);
        float* kb = (float*)malloc(sizeof(float) * ns);
        float* xb = (float*)malloc(sizeof(float) * 4 * ns);
        float* g = grad ? (float*)calloc(m.n_params, sizeof(float)) : NULL;
        double sums[6] = { 0, 0, 0, 0, 0, 0 };
#pragma omp for schedule(static)
        for (int col = 0; col < n_col; col++) {
            const float* bc = bcs + (size_t)col * m.n_bc;
            const float* tr = truth + (size_t)col * c->n_save * ns;
            memcpy(steps, x0 + (size_t)col * ns, sizeof(float) * ns);
            int si = 0;
            for (int iv = 0; iv < c->n_save - 1; iv++) {
                float t0 = c->save_times[iv];
                float dt = (c->save_times[iv + 1] - t0) / (float)c->substeps;
                for (int s = 0; s < c->substeps; s++, si++)
                    rk4_step(&m, theta, bc, t0 + (float)s * dt, dt, steps + (size_t)si * ns, steps + (size_t)(si + 1) * ns, tp, k, xs);
            }
            if (sol)
                for (int n = 0; n < c->n_save; n++)
                    memcpy(sol + ((size_t)col * c->n_save + n) * ns, steps + (size_t)n * c->substeps * ns, sizeof(float) * ns);
            memset(lam, 0, sizeof(float) * ns);
            /* save point 0 contributes to the loss value only (x0 does not depend on the weights) */
            {
                float dummy[3 * MAXNZ];
                memset(dummy, 0, sizeof(dummy));
                inject(&m, steps, tr, w6, dummy, sums);
            }
            if (!grad) {
                for (int n = 1; n < c->n_save; n++) {
                    float dummy[3 * MAXNZ];
                    memset(dummy, 0, sizeof(dummy));
                    inject(&m, steps + (size_t)n * c->substeps * ns, tr + (size_t)n * ns, w6, dummy, sums);
                }
                continue;
            }
            for (int iv = c->n_save - 2; iv >= 0; iv--) {
                float t0 = c->save_times[iv];
                float dt = (c->save_times[iv + 1] - t0) / (float)c->substeps;
                inject(&m, steps + (size_t)(iv + 1) * c->substeps * ns, tr + (size_t)(iv + 1) * ns, w6, lam, sums);
                for (int s = c->substeps - 1; s >= 0; s--) {
                    si = iv * c->substeps + s;
                    rk4_step(&m, theta, bc, t0 + (float)s * dt, dt, steps + (size_t)si * ns, xn, tp, k, xs);
                    /* k4b = dt/6 λ; k3b = dt/3 λ + dt x4b; k2b = dt/3 λ + dt/2 x3b; k1b = dt/6 λ + dt/2 x2b */
                    const float wl[4] = { dt / 6.0f, dt / 3.0f, dt / 3.0f, dt / 6.0f };
                    const float wx[4] = { 0.5f * dt, 0.5f * dt, dt, 0.0f };
                    for (int st = 3; st >= 0; st--) {
                        for (int i = 0; i < ns; i++)
                            kb[i] = wl[st] * lam[i] + (st < 3 ? wx[st] * xb[(st + 1) * ns + i] : 0.0f);
                        vjp_any(&m, theta, tp[st], kb, xb + st * ns, g);
                    }
                    for (int i = 0; i < ns; i++) lam[i] += xb[i] + xb[ns + i] + xb[2 * ns + i] + xb[3 * ns + i];
                }
            }
        }
#pragma omp critical
        {
            for (int q = 0; q < 6; q++) sums_all[q] += sums[q];
            if (grad) for (int i = 0; i < m.n_params; i++) grad[i] += g[i];
        }
        for (int s = 0; s < 4; s++) tape_free(tp[s]);
        free(k); free(xs); free(xn); free(steps); free(lam); free(kb); free(xb); free(g);
    }
    double tot = 0;
    for (int q = 0; q < 6; q++) {
        double cnt = q < 3 ? cnt_p : cnt_g;
        double v = (c->model == COLNDE_MODEL_WIND_MIXING || q == 2) ? scalings[q] * sums_all[q] / cnt : 0.0;
        terms[q] = (float)v;
        tot += v;
    }
    *total = (float)tot;
    return 0;
}

/* double_gyre_nn.jl:149-168 */
int colnde_ref_infer_forcing(const colnde_config* c, const float* theta, const float* T, const float* top_flux,
                             float Lz, float* out, int n_col) {
    model_t m; model_init(&m, c);
    int Nz = m.Nz, nout = Nz - 1;
    tape_t* tp = tape_new(&m);
    float xs[MAXNZ], wT[MAXNZ + 1];
    float dz = Lz / (float)Nz;
    for (int col = 0; col < n_col; col++) {
        for (int k = 0; k < Nz; k++) xs[k] = ((19.65f + T[(size_t)col * Nz + k] / 20.0f) - c->mu[2]) / c->sigma[2];
        mlp_fwd(&m, theta, xs, tp->z, tp->a);
        const float* o = tp->a + (m.act_total - nout);
        wT[0] = 0.0f; wT[Nz] = top_flux[col];
        for (int f = 1; f < Nz; f++) wT[f] = c->sigma[5] * o[f - 1] + c->mu[5];
        for (int k = 0; k < Nz; k++) out[(size_t)col * Nz + k] = -(wT[k + 1] - wT[k]) / dz;
    }
    tape_free(tp);
    return 0;
}
