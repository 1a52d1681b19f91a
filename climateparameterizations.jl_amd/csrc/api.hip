// api.hip — the C ABI of include/colnde.h on top of the MFMA tile engine.  gfx950 only; no CPU fallback.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "engine_tile16.h"
#include "engine_regtile.h"
#include "engine_fc.h"
#include "column_ops.h"

static thread_local std::string g_err;

static int fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

// shared with comm.hip (not part of the public header)
extern "C" int colnde_internal_set_error(const char* msg) { g_err = msg ? msg : "error"; return 1; }

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

enum { K_FORWARD = 0, K_ADJOINT = 1, K_REDUCE = 2, K_RHS = 3, K_INFER = 4, K_DW1 = 5, K_CONVADJ = 6, K_ADAM = 7, K_IMPLDIFF = 8, K_COUNT = 9 };

struct PendingEvent { hipEvent_t a, b; int which; };

struct colnde_handle {
    colnde_config cfg;
    std::vector<float> save_times;
    DevModel m;
    PackInfo pk;
    AdjointGeom geo;
    bool geo_ok = false;
    int device = 0;
    hipStream_t stream = nullptr;
    int n_col = 0, n_tiles = 0;
    int64_t n_col_total = 0;
    size_t lds_fwd = 0, lds_adj = 0, lds_fwd_solve = 0;
    int fwd_threads = 256;
    bool fwd_wlds = false;
    bool adj_helper = true;         // ... and in the adjoint: a helper wave carries λ, x̄ and the physics pullback for the three net waves
    bool fwd_helper = true;         // ... four waves per tile: a helper wave evaluates the Richardson-number closure for the three net waves
    bool split_rich = false;        // ... with the rich tape (activations, derivatives, physics coefficients) in place of the pre-activation tape
    bool adj_split = false;         // ... and the gradient by rt16s_adjoint_kernel + tile16's dW GEMM
    bool fwd_split = false;         // forward solves by the net-split kernels (rt16sh_forward_kernel: three net waves + a helper wave per tile)
    bool use_rt = false;            // register-resident tile engine (static 96-50-20-31 wind-mixing shape)
    bool sp_fwd = true, sp_adj = true, sp_dw = true;   // matrix arithmetic of the forward-solve / adjoint / weight-gradient kernels: exact three-way bf16 split (true) or
                                                      // f32 MFMA — cfg.matrix_arithmetic with the test overrides COLNDE_{FWD,ADJ,DW}_SPLIT (resolve_arithmetic)
    bool use_fc = false;            // 32-column free-convection engine (engine_fc.hip: Nz = 32 | 64, the reference's relu network, RK4)
    float *d_fc_imgf = nullptr, *d_fc_imgb = nullptr, *d_fc_bias = nullptr;
    unsigned int *d_fc_simgf = nullptr, *d_fc_simgb = nullptr;   // the split operand images (COLNDE_MATRIX_BF16X3_EXACT; 32-column tiles)
    unsigned int* d_fc_masks = nullptr;
    unsigned long long* d_fc_switch = nullptr;   // ConvectiveAdjustmentNDE: the taped switch patterns
    int fc_block = 0, fc_nblocks = 0, fc_rows = 0;   // gradient path: columns per pass (multiple of 32), passes, slab rows
    int fc_seg = 0, fc_nseg = 0;                      // ... save intervals per time segment of the tapes, segments (1: the tapes hold the whole axis)
    int fc_cw = 32;                                   // columns per workgroup tile: 32, or 16 for problems of at most 4,096 columns (fc_tile_width)
    float* d_fc_lam = nullptr;                        // λ handed from one time segment to the one before it
    float* d_wimg = nullptr;
    float *d_rt_tape = nullptr, *d_rt_tape2 = nullptr, *d_rt_slab = nullptr, *d_rt_tapez = nullptr;
    bool rt_fwd32 = false;         // COLNDE_RT_FWD=32 at creation: the 32-column forward kernel (no Z1 tape)
    bool rt_ztape = false;         // layer-1 pre-activations taped by the forward kernel instead of recomputed by the adjoint
    int rt_rows = 0;
    int rt_block = 0;              // columns per pass of the gradient path (multiple of 32): the tapes hold one block at a time
    int rt_nblocks = 0;
    float *d_w = nullptr, *d_wf = nullptr, *d_wb = nullptr, *d_x0 = nullptr, *d_bcs = nullptr, *d_truth = nullptr,
          *d_sol = nullptr, *d_tape = nullptr, *d_slab = nullptr, *d_out = nullptr, *d_times = nullptr,
          *d_partial = nullptr, *d_tmp_a = nullptr, *d_tmp_b = nullptr, *d_tmp_c = nullptr;
    size_t tmp_cols = 0;
    TileDesc* d_tiles = nullptr;
    // tile16 taped-dW mode (networks whose weight-gradient tiles overflow the register file)
    int t16_dwtape = -1;            // -1 undecided, 0 off, 1 on
    float* d_dwtape = nullptr;
    float* d_t16_ztape = nullptr;   // taped mode: hidden pre-activations written by the forward kernel (the adjoint skips its forward GEMMs)
    DwMacro* d_macros = nullptr;
    DwSplitPlan dw_split;                       // the dW GEMM on the bf16 pipe (exact operand splitting), built with the tapes' plan whenever the records fit LDS; used when sp_dw
    int n_macros = 0, dw_slices = 0, t16_rows = 0;
    int t16_block = 0, t16_nblocks = 0;   // taped mode: columns per pass (multiple of 16) — the tapes hold one block
    int *d_bias_zoff = nullptr, *d_bias_goff = nullptr;
    bool have_problem = false, have_truth = false;
    bool prof = false;
    int min_substeps = 1;           // least RK4 sub-steps per save interval inside the diffusive stability bound
    bool auto_substeps = false;     // cfg.substeps = 0 at creation: the first solve call chooses the sub-step count from cfg.reltol (choose_substeps)
    float last_estimate = -1.0f;    // ... and the error estimate it settled on
    unsigned* d_sf = nullptr;       // DevModel::sf / sb: bf16 plane images of the dense chains (networks with rows in global memory, BF16X3_EXACT)
    unsigned* d_sb = nullptr;
    float* d_ag = nullptr;          // DevModel::ag: per-tile activation / delta rows in global memory (networks whose rows do not fit the LDS)
    size_t ag_tiles = 0;            // ... tiles it holds
    std::vector<float> rkc_host;    // host copy of the RKC2 coefficient table in use (refresh_rkc)
    bool ag_rows = false;           // the tile16 kernels keep the activation rows in global memory (DevModel::ag)
    bool substeps_chosen = false;   // the count in use came out of choose_substeps_impl (colnde_describe says so, with the estimate)
    float* d_rkc = nullptr;         // RKC2 coefficient table (DevModel::rkc)
    std::vector<PendingEvent> pending;
    double ms[K_COUNT] = {};
    int launches[K_COUNT] = {};
};

extern "C" const char* colnde_last_error(void) { return g_err.c_str(); }
extern "C" int colnde_version(void) { return COLNDE_VERSION; }

// ---- model construction -----------------------------------------------------------------------------
static int validate(const colnde_config* c) {
    if (!c) return fail("null config");
    if (c->model < 0 || c->model > 2) return fail("unknown model %d", c->model);
    if (c->Nz < 4 || c->Nz > 128) return fail("Nz = %d outside 4..128", c->Nz);
    if (c->n_layers < 1 || c->n_layers > COLNDE_MAX_LAYERS) return fail("n_layers = %d outside 1..%d", c->n_layers, COLNDE_MAX_LAYERS);
    const int ns = c->model == COLNDE_MODEL_WIND_MIXING ? 3 * c->Nz : c->Nz;
    if (c->layer_sizes[0] != ns) return fail("first layer input %d != state size %d", c->layer_sizes[0], ns);
    if (c->layer_sizes[c->n_layers] != c->Nz - 1)
        return fail("last layer output %d != Nz-1 = %d interior faces", c->layer_sizes[c->n_layers], c->Nz - 1);
    for (int l = 0; l <= c->n_layers; l++)
        if (c->layer_sizes[l] < 1 || c->layer_sizes[l] > 4096) return fail("layer size %d out of range", c->layer_sizes[l]);
    for (int l = 0; l < c->n_layers; l++)
        if (c->activations[l] < 0 || c->activations[l] > COLNDE_ACT_LEAKYRELU) return fail("unknown activation %d", c->activations[l]);
    if (c->model == COLNDE_MODEL_WIND_MIXING) {
        if (c->modified_pacanowski_philander && c->convective_adjustment && !c->inplace_variant)
            return fail("modified_pacanowski_philander and convective_adjustment are exclusive (NDE_training.jl:171)");
        if (c->zero_weights && !c->modified_pacanowski_philander)
            return fail("zero_weights requires modified_pacanowski_philander (NDE_training.jl:192-194)");
    }
    if (c->n_save < 2 || !c->save_times) return fail("need >= 2 save times");
    if (c->substeps < 0) return fail("substeps must be >= 1 (or 0: chosen from reltol by the first solve, colnde_choose_substeps)");
    if (!(c->reltol >= 0.0f) || c->reltol >= 1.0f) return fail("reltol = %g outside [0, 1) (0 = the reference's 1e-3)", c->reltol);
    for (int i = 1; i < c->n_save; i++)
        if (!(c->save_times[i] > c->save_times[i - 1])) return fail("save_times must be strictly increasing");
    if (c->n_columns < 1) return fail("n_columns must be >= 1");
    if (c->engine != COLNDE_ENGINE_AUTO && c->engine != COLNDE_ENGINE_GENERIC && c->engine != COLNDE_ENGINE_MFMA && c->engine != COLNDE_ENGINE_FC32)
        return fail("unknown engine %d", c->engine);
    if (c->stepper != COLNDE_STEPPER_RK4 && c->stepper != COLNDE_STEPPER_RKC2) return fail("unknown stepper %d", c->stepper);
    if (c->rkc_stages != 0 && (c->rkc_stages < 2 || c->rkc_stages > 256)) return fail("rkc_stages = %d outside 2..256 (0 = automatic)", c->rkc_stages);
    if (c->matrix_arithmetic != COLNDE_MATRIX_BF16X3_EXACT && c->matrix_arithmetic != COLNDE_MATRIX_F32_MFMA)
        return fail("unknown matrix_arithmetic %d (COLNDE_MATRIX_BF16X3_EXACT = 0, COLNDE_MATRIX_F32_MFMA = 1)", c->matrix_arithmetic);
    return 0;
}

// ---- explicit-RK4 stability (DESIGN §2 "Stiffness") ---------------------------------------------------------
// The reference integrates these right-hand sides with a stabilised ROCK4 (wind_mixing/train_NDE.jl:143) because their vertical
// diffusion is stiff in proportion to Nz²: the discrete Laplacian with diffusivity D (in nondimensional time) has eigenvalues down
// to -4 D Nz².  Classical RK4 is stable on the real axis for |lambda dt| <= 2.785.  D is the largest diffusivity the configuration
// can switch on: tau (nu0 + nu_minus) max(1, 1/Pr) / H² for the Richardson-number closure, tau kappa / H² for the
// convective-adjustment branches (NDE_training.jl:141-143; NDE! nu_T switch, training_postprocessing.jl:118-121), C K for
// ConvectiveAdjustmentNDE (convective_adjustment_nde.jl:43-47).  (The MLP's own Jacobian is not bounded here.)
static double stiff_lambda(const colnde_config* c) {
    const double Nz2 = (double)c->Nz * c->Nz;
    double D = 0.0;
    if (c->model == COLNDE_MODEL_WIND_MIXING) {
        const double k = (double)c->tau / ((double)c->H * c->H);
        if (c->modified_pacanowski_philander) {
            D = k * ((double)c->nu0 + c->nu_minus) * fmax(1.0, 1.0 / (double)c->Pr);
            if (c->inplace_variant && c->convective_adjustment) D = fmax(D, k * c->kappa);
        } else if (c->convective_adjustment) {
            D = k * c->kappa;
        }
    } else if (c->model == COLNDE_MODEL_CONV_ADJ_NDE) {
        D = (double)c->sigma[5] / c->sigma[2] * ((double)c->tau / c->H) * c->ca_K;
    }
    return 4.0 * D * Nz2;
}

#define COLNDE_RK4_REAL_BOUND 2.785

// ---- RKC2 (Sommeijer, Shampine, Verwer 1998, eqs. 2.1-2.8; damping eps = 2/13) ------------------------------------------------
//   Y_0 = y,  Y_1 = Y_0 + mu~_1 h F_0,  Y_j = (1 - mu_j - nu_j) Y_0 + mu_j Y_{j-1} + nu_j Y_{j-2} + mu~_j h F_{j-1} + gamma~_j h F_0
// tab = [5][s + 1]: mu, nu, mu~, gamma~, c; returns the real stability boundary beta(s).
static double rkc_tables(int s, std::vector<double>* tab) {
    const double eps = 2.0 / 13.0, w0 = 1.0 + eps / ((double)s * s);
    std::vector<double> T(s + 1, 0.0), dT(s + 1, 0.0), d2T(s + 1, 0.0), b(s + 1, 0.0);
    T[0] = 1.0; T[1] = w0; dT[1] = 1.0;
    for (int j = 2; j <= s; j++) {
        T[j] = 2 * w0 * T[j - 1] - T[j - 2];
        dT[j] = 2 * T[j - 1] + 2 * w0 * dT[j - 1] - dT[j - 2];
        d2T[j] = 4 * dT[j - 1] + 2 * w0 * d2T[j - 1] - d2T[j - 2];
    }
    const double w1 = dT[s] / d2T[s];
    for (int j = 2; j <= s; j++) b[j] = d2T[j] / (dT[j] * dT[j]);
    b[0] = b[1] = b[2];
    if (tab) {
        tab->assign((size_t)5 * (s + 1), 0.0);
        double* mu = tab->data(), *nu = mu + (s + 1), *mut = nu + (s + 1), *gat = mut + (s + 1), *c = gat + (s + 1);
        mut[1] = b[1] * w1;
        c[1] = mut[1];
        for (int j = 2; j <= s; j++) {
            mu[j] = 2 * b[j] * w0 / b[j - 1];
            nu[j] = -b[j] / b[j - 2];
            mut[j] = 2 * b[j] * w1 / b[j - 1];
            gat[j] = -(1.0 - b[j - 1] * T[j - 1]) * mut[j];
            c[j] = w1 * d2T[j] / dT[j];
        }
    }
    return (w0 + 1.0) * d2T[s] / dT[s];
}

#define COLNDE_RKC_SAFETY 0.9

static double widest_interval(const colnde_config* c) {
    double span = 0.0;
    for (int i = 1; i < c->n_save; i++) span = fmax(span, (double)c->save_times[i] - (double)c->save_times[i - 1]);
    return span;
}

extern "C" int colnde_rkc_stages(const colnde_config* c) {
    if (validate(c)) return -1;
    if (c->rkc_stages) return c->rkc_stages;
    const double z = stiff_lambda(c) * widest_interval(c) / (c->substeps > 0 ? c->substeps : 1);
    int s = 2;
    while (s < 256 && COLNDE_RKC_SAFETY * rkc_tables(s, nullptr) < z) s++;
    return s;
}

extern "C" int colnde_min_substeps(const colnde_config* c) {
    if (validate(c)) return -1;
    const double span = widest_interval(c);
    if (c->stepper == COLNDE_STEPPER_RKC2) {          // s stages cover lambda dt <= 0.9 beta(s); at most 256 stages per step
        const int s = c->rkc_stages ? c->rkc_stages : 256;
        const double need = span * stiff_lambda(c) / (COLNDE_RKC_SAFETY * rkc_tables(s, nullptr));
        return need <= 1.0 ? 1 : (int)ceil(need);
    }
    const double need = span * stiff_lambda(c) / COLNDE_RK4_REAL_BOUND;
    return need <= 1.0 ? 1 : (int)ceil(need);
}

static void build_model(const colnde_config* c, DevModel* m, PackInfo* pk) {
    memset(m, 0, sizeof(*m));
    memset(pk, 0, sizeof(*pk));
    const bool wm = c->model == COLNDE_MODEL_WIND_MIXING;
    m->model = c->model;
    m->Nz = c->Nz;
    m->ns = wm ? 3 * c->Nz : c->Nz;
    m->n_nets = wm ? 3 : 1;
    m->n_bc = wm ? 6 : 2;
    m->n_layers = c->n_layers;
    int woff = 0, aoff = 0, nb = 0, pf = 0, pb = 0;
    for (int l = 0; l <= c->n_layers; l++) m->sizes[l] = c->layer_sizes[l];
    for (int l = 0; l < c->n_layers; l++) {
        const int ni = c->layer_sizes[l], no = c->layer_sizes[l + 1];
        m->acts[l] = c->activations[l];
        m->w_off[l] = woff;
        m->b_off[l] = woff + ni * no;
        woff += ni * no + no;
        m->act_off[l] = aoff;
        aoff += (no + 3) & ~3;
        nb += no;
        pk->pf_off[l] = pf;
        pk->pb_off[l] = pb;
        pf += ((no + 15) / 16) * ((ni + 15) / 16) * 256;      // [m-tile][group of 16 k][64 lanes][4]
        pb += ((ni + 15) / 16) * ((no + 15) / 16) * 256;
    }
    pk->pf_off[c->n_layers] = pf;
    pk->pb_off[c->n_layers] = pb;
    pk->pf_net = pf;
    pk->pb_net = pb;
    m->net_size = woff;
    m->n_params = woff * m->n_nets;
    m->act_total = aoff;
    m->n_bias = nb * m->n_nets;
    // tile reads may overrun a feature block by up to 15 floats: keep that inside the row
    m->ld_x = lds_pad(((m->ns + 15) / 16) * 16);
    m->ld_a = lds_pad(aoff + 16);
    m->ld_f = lds_pad(c->Nz + 1);
    m->mpp = wm && c->modified_pacanowski_philander;
    m->ca = wm && c->convective_adjustment;
    m->zero_w = wm && c->zero_weights;
    m->smooth_NN = wm && c->smooth_NN;
    m->smooth_Ri = wm && c->smooth_Ri;
    m->diurnal = wm && c->diurnal;
    m->inplace = wm && c->inplace_variant;
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
    m->sig_u = sg[0]; m->sig_v = sg[1]; m->mu_u = mu[0]; m->mu_v = mu[1];
    m->mu_wT = mu[5]; m->sig_wT = sg[5]; m->mu_T = mu[2]; m->sig_T = sg[2];
    m->nu0 = c->nu0; m->nu_minus = c->nu_minus; m->Ric = c->Ric; m->dRi = c->dRi; m->inv_dRi = 1.0f / c->dRi; m->inv_Pr = 1.0f / c->Pr; m->c_rib = -c->nu_minus / (2.0f * c->dRi); m->Pr = c->Pr;
    m->kappa = c->kappa; m->eps = c->eps; m->ca_K = c->ca_K; m->tau = c->tau; m->alpha_g = c->alpha * c->g;
}

static void build_tables(const DevModel& m, std::vector<TileDesc>* tiles, std::vector<int>* bz, std::vector<int>* bg) {
    for (int net = 0; net < m.n_nets; net++)
        for (int l = 0; l < m.n_layers; l++) {
            const int ni = m.sizes[l], no = m.sizes[l + 1];
            for (int it = 0; it < (ni + 15) / 16; it++)
                for (int jt = 0; jt < (no + 15) / 16; jt++) {
                    TileDesc d;
                    d.a_src = l == 0 ? 0 : 1;
                    d.a_off = (l == 0 ? 0 : m.act_off[l - 1]) + it * 16;
                    d.d_off = m.act_off[l] + jt * 16;
                    d.net = net;
                    d.ni_rem = ni - it * 16 < 16 ? ni - it * 16 : 16;
                    d.no_rem = no - jt * 16 < 16 ? no - jt * 16 : 16;
                    d.g_off = net * m.net_size + m.w_off[l] + it * 16 * no + jt * 16;
                    d.no = no;
                    tiles->push_back(d);
                }
            for (int j = 0; j < no; j++) {
                bz->push_back(net * CT * m.ld_a + m.act_off[l] + j);
                bg->push_back(net * m.net_size + m.b_off[l] + j);
            }
        }
}

// cfg.matrix_arithmetic, per kernel family; COLNDE_FWD_SPLIT / COLNDE_ADJ_SPLIT / COLNDE_DW_SPLIT = 0 | 1 override one family each (test aid:
// isolates one kernel's arithmetic against the others').  Read here only — at creation and in colnde_set_matrix_arithmetic — never per call.
static void resolve_arithmetic(colnde_handle* h) {
    const bool split = h->cfg.matrix_arithmetic == COLNDE_MATRIX_BF16X3_EXACT;
    auto ov = [split](const char* name) {
        const char* e = getenv(name);
        return (e && *e) ? atoi(e) != 0 : split;
    };
    h->sp_fwd = ov("COLNDE_FWD_SPLIT");
    h->sp_adj = ov("COLNDE_ADJ_SPLIT");
    h->sp_dw = ov("COLNDE_DW_SPLIT");
    // tile16 with rows in global memory: the dense chains follow the arithmetic (plane images packed beside the f32 ones)
    h->m.sf = (h->ag_rows && h->sp_fwd) ? h->d_sf : nullptr;
    h->m.sb = (h->ag_rows && h->sp_adj) ? h->d_sb : nullptr;
}

// DevModel::ag for `tiles` workgroups (zero-filled once: the pad slots behind a layer's last feature are never written and must read as zero)
static int ensure_ag(colnde_handle* h, size_t tiles) {
    if (!h->ag_rows || h->ag_tiles >= tiles) return 0;
    (void)hipStreamSynchronize(h->stream);
    if (h->d_ag) { (void)hipFree(h->d_ag); h->d_ag = nullptr; h->m.ag = nullptr; h->ag_tiles = 0; }
    const size_t bytes = tiles * ag_floats_per_tile(h->m) * sizeof(float);
    if (hipMalloc((void**)&h->d_ag, bytes) != hipSuccess) { (void)hipGetLastError(); return fail("hipMalloc of the activation rows (%zu B) failed", bytes); }
    if (hipMemset(h->d_ag, 0, bytes) != hipSuccess) return fail("hipMemset of the activation rows failed");
    h->ag_tiles = tiles;
    h->m.ag = h->d_ag;
    return 0;
}

// The RKC2 coefficient table for the sub-step count now in h->cfg: with rkc_stages = 0 the stage count FOLLOWS the step (least s with 0.9 beta(s) >= lambda dt),
// so colnde_choose_substeps / colnde_set_substeps / the doubled solve of colnde_error_estimate each run with the stage count their own step needs (a table
// frozen at creation would be too short for a longer step — unstable — and wastefully long for a shorter one).  Uploaded in stream order.
static int refresh_rkc(colnde_handle* h) {
    if (h->cfg.stepper != COLNDE_STEPPER_RKC2) return 0;
    const int s = colnde_rkc_stages(&h->cfg);
    if (s < 2) return 1;
    if (s == h->m.nst && !h->rkc_host.empty()) return 0;
    std::vector<double> tab;
    rkc_tables(s, &tab);
    h->rkc_host.assign((size_t)6 * RKC_LD, 0.0f);
    for (int q = 0; q < 5; q++)
        for (int j = 0; j <= s; j++) h->rkc_host[(size_t)q * RKC_LD + j] = (float)tab[(size_t)q * (s + 1) + j];
    // 1 - mu_j - nu_j (~ 1/s^2: formed in double — in float32 the cancellation would cost four digits of the Y_0 weight)
    for (int j = 0; j <= s; j++) h->rkc_host[(size_t)5 * RKC_LD + j] = (float)(1.0 - tab[j] - tab[(size_t)(s + 1) + j]);
    // (the previous table may still be read by kernels in flight on the stream; the host vector is reused, so the copy is completed before returning)
    if (hipMemcpyAsync(h->d_rkc, h->rkc_host.data(), h->rkc_host.size() * sizeof(float), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess)
        return fail("uploading the RKC coefficient table failed");
    h->m.nst = s;
    return 0;
}

extern "C" int colnde_create(const colnde_config* cfg, colnde_handle** out) {
    if (!out) return fail("null out pointer");
    *out = nullptr;
    if (validate(cfg)) return 1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail("no HIP device visible: colnde has no CPU fallback (the product path is the gfx950 HIP engine)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail("device %d not in 0..%d", cfg->device, ndev - 1);
    HIPCHK(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail("device %d is %s; this library is built for gfx950 (MI355X) only", cfg->device, prop.gcnArchName);

    colnde_handle* h = new (std::nothrow) colnde_handle();
    if (!h) return fail("out of host memory");
    h->cfg = *cfg;
    if (cfg->substeps == 0) {       // automatic: start from the least power of two inside the stability bound; the first solve call refines it
        h->auto_substeps = true;
        colnde_config c1 = *cfg;
        c1.substeps = 1;
        const int ms = colnde_min_substeps(&c1);
        int s2 = 1;
        while (s2 < ms) s2 *= 2;
        h->cfg.substeps = s2;
    }
    if (h->cfg.reltol == 0.0f) h->cfg.reltol = 1e-3f;            // solve(...; reltol=1f-3): NDE_training.jl:291
    h->save_times.assign(cfg->save_times, cfg->save_times + cfg->n_save);
    h->cfg.save_times = h->save_times.data();
    h->device = cfg->device;
    h->n_col = cfg->n_columns;
    h->n_col_total = cfg->n_columns;
    h->n_tiles = (cfg->n_columns + CT - 1) / CT;
    h->min_substeps = colnde_min_substeps(&h->cfg);
    resolve_arithmetic(h);
    build_model(cfg, &h->m, &h->pk);
    h->m.nst = 4;
    h->m.rkc = nullptr;
    if (cfg->stepper == COLNDE_STEPPER_RKC2) {
        if (hipMalloc((void**)&h->d_rkc, (size_t)6 * RKC_LD * sizeof(float)) != hipSuccess) {
            delete h;
            return fail("allocating the RKC coefficient table failed");
        }
        h->m.rkc = h->d_rkc;
        if (refresh_rkc(h)) { colnde_destroy(h); return 1; }
    }
    std::vector<TileDesc> tiles;
    std::vector<int> bz, bg;
    build_tables(h->m, &tiles, &bz, &bg);
    h->m.n_tiles = (int)tiles.size();
    const size_t lds_cap = 160 * 1024;
    {   // COLNDE_ADJ_GEOM=<index> / COLNDE_FWD_WLDS=0|1 / COLNDE_FWD_THREADS=256|512 override the heuristics (tuning aid)
        const char* eg = getenv("COLNDE_ADJ_GEOM");
        h->geo_ok = pick_adjoint_geom(h->m, &h->geo, eg ? atoi(eg) : -1);
    }
    h->lds_fwd = lds_floats_forward(h->m) * sizeof(float);
    h->lds_adj = h->geo_ok ? lds_floats_adjoint_geom(h->m, h->geo) * sizeof(float) : lds_floats_adjoint(h->m) * sizeof(float);
    {
        const size_t wl = ((size_t)((h->m.n_params + 3) & ~3) + 128) * sizeof(float);
        // Weights staged in LDS when they fit — unless that leaves room for only ONE workgroup per CU on a problem with more tiles than
        // CUs: there two 256-thread workgroups streaming the packed image from L2 (four k-steps per load) are faster
        // (32-128-128-31, 16,384 columns: forward 55.8 -> 42.9 ms); latency points keep the LDS copy (no L2 round trip per chain).
        h->fwd_wlds = h->lds_fwd + wl <= lds_cap && (h->n_tiles <= 256 || 2 * (h->lds_fwd + wl) <= lds_cap);
        const char* ew = getenv("COLNDE_FWD_WLDS");
        if (ew) h->fwd_wlds = h->lds_fwd + wl <= lds_cap && atoi(ew) != 0;
        // weights in LDS: 1,024 threads (four waves per SIMD; 8 simulations 19.7 vs 21.3 ms at 512, 32-128-128-31 54.7 vs 55.7 ms);
        // weights streamed from L2 (64-256-256-63): two independent 256-thread workgroups per CU do better (113 vs 121 ms)
        h->fwd_threads = h->fwd_wlds ? 1024 : 256;
        const char* et = getenv("COLNDE_FWD_THREADS");
        if (et && (atoi(et) == 256 || atoi(et) == 512 || atoi(et) == 1024)) h->fwd_threads = atoi(et);
        h->lds_fwd_solve = h->lds_fwd + (h->fwd_wlds ? wl : 0);
    }
    h->m.ag = nullptr;
    if (h->lds_fwd > lds_cap) {
        // The reference's wide wind-mixing architectures (3 x 96-400-400-31: wind_mixing/train_NDE.jl:101-102, train_NDE_args.jl:150-166) need 160 KB for the
        // tile's activation rows alone.  They run with THAT array in global memory (DevModel::ag; L2-resident, one slab per workgroup) and everything else as
        // before: weights streamed from L2, 256-thread forward, the taped-dW adjoint with the Z tape (no in-register weight-gradient tiles).
        const size_t need = h->lds_fwd;
        const size_t rest = lds_floats_forward_ag(h->m) * sizeof(float);
        if (rest > lds_cap || CT * h->m.ns > 2 * 1024) {
            delete h;
            return fail("network too large for the tile engine: forward needs %zu B of LDS (> %zu), %zu B even with the activation rows in global memory", need, lds_cap, rest);
        }
        h->ag_rows = true;
        h->lds_fwd = rest;
        h->lds_fwd_solve = rest;
        h->fwd_wlds = false;
        // one 1,024-thread workgroup per tile (four waves per SIMD hide the row and weight loads from L2): 12.4 M column-timesteps/s forward at 4,096 ... 32,768
        // columns against 8.6 M with 256-thread workgroups (tools/wide_threads.py).  COLNDE_FWD_THREADS=256|1024 overrides.
        h->fwd_threads = 1024;
        {
            const char* et = getenv("COLNDE_FWD_THREADS");
            if (et && (atoi(et) == 256 || atoi(et) == 1024)) h->fwd_threads = atoi(et);
        }
        h->geo_ok = false;
        if (ensure_ag(h, (size_t)h->n_tiles)) { colnde_destroy(h); return 1; }
        // plane images of the dense chains (16-byte items): forward [16-row tile][32-deep k-block][3 planes][64 lanes], transposed likewise
        int fo = 0, bo = 0;
        for (int l = 0; l < h->m.n_layers; l++) {
            const int ni = h->m.sizes[l], no = h->m.sizes[l + 1];
            h->m.sf_off[l] = fo;
            h->m.sb_off[l] = bo;
            fo += ((no + 15) / 16) * ((ni + 31) / 32) * 3 * 64;
            bo += ((ni + 15) / 16) * ((no + 31) / 32) * 3 * 64;
        }
        h->m.sf_net = fo;
        h->m.sb_net = bo;
        if (hipMalloc((void**)&h->d_sf, (size_t)fo * h->m.n_nets * 16) != hipSuccess || hipMalloc((void**)&h->d_sb, (size_t)bo * h->m.n_nets * 16) != hipSuccess) {
            (void)hipGetLastError();
            colnde_destroy(h);
            return fail("allocating the bf16 plane images failed");
        }
        resolve_arithmetic(h);
    }
    if (h->lds_adj > lds_cap) h->geo_ok = false;
    {
        hipError_t e = set_kernel_attributes(lds_cap);
        if (e == hipSuccess) e = rt_set_attributes();
        if (e != hipSuccess) { delete h; return fail("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); }
    }
    // AUTO: regtile where it applies, except for problems of at most 8,192 columns (two rounds of one 16-column tile per CU).  Those
    // are latency points for regtile — 32 columns per wavefront leave most SIMDs without a wave — and go to the net-split kernels
    // (one wavefront per flux net plus a helper wavefront per 16-column tile, below) with tile16's tapes and dW GEMM: 4,096 columns x 64 steps
    // 3.55 vs 9.48 ms per iteration, 8,192: 6.99 vs 9.84, 16,384: 12.8 vs 10.7 (tools/crossover.py)
    h->use_rt = rt_supported(h->m) && cfg->stepper == COLNDE_STEPPER_RK4 && cfg->engine != COLNDE_ENGINE_GENERIC &&
                (cfg->engine == COLNDE_ENGINE_MFMA || cfg->n_columns > 8192 || !h->geo_ok);
    h->rt_fwd32 = h->use_rt && rt_forward_is32();
    // AUTO on a regtile-shaped problem too small for regtile (a latency point): the net-split kernels take the forward solves and
    // the adjoint (8 simulations: forward 19.8 -> 8.3 ms, adjoint 25.0 -> 8.2 ms), tile16 the tapes' formats, the dW GEMM and the reduction.
    // An explicit engine = tile16 stays pure tile16; COLNDE_T16_FWD_SPLIT=0|1 and COLNDE_T16_ADJ_SPLIT=0 override.
    {
        // (round 3: the stabilised RKC2 stepper too — in the four-wave kernels, so only with both helper waves on)
        const char* eh = getenv("COLNDE_T16_FWD_HELPER");      // 0: the three-wave forward (8 simulations 9.8 vs 8.3 ms, 4,096 columns 8.27 vs 7.65 ms)
        h->fwd_helper = !(eh && atoi(eh) == 0);
        const char* eah = getenv("COLNDE_T16_ADJ_HELPER");     // 0: the three-wave adjoint (8 simulations 11.3 vs 8.2 ms)
        h->adj_helper = !(eah && atoi(eah) == 0);
        const bool stepper_ok = cfg->stepper == COLNDE_STEPPER_RK4 || (cfg->stepper == COLNDE_STEPPER_RKC2 && h->fwd_helper && h->adj_helper);
        h->fwd_split = !h->use_rt && rt_supported(h->m) && stepper_ok && cfg->engine == COLNDE_ENGINE_AUTO;
        const char* es = getenv("COLNDE_T16_FWD_SPLIT");
        if (es) h->fwd_split = !h->use_rt && rt_supported(h->m) && stepper_ok && atoi(es) != 0;
        // the gradient behind a split forward: rt16s_adjoint_kernel (same decomposition) when the taped mode with both tapes is planned
        const char* ea = getenv("COLNDE_T16_ADJ_SPLIT");
        h->adj_split = h->fwd_split && !(ea && atoi(ea) == 0);
    }
    if (cfg->engine == COLNDE_ENGINE_MFMA && !h->use_rt) {
        delete h;
        return fail("engine = regtile requested, but it covers only Nz=32, three 96-50-20-31 nets, no smoothing, training RHS, RK4");
    }
    // AUTO on the free-convection shape the reference trains: the 32-column engine (COLNDE_FC=0 keeps tile16)
    {
        const char* ef = getenv("COLNDE_FC");
        h->use_fc = fc_supported(h->m, cfg->stepper) && (cfg->engine == COLNDE_ENGINE_FC32 || (cfg->engine == COLNDE_ENGINE_AUTO && !(ef && atoi(ef) == 0)));
        if (cfg->engine == COLNDE_ENGINE_FC32 && !h->use_fc) {
            delete h;
            return fail("engine = fc32 requested, but it covers only FreeConvectionNDE (RK4) and ConvectiveAdjustmentNDE (RK4, RKC2) with Dense(Nz,4Nz,relu), Dense(4Nz,4Nz,relu), Dense(4Nz,Nz-1), Nz = 32 or 64");
        }
        if (h->use_fc) {
            h->fc_cw = fc_tile_width(cfg->n_columns);
            hipError_t e = fc_set_kernel_attributes();
            if (e != hipSuccess) { delete h; return fail("hipFuncSetAttribute (fc32) failed: %s", hipGetErrorString(e)); }
        }
    }
    const DevModel& m = h->m;
#define ALLOC(ptr, n, T)                                                                   \
    do {                                                                                   \
        hipError_t e_ = hipMalloc((void**)&(ptr), (size_t)(n) * sizeof(T));                \
        if (e_ != hipSuccess) {                                                            \
            colnde_destroy(h);                                                             \
            return fail("hipMalloc of %zu bytes failed: %s", (size_t)(n) * sizeof(T), hipGetErrorString(e_)); \
        }                                                                                  \
    } while (0)
    ALLOC(h->d_w, m.n_params, float);
    ALLOC(h->d_wf, (size_t)h->pk.pf_net * m.n_nets, float);
    ALLOC(h->d_wb, (size_t)h->pk.pb_net * m.n_nets, float);
    ALLOC(h->d_wimg, RT_IMG_ALLOC, float);
    if (h->use_fc) {
        ALLOC(h->d_fc_imgf, fc_image_floats(m.Nz), float);
        ALLOC(h->d_fc_imgb, fc_image_floats(m.Nz), float);
        ALLOC(h->d_fc_bias, fc_bias_floats(m.Nz), float);
        ALLOC(h->d_fc_simgf, fc_split_image_words(m.Nz), unsigned int);
        ALLOC(h->d_fc_simgb, fc_split_image_words(m.Nz), unsigned int);
    }
    ALLOC(h->d_x0, (size_t)h->n_col * m.ns, float);
    ALLOC(h->d_bcs, (size_t)h->n_col * m.n_bc, float);
    ALLOC(h->d_sol, (size_t)h->n_col * cfg->n_save * m.ns, float);
    ALLOC(h->d_out, m.n_params + 8, float);
    ALLOC(h->d_times, cfg->n_save, float);
    ALLOC(h->d_partial, 1024 * 8, float);
    ALLOC(h->d_tiles, tiles.size(), TileDesc);
    ALLOC(h->d_bias_zoff, bz.size(), int);
    ALLOC(h->d_bias_goff, bg.size(), int);
#undef ALLOC
    hipError_t e = hipMemcpy(h->d_times, h->save_times.data(), sizeof(float) * cfg->n_save, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->d_tiles, tiles.data(), sizeof(TileDesc) * tiles.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->d_bias_zoff, bz.data(), sizeof(int) * bz.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->d_bias_goff, bg.data(), sizeof(int) * bg.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { colnde_destroy(h); return fail("table upload failed: %s", hipGetErrorString(e)); }
    *out = h;
    return 0;
}

static void drain_events(colnde_handle* h) {
    for (PendingEvent& p : h->pending) {
        float ms = 0.0f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            h->ms[p.which] += ms;
            h->launches[p.which] += 1;
        }
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    h->pending.clear();
}

extern "C" void colnde_destroy(colnde_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    drain_events(h);
    void* ptrs[] = {h->d_rt_tapez, h->d_rt_tape, h->d_rt_tape2, h->d_rt_slab, h->d_wimg, h->d_w, h->d_wf, h->d_wb, h->d_x0, h->d_bcs, h->d_truth, h->d_sol, h->d_tape, h->d_slab, h->d_out,
                    h->d_times, h->d_partial, h->d_tmp_a, h->d_tmp_b, h->d_tmp_c, h->d_tiles, h->d_bias_zoff, h->d_bias_goff, h->d_dwtape, h->d_macros, h->d_t16_ztape, h->d_rkc, h->d_ag, h->d_sf, h->d_sb,
                    h->d_fc_imgf, h->d_fc_imgb, h->d_fc_bias, h->d_fc_masks, h->d_fc_switch, h->d_fc_lam, h->d_fc_simgf, h->d_fc_simgb};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    dw_split_free(h->dw_split);
    delete h;
}

extern "C" int colnde_n_params(const colnde_handle* h) { return h ? h->m.n_params : -1; }
extern "C" int colnde_engine(const colnde_handle* h) { return h ? (h->use_rt ? COLNDE_ENGINE_MFMA : (h->use_fc ? COLNDE_ENGINE_FC32 : COLNDE_ENGINE_GENERIC)) : -1; }

extern "C" int colnde_set_stream(colnde_handle* h, void* s) {
    if (!h) return fail("null handle");
    h->stream = (hipStream_t)s;
    return 0;
}

extern "C" int colnde_set_matrix_arithmetic(colnde_handle* h, int ma) {
    if (!h) return fail("null handle");
    if (ma != COLNDE_MATRIX_BF16X3_EXACT && ma != COLNDE_MATRIX_F32_MFMA)
        return fail("unknown matrix_arithmetic %d (COLNDE_MATRIX_BF16X3_EXACT = 0, COLNDE_MATRIX_F32_MFMA = 1)", ma);
    h->cfg.matrix_arithmetic = ma;
    resolve_arithmetic(h);
    return 0;
}
extern "C" int colnde_matrix_arithmetic(const colnde_handle* h) { return h ? h->cfg.matrix_arithmetic : -1; }

extern "C" int colnde_set_global_columns(colnde_handle* h, int64_t n) {
    if (!h) return fail("null handle");
    if (n < h->n_col) return fail("global column count %lld < local %d", (long long)n, h->n_col);
    h->n_col_total = n;
    return 0;
}

// ---- profiling -----------------------------------------------------------------------------------------
struct Timed {
    colnde_handle* h;
    PendingEvent p;
    bool on;
    Timed(colnde_handle* h_, int which) : h(h_), on(h_->prof) {
        if (!on) return;
        p.which = which;
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(p.a, h->stream);
    }
    ~Timed() {
        if (!on) return;
        (void)hipEventRecord(p.b, h->stream);
        h->pending.push_back(p);
        if (h->pending.size() > 2048) drain_events(h);
    }
};

extern "C" int colnde_set_profiling(colnde_handle* h, int enabled) {
    if (!h) return fail("null handle");
    h->prof = enabled != 0;
    return 0;
}
extern "C" int colnde_kernel_time(colnde_handle* h, int which, float* ms_total, int* n_launches) {
    if (!h) return fail("null handle");
    if (which < 0 || which >= K_COUNT) return fail("kernel id %d outside 0..%d", which, K_COUNT - 1);
    HIPCHK(hipSetDevice(h->device));
    drain_events(h);
    if (ms_total) *ms_total = (float)h->ms[which];
    if (n_launches) *n_launches = h->launches[which];
    return 0;
}
extern "C" int colnde_reset_kernel_times(colnde_handle* h) {
    if (!h) return fail("null handle");
    drain_events(h);
    for (int i = 0; i < K_COUNT; i++) { h->ms[i] = 0; h->launches[i] = 0; }
    return 0;
}

// ---- problem data --------------------------------------------------------------------------------------
static int set_problem_impl(colnde_handle* h, const float* x0, const float* bcs, const float* truth, hipMemcpyKind kind) {
    if (!h) return fail("null handle");
    if (!x0 || !bcs) return fail("x0 and bcs must not be null");
    HIPCHK(hipSetDevice(h->device));
    const DevModel& m = h->m;
    HIPCHK(hipMemcpyAsync(h->d_x0, x0, sizeof(float) * (size_t)h->n_col * m.ns, kind, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_bcs, bcs, sizeof(float) * (size_t)h->n_col * m.n_bc, kind, h->stream));
    if (truth) {
        const size_t n = (size_t)h->n_col * h->cfg.n_save * m.ns;
        if (!h->d_truth) HIPCHK(hipMalloc((void**)&h->d_truth, n * sizeof(float)));
        HIPCHK(hipMemcpyAsync(h->d_truth, truth, n * sizeof(float), kind, h->stream));
    }
    h->have_truth = truth != nullptr;
    h->have_problem = true;
    if (kind == hipMemcpyHostToDevice) HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
extern "C" int colnde_set_problem(colnde_handle* h, const float* x0, const float* bcs, const float* truth) {
    return set_problem_impl(h, x0, bcs, truth, hipMemcpyHostToDevice);
}
extern "C" int colnde_set_problem_dev(colnde_handle* h, const float* x0, const float* bcs, const float* truth) {
    return set_problem_impl(h, x0, bcs, truth, hipMemcpyDeviceToDevice);
}

static int pack(colnde_handle* h, const float* d_weights) {
    hipError_t e = launch_pack(h->m, h->pk, d_weights, h->d_wf, h->d_wb, h->stream);
    if (e != hipSuccess) return fail("pack_weights launch failed: %s", hipGetErrorString(e));
    if (h->d_sf && (h->m.sf || h->m.sb)) {
        e = launch_pack_planes(h->m, d_weights, h->d_sf, h->d_sb, h->stream);
        if (e != hipSuccess) return fail("pack_planes launch failed: %s", hipGetErrorString(e));
    }
    return 0;
}

static void loss_weights(const colnde_handle* h, const float scalings[6], LossWeights* lw) {
    const double np = (double)h->n_col_total * h->cfg.n_save * h->m.Nz;
    const double ng = (double)h->n_col_total * h->cfg.n_save * (h->m.Nz + 1);
    for (int q = 0; q < 8; q++) lw->w[q] = 0.0f;
    if (h->m.model == COLNDE_MODEL_WIND_MIXING) {
        for (int k = 0; k < 3; k++) {
            lw->w[k] = (float)(scalings[k] / np);
            lw->w[3 + k] = (float)(scalings[3 + k] / ng);
        }
    } else {
        lw->w[2] = (float)(scalings[2] / np);
    }
}

// ---- rhs -----------------------------------------------------------------------------------------------
extern "C" int colnde_rhs_dev(colnde_handle* h, const float* d_x, const float* d_weights, const float* d_bcs, float t,
                              float* d_dx, int n_columns) {
    if (!h) return fail("null handle");
    if (!d_x || !d_weights || !d_bcs || !d_dx) return fail("null pointer argument");
    if (n_columns < 1) return fail("n_columns must be >= 1");
    HIPCHK(hipSetDevice(h->device));
    if (pack(h, d_weights)) return 1;
    if (ensure_ag(h, ((size_t)n_columns + CT - 1) / CT)) return 1;
    Timed tm(h, K_RHS);
    hipError_t e = launch_rhs(h->m, h->pk, d_weights, h->d_wf, d_x, d_bcs, t, d_dx, n_columns, 256, h->lds_fwd, h->stream);
    if (e != hipSuccess) return fail("rhs launch failed: %s", hipGetErrorString(e));
    return 0;
}

static int ensure_tmp(colnde_handle* h, size_t n_columns) {
    if (h->tmp_cols >= n_columns) return 0;
    for (float** p : {&h->d_tmp_a, &h->d_tmp_b, &h->d_tmp_c})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    h->tmp_cols = 0;
    HIPCHK(hipMalloc((void**)&h->d_tmp_a, n_columns * h->m.ns * sizeof(float)));
    HIPCHK(hipMalloc((void**)&h->d_tmp_b, n_columns * 8 * sizeof(float)));
    HIPCHK(hipMalloc((void**)&h->d_tmp_c, n_columns * h->m.ns * sizeof(float)));
    h->tmp_cols = n_columns;
    return 0;
}

extern "C" int colnde_rhs(colnde_handle* h, const float* x, const float* weights, const float* bcs, float t, float* dx,
                          int n_columns) {
    if (!h) return fail("null handle");
    if (!x || !weights || !bcs || !dx) return fail("null pointer argument");
    if (n_columns < 1) return fail("n_columns must be >= 1");
    HIPCHK(hipSetDevice(h->device));
    if (ensure_tmp(h, (size_t)n_columns)) return 1;
    const DevModel& m = h->m;
    HIPCHK(hipMemcpyAsync(h->d_w, weights, sizeof(float) * m.n_params, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_tmp_a, x, sizeof(float) * (size_t)n_columns * m.ns, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_tmp_b, bcs, sizeof(float) * (size_t)n_columns * m.n_bc, hipMemcpyHostToDevice, h->stream));
    if (colnde_rhs_dev(h, h->d_tmp_a, h->d_w, h->d_tmp_b, t, h->d_tmp_c, n_columns)) return 1;
    HIPCHK(hipMemcpyAsync(dx, h->d_tmp_c, sizeof(float) * (size_t)n_columns * m.ns, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// Sizes the regtile engine's tapes.  They hold ONE block of columns; a problem whose tapes exceed the free HBM (many columns, or a
// long horizon: 1,153 frames need 4x the bytes per column of the 2-day suite) runs its gradient path block after block into the same
// buffers.  COLNDE_RT_BLOCK=<columns> forces a block size (testing aid).
static int rt_plan_tapes(colnde_handle* h) {
    if (h->d_rt_tape) return 0;
    const int n_steps = (h->cfg.n_save - 1) * h->cfg.substeps;
    const char* ez = getenv("COLNDE_RT_ZTAPE");
    bool want_z = !h->rt_fwd32 && !(ez && atoi(ez) == 0);
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    const size_t margin = (size_t)2 << 30;
    const size_t budget = free_b > margin ? free_b - margin : 0;
    const size_t per_col_x = (size_t)n_steps * 4 * 96 * sizeof(float), per_col_2 = (size_t)n_steps * 4 * ((21 * 256) / 32) * sizeof(float);
    const size_t per_col_z = rt_tapez_floats(32, n_steps) / 32 * sizeof(float);      // (twice per_col_2 in the A/B build that tapes activation pairs)
    const int n32 = ((h->n_col + 31) / 32) * 32;
    int block = 0;
    for (int pass = 0; pass < 2 && block == 0; pass++) {
        const size_t per_col = per_col_x + per_col_2 + (want_z ? per_col_z : 0);
        const size_t fit = budget / per_col;
        if (fit >= (size_t)n32) block = n32;
        else if (fit >= 1024) {
            const int nb = (int)(((size_t)n32 + fit - 1) / fit);
            block = (((n32 + nb - 1) / nb) + 1023) / 1024 * 1024;
            if ((size_t)block > fit) block = (int)(fit / 1024) * 1024;
        } else if (want_z) want_z = false;       // not even 1,024 columns with the Z1 tape: try without it
    }
    const char* eb = getenv("COLNDE_RT_BLOCK");
    if (eb && atoi(eb) >= 32) block = std::min(n32, (atoi(eb) / 32) * 32);
    if (block == 0) return fail("the stage tapes of even 1,024 columns (%zu bytes per column) do not fit in %zu free bytes of HBM",
                                per_col_x + per_col_2, free_b);
    const size_t n1 = rt_tape_floats(block, n_steps), n2 = rt_tape2_floats(block, n_steps);
    hipError_t e = hipMalloc((void**)&h->d_rt_tape, n1 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_rt_tape2, n2 * sizeof(float));
    h->rt_ztape = want_z;
    if (e == hipSuccess && h->rt_ztape && hipMalloc((void**)&h->d_rt_tapez, rt_tapez_floats(block, n_steps) * sizeof(float)) != hipSuccess) {
        (void)hipGetLastError();
        h->d_rt_tapez = nullptr;
        h->rt_ztape = false;
    }
    if (e != hipSuccess) {
        for (float** q : {&h->d_rt_tape, &h->d_rt_tape2, &h->d_rt_tapez})      // leave no half-built state behind
            if (*q) { (void)hipFree(*q); *q = nullptr; }
        return fail("hipMalloc of the %zu-byte stage tapes failed: %s", (n1 + n2) * sizeof(float), hipGetErrorString(e));
    }
    h->rt_block = block;
    h->rt_nblocks = (n32 + block - 1) / block;
    return 0;
}

// regtile forward solve of columns [c0, c0 + nc): sol rows and, when taping, the block's tapes
static int rt_forward_range(colnde_handle* h, float* d_sol, bool with_tape, int c0, int nc) {
    const size_t ns = h->m.ns;
    Timed tm(h, K_FORWARD);
    hipError_t e = rt_launch_forward(h->m, h->d_wimg, h->d_x0 + (size_t)c0 * ns, h->d_bcs + (size_t)c0 * h->m.n_bc, h->d_times, h->cfg.n_save,
                                     h->cfg.substeps, d_sol ? d_sol + (size_t)c0 * h->cfg.n_save * ns : nullptr,
                                     with_tape ? h->d_rt_tape : nullptr, (with_tape && h->rt_ztape) ? h->d_rt_tapez : nullptr, nc,
                                     h->rt_fwd32, h->sp_fwd, h->stream);
    if (e != hipSuccess) return fail("rt forward launch failed: %s", hipGetErrorString(e));
    return 0;
}

// tile16 forward solve of columns [c0, c0 + nc) (c0 a multiple of the 16-column tile); with_tape: into the handle's (block) tapes
static int t16_forward_range(colnde_handle* h, const float* d_weights, float* d_sol, bool with_tape, int c0, int nc) {
    const size_t ns = h->m.ns;
    if (with_tape && !h->d_tape) {     // in-register gradient mode: the whole problem's stage tape
        const size_t n = (size_t)h->n_tiles * (h->cfg.n_save - 1) * h->cfg.substeps * h->m.nst * CT * ns;
        hipError_t e = hipMalloc((void**)&h->d_tape, n * sizeof(float));
        if (e != hipSuccess) return fail("hipMalloc of the %zu-byte stage tape failed: %s", n * sizeof(float), hipGetErrorString(e));
    }
    Timed tm(h, K_FORWARD);
    if (h->fwd_split) {
        // latency points of the wind-mixing shape: one wavefront per flux net (+ a helper) per 16-column tile (engine_regtile.hip); the tapes
        // come out in tile16's formats for its taped adjoint
        hipError_t es = rt_launch_pack(h->m, d_weights, h->d_wimg, h->stream);
        if (es == hipSuccess)
            es = rt_launch_forward_split(h->m, h->d_wimg, h->d_x0 + (size_t)c0 * ns, h->d_bcs + (size_t)c0 * h->m.n_bc, h->d_times, h->cfg.n_save,
                                         h->cfg.substeps, d_sol ? d_sol + (size_t)c0 * h->cfg.n_save * ns : nullptr,
                                         with_tape ? h->d_tape : nullptr, with_tape ? h->d_t16_ztape : nullptr, nc, with_tape && h->split_rich, h->fwd_helper, h->sp_fwd, h->stream);
        if (es != hipSuccess) return fail("split forward launch failed: %s", hipGetErrorString(es));
        return 0;
    }
    hipError_t e = launch_forward(h->m, h->pk, d_weights, h->d_wf, h->d_x0 + (size_t)c0 * ns, h->d_bcs + (size_t)c0 * h->m.n_bc, h->d_times,
                                  h->cfg.n_save, h->cfg.substeps, d_sol ? d_sol + (size_t)c0 * h->cfg.n_save * ns : nullptr,
                                  with_tape ? h->d_tape : nullptr, nc, h->fwd_threads, h->fwd_wlds, h->lds_fwd_solve, h->stream,
                                  with_tape ? h->d_t16_ztape : nullptr);
    if (e != hipSuccess) return fail("forward launch failed: %s", hipGetErrorString(e));
    return 0;
}

// fc32 forward solve of columns [c0, c0 + nc) (c0 a multiple of 32) over the save intervals [iv0, iv1); with_tape: into the handle's records and
// bits.  iv0 > 0 restarts from the state a previous (tape-less) pass saved in d_sol at save point iv0 — exact: the saved state is the stepper's.
static int fc_forward_range(colnde_handle* h, float* d_sol, bool with_tape, int c0, int nc, int iv0 = 0, int iv1 = -1, int tape_iv0 = -1) {
    const size_t ns = h->m.ns;
    if (iv1 < 0) iv1 = h->cfg.n_save - 1;
    if (tape_iv0 < 0) tape_iv0 = iv0;                // (a later tape_iv0: the tape-less pass of a segmented gradient tapes its last segment on the way)
    const float* init = iv0 == 0 ? h->d_x0 + (size_t)c0 * ns : d_sol + ((size_t)c0 * h->cfg.n_save + iv0) * ns;
    const size_t stride = iv0 == 0 ? ns : (size_t)h->cfg.n_save * ns;
    Timed tm(h, K_FORWARD);
    hipError_t e = fc_launch_forward(h->m, h->fc_cw, h->d_fc_imgf, (h->sp_fwd && fc_split_supported(h->fc_cw)) ? h->d_fc_simgf : nullptr, h->d_fc_bias, init, stride, h->d_bcs + (size_t)c0 * h->m.n_bc, h->d_times, h->cfg.n_save,
                                     iv0, iv1, tape_iv0, h->cfg.substeps, d_sol ? d_sol + (size_t)c0 * h->cfg.n_save * ns : nullptr,
                                     with_tape ? h->d_dwtape : nullptr, with_tape ? h->d_fc_masks : nullptr, with_tape ? h->d_fc_switch : nullptr, nc,
                                     h->stream);
    if (e != hipSuccess) return fail("fc32 forward launch failed: %s", hipGetErrorString(e));
    return 0;
}

// ---- forward solve -------------------------------------------------------------------------------------
// A time step outside RK4's stability region gives a blown-up or NaN trajectory, loss and gradient with rc = 0: refuse it.
static int check_stability(const colnde_handle* h) {
    if (h->cfg.substeps >= h->min_substeps) return 0;
    const char* e = getenv("COLNDE_ALLOW_UNSTABLE_DT");
    if (e && atoi(e) != 0) return 0;
    if (h->cfg.stepper == COLNDE_STEPPER_RKC2)
        return fail("substeps = %d with %d RKC2 stages does not cover the stiffest diffusive mode (lambda = -%.4g): need substeps >= %d, "
                    "or rkc_stages = 0 for the automatic stage count (COLNDE_ALLOW_UNSTABLE_DT=1 overrides)",
                    h->cfg.substeps, h->m.nst, stiff_lambda(&h->cfg), h->min_substeps);
    return fail("substeps = %d puts the RK4 step outside its stability region: the stiffest diffusive mode this configuration can "
                "switch on has lambda = -%.4g (4 D Nz^2), and lambda dt must stay within %.3f — need substeps >= %d "
                "(colnde_min_substeps; the reference uses a stabilised ROCK4 here; COLNDE_ALLOW_UNSTABLE_DT=1 overrides)",
                h->cfg.substeps, stiff_lambda(&h->cfg), COLNDE_RK4_REAL_BOUND, h->min_substeps);
}

static int choose_substeps_impl(colnde_handle* h, const float* d_weights, float reltol, int* chosen, float* estimate);

// substeps = 0 picks the count from THIS handle's columns.  On a column shard (colnde_set_global_columns > the local count) every rank would settle on its
// own count: the SUM-all-reduced gradient would mix discretisations and the tapes would differ per rank.  The sharded recipe is explicit:
// colnde_choose_substeps on every rank, MAX over ranks, colnde_set_substeps (colnde.distributed.agree_substeps does exactly that).
static int refuse_auto_on_a_shard(const colnde_handle* h) {
    if (h->n_col_total > h->n_col)
        return fail("substeps = 0 (chosen from reltol) on a column shard (%d of %lld columns): each rank would choose its own count — call colnde_choose_substeps "
                    "on every rank, take the MAX over ranks and impose it with colnde_set_substeps", h->n_col, (long long)h->n_col_total);
    return 0;
}

static int forward_impl(colnde_handle* h, const float* d_weights, float* d_sol, bool with_tape) {
    if (!h->have_problem) return fail("colnde_set_problem has not been called");
    if (h->auto_substeps) {           // substeps = 0 at creation: settle the count now, from these weights and cfg.reltol
        if (refuse_auto_on_a_shard(h)) return 1;
        h->auto_substeps = false;
        if (choose_substeps_impl(h, d_weights, h->cfg.reltol, nullptr, nullptr)) { h->auto_substeps = true; return 1; }
    }
    if (check_stability(h)) return 1;
    if (h->use_rt) {
        // (the gradient path tapes block by block: colnde_loss_grad_dev drives rt_forward_range itself)
        hipError_t e = rt_launch_pack(h->m, d_weights, h->d_wimg, h->stream);
        if (e != hipSuccess) return fail("rt pack launch failed: %s", hipGetErrorString(e));
        return rt_forward_range(h, d_sol, false, 0, h->n_col);
    }
    if (h->use_fc) {
        hipError_t e = fc_launch_pack(h->m, h->fc_cw, d_weights, h->d_fc_imgf, h->d_fc_imgb, h->d_fc_bias, h->d_fc_simgf, h->d_fc_simgb, h->stream);
        if (e != hipSuccess) return fail("fc32 pack launch failed: %s", hipGetErrorString(e));
        return fc_forward_range(h, d_sol, false, 0, h->n_col);
    }
    if (pack(h, d_weights)) return 1;
    return t16_forward_range(h, d_weights, d_sol, with_tape, 0, h->n_col);
}


extern "C" int colnde_forward_dev(colnde_handle* h, const float* d_weights, float* d_sol) {
    if (!h) return fail("null handle");
    if (!d_weights) return fail("null weights");
    HIPCHK(hipSetDevice(h->device));
    return forward_impl(h, d_weights, d_sol ? d_sol : h->d_sol, false);
}

extern "C" int colnde_forward(colnde_handle* h, const float* weights, float* sol) {
    if (!h) return fail("null handle");
    if (!weights) return fail("null weights");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(h->d_w, weights, sizeof(float) * h->m.n_params, hipMemcpyHostToDevice, h->stream));
    if (forward_impl(h, h->d_w, h->d_sol, false)) return 1;
    if (sol)
        HIPCHK(hipMemcpyAsync(sol, h->d_sol, sizeof(float) * (size_t)h->n_col * h->cfg.n_save * h->m.ns, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// ---- loss / loss + gradient ----------------------------------------------------------------------------
extern "C" int colnde_loss_dev(colnde_handle* h, const float* d_weights, const float scalings[6], float* d_out8) {
    if (!h) return fail("null handle");
    if (!d_weights || !scalings || !d_out8) return fail("null pointer argument");
    if (!h->have_truth) return fail("no truth trajectories: pass truth to colnde_set_problem");
    HIPCHK(hipSetDevice(h->device));
    if (forward_impl(h, d_weights, h->d_sol, false)) return 1;
    LossWeights lw;
    loss_weights(h, scalings, &lw);
    const int nblk = 256;
    hipError_t e = launch_loss(h->m, h->d_sol, h->d_truth, h->cfg.n_save, h->n_col, h->d_partial, nblk, h->stream);
    if (e == hipSuccess) e = launch_reduce(h->d_partial, nblk, 0, 8, lw, d_out8, h->stream);
    if (e != hipSuccess) return fail("loss launch failed: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int colnde_loss(colnde_handle* h, const float* weights, const float scalings[6], float terms[6], float* total) {
    if (!h) return fail("null handle");
    if (!weights || !scalings || !terms || !total) return fail("null pointer argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(h->d_w, weights, sizeof(float) * h->m.n_params, hipMemcpyHostToDevice, h->stream));
    if (colnde_loss_dev(h, h->d_w, scalings, h->d_out)) return 1;
    float o[8];
    HIPCHK(hipMemcpyAsync(o, h->d_out, sizeof(o), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int q = 0; q < 6; q++) terms[q] = o[q];
    *total = o[6];
    if (!std::isfinite(o[6])) return fail("the loss is not finite (%g): the solve left the stable regime (time step, weights or inputs)", o[6]);
    return 0;
}

// tile16, taped weight gradients: decide once per handle.  On when the tapes fit in the free HBM (the in-register adjoint_kernel
// geometries remain the fallback; 64-256-256-63's 384 gradient tiles spill there);
// COLNDE_T16_DWTAPE=1 / 0 forces it on / off.
// The dW GEMM's work list: 64x64 blocks of every layer's weight matrix, and the number of K-slices of the records
static void build_dw_macros(colnde_handle* h, size_t n_rec, std::vector<DwMacro>& mac) {
    const DevModel& m = h->m;
    const size_t R = dwtape_row_floats(m);
    std::vector<int> matrix_of;
    for (int net = 0; net < m.n_nets; net++)
        for (int l = 0; l < m.n_layers; l++) {
            const int ni = m.sizes[l], no = m.sizes[l + 1];
            for (int i0 = 0; i0 < ni; i0 += 64)
                for (int j0 = 0; j0 < no; j0 += 64) {
                    DwMacro d;
                    d.a_feat = l == 0 ? i0 : dwtape_ns4(m) + net * dwtape_act4(m) + m.act_off[l - 1] + i0;
                    d.d_feat = dwtape_ns4(m) + (m.n_nets + net) * dwtape_act4(m) + m.act_off[l] + j0;
                    d.ni_rem = std::min(64, ni - i0);
                    d.no_rem = std::min(64, no - j0);
                    d.g_off = net * m.net_size + m.w_off[l] + i0 * no + j0;
                    d.no = no;
                    mac.push_back(d);
                    matrix_of.push_back(net * m.n_layers + l);
                }
        }
    h->n_macros = (int)mac.size();
    dw_split_free(h->dw_split);
    // the split (bf16-pipe) dW GEMM keeps only a pass's operand FEATURES in LDS, as planes: it also serves records that do not fit the LDS whole (the wide
    // wind-mixing networks: 325 KB per 16-column record), with the large matrices cut into chunks of output blocks (dw_split_build)
    const bool lds_fit = dw_gemm_lds_fits((int)R, h->n_macros);
    const bool split_ok = dw_split_build(mac, matrix_of, (int)R, h->dw_split);
    const int n_groups = (h->n_macros + 3) / 4;
    size_t slices = std::max<size_t>(8, ((size_t)2048 / n_groups + 7) / 8 * 8);
    slices = std::min(slices, std::max<size_t>(8, (n_rec / 8 + 7) / 8 * 8));
    if (lds_fit || (split_ok && h->sp_dw))           // one workgroup per CU (two records / two plane buffers in LDS): two rounds of slices
        slices = std::min<size_t>(512, std::max<size_t>(1, n_rec));
    h->dw_slices = (int)slices;
}

// fc32 gradient path: the records (forward: xs, a1, a2; adjoint: dz1, dz2, dz3) and the bit tapes.  When they do not fit in the free HBM for the
// whole problem there are two ways to cut it:
//   * column blocks (a multiple of the 32-column tile): forward -> adjoint -> dW GEMM block after block.  No extra work, but a block of fewer than
//     16,384 columns leaves CUs with one workgroup or none (the kernels get their speed from two per CU);
//   * time segments: ALL columns, the tapes hold `fc_seg` save intervals.  One tape-less forward pass first (it saves the state at every save
//     point, and tapes the last segment on its way), then, from the last segment to the first, a taped forward restarted from the saved state,
//     the adjoint over the segment (λ handed on through d_fc_lam) and the dW GEMM.  Costs (n_seg - 1)/n_seg of an extra forward solve, keeps every
//     CU at two workgroups.
// Blocks are used when they hold at least 16,384 columns (or everything), segments otherwise; COLNDE_FC_BLOCK=<columns> / COLNDE_FC_SEG=<intervals> force.
static int fc_plan_tapes(colnde_handle* h) {
    if (h->d_dwtape) return 0;
    const DevModel& m = h->m;
    const int n_iv = h->cfg.n_save - 1;
    const size_t R = dwtape_row_floats(m);
    if (R != fc_record_row_floats(m.Nz)) return fail("fc32: record layout mismatch (%zu vs %zu floats per column)", R, fc_record_row_floats(m.Nz));
    const bool ca = m.model == COLNDE_MODEL_CONV_ADJ_NDE;
    // bytes of tape per column and save interval
    const int cw = h->fc_cw;
    const size_t per_col_iv = (size_t)h->cfg.substeps * m.nst * (R * sizeof(float) + fc_mask_words() * sizeof(unsigned int) / cw + (ca ? sizeof(unsigned long long) : 0));
    const int n32 = (h->n_col + 31) / 32 * 32;
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    const size_t margin = ((size_t)3 << 30) + (size_t)(n32 / cw * 8 + 4096) * (m.n_params + 8) * sizeof(float) + (size_t)n32 * m.Nz * sizeof(float);
    const size_t budget = free_b > margin ? free_b - margin : 0;
    const size_t fit = budget / (per_col_iv * n_iv);                 // columns whose whole-axis tapes fit
    int block = 0, seg = n_iv;
    if (fit >= (size_t)n32) block = n32;
    else {
        if (fit >= 32) {
            const int nb = (int)(((size_t)n32 + fit - 1) / fit);
            block = ((n32 + nb - 1) / nb + 31) / 32 * 32;
            if (block >= 8192) block = (block + 8191) / 8192 * 8192;       // whole rounds of one workgroup pair per CU
            while ((size_t)block > fit) block -= block > 8192 ? 8192 : 32;
        }
        if (block < 16384) {
            // time segments of all columns instead (or of the largest column block one interval fits for)
            const size_t cols_iv = budget / per_col_iv / 32 * 32;     // columns whose ONE-interval tapes fit
            if (cols_iv >= 32) {
                block = (int)std::min<size_t>((size_t)n32, cols_iv);
                if (block < n32 && block >= 8192) block = block / 8192 * 8192;
                seg = (int)std::min<size_t>((size_t)n_iv, budget / (per_col_iv * (size_t)block));
                // the partial-gradient slab grows with the number of segments ([tile][segment] + [block][segment][<= 512 slices] rows of n_params + 8
                // floats: 50 GB at 128 segments of the 64-level network): it must fit beside the tapes it is chosen for
                auto slab_bytes = [&](int sg) {
                    const size_t nsg = ((size_t)n_iv + sg - 1) / sg, nblk = ((size_t)n32 + block - 1) / block;
                    return ((size_t)(n32 / cw) + nblk * 512) * nsg * (size_t)(m.n_params + 8) * sizeof(float);
                };
                while (seg > 1 && per_col_iv * (size_t)block * seg + slab_bytes(seg) > budget) seg--;
                if (per_col_iv * (size_t)block * seg + slab_bytes(seg) > budget) seg = 0;      // not even one interval with its slab: reported below
            }
        }
    }
    const char* eb = getenv("COLNDE_FC_BLOCK");
    if (eb && atoi(eb) >= 32) { block = std::min(n32, (atoi(eb) / 32) * 32); if (!getenv("COLNDE_FC_SEG")) seg = n_iv; }
    const char* es = getenv("COLNDE_FC_SEG");
    if (es && atoi(es) >= 1) seg = std::min(n_iv, atoi(es));
    if (block < 32 || seg < 1) return fail("fc32: the tapes of even one 32-column tile and one save interval (%zu bytes) do not fit in the free device memory", 32 * per_col_iv);
    h->fc_block = block;
    h->fc_nblocks = (n32 + block - 1) / block;
    h->fc_seg = seg;
    h->fc_nseg = (n_iv + seg - 1) / seg;
    const size_t tiles_b = (size_t)block / cw;                                  // (block is a multiple of 32)
    const size_t stage_recs = (size_t)seg * h->cfg.substeps * m.nst;          // (tile, stage) records per tile held by the tapes
    const size_t n_rec = tiles_b * (cw / 16) * stage_recs;                     // records are tile16's: 16 columns each
    std::vector<DwMacro> mac;
    build_dw_macros(h, n_rec, mac);
    h->fc_rows = (n32 / cw) * h->fc_nseg + h->fc_nblocks * h->fc_nseg * h->dw_slices;
    const int stride = m.n_params + 8;
    hipError_t e = hipMalloc((void**)&h->d_dwtape, n_rec * CT * R * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_fc_masks, tiles_b * stage_recs * fc_mask_words() * sizeof(unsigned int));
    if (e == hipSuccess && ca) e = hipMalloc((void**)&h->d_fc_switch, tiles_b * stage_recs * fc_switch_words(cw) * sizeof(unsigned long long));
    if (e == hipSuccess && h->fc_nseg > 1) e = hipMalloc((void**)&h->d_fc_lam, (size_t)n32 * m.Nz * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_macros, mac.size() * sizeof(DwMacro));
    if (e == hipSuccess) e = hipMemcpy(h->d_macros, mac.data(), mac.size() * sizeof(DwMacro), hipMemcpyHostToDevice);
    if (e == hipSuccess && !h->d_slab) e = hipMalloc((void**)&h->d_slab, (size_t)h->fc_rows * stride * sizeof(float));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        for (void** p : {(void**)&h->d_dwtape, (void**)&h->d_fc_masks, (void**)&h->d_fc_switch, (void**)&h->d_fc_lam, (void**)&h->d_macros})
            if (*p) { (void)hipFree(*p); *p = nullptr; }
        return fail("fc32: hipMalloc of the tapes (%zu bytes for %d columns x %d save intervals) failed: %s", (size_t)block * per_col_iv * seg, block, seg,
                    hipGetErrorString(e));
    }
    return 0;
}

static int t16_plan_dwtape(colnde_handle* h) {
    if (h->t16_dwtape >= 0) return 0;
    const DevModel& m = h->m;
    const char* ev = getenv("COLNDE_T16_DWTAPE");
    // default: on whenever the tapes fit — with the hidden pre-activations taped too, the 1,024-thread accumulator-free adjoint beats
    // the in-register kernels at every size measured (8 columns: 53 vs 64 ms per iteration; 32-128-128-31: 162 vs 198 ms)
    bool want = ev ? atoi(ev) != 0 : true;
    const int n_steps = (h->cfg.n_save - 1) * h->cfg.substeps;
    const size_t R = dwtape_row_floats(m);
    if (want && (size_t)CT * m.ns > 6 * 512) want = false;
    // LDS of the taped adjoint: with the Z tape (the default) it holds no activation array; a network that fits only that way (3 x 96-400-31: 166 KB with
    // the array, 83 KB without) then NEEDS the Z tape
    bool need_z = false;
    if (want && !h->ag_rows && (MODEL_FLOATS + lds_floats_adjoint(m)) * sizeof(float) > 160 * 1024) {
        const char* ez = getenv("COLNDE_T16_ZTAPE");
        if (!(ez && atoi(ez) == 0) && (MODEL_FLOATS + lds_floats_adjoint_noA(m)) * sizeof(float) <= 160 * 1024 && m.n_bias <= 4 * 512) need_z = true;
        else want = false;
    }
    if (h->ag_rows) {       // rows in global memory: the taped adjoint with the Z tape is the ONLY gradient path (1,024 threads: n_bias <= 4 x 1,024, CT ns <= 2 x 1,024)
        if (!want) return fail("COLNDE_T16_DWTAPE=0: this network's activation rows live in global memory, and only the taped-dW adjoint runs that way");
        if (m.n_bias > 4 * 1024 || (MODEL_FLOATS + lds_floats_adjoint_ag(m)) * sizeof(float) > 160 * 1024)
            return fail("network too large for the taped adjoint with rows in global memory (%d biases, %zu B of LDS)", m.n_bias, (MODEL_FLOATS + lds_floats_adjoint_ag(m)) * sizeof(float));
    }
    // The tapes hold ONE block of columns (a multiple of the 16-column tile; whole rounds of 4,096 columns = one workgroup per CU when
    // possible); larger problems run forward -> adjoint -> dW GEMM block after block.  COLNDE_T16_BLOCK=<columns> forces a size.
    const size_t per_col = (size_t)n_steps * m.nst * (R + t16_ztape_col_floats(m) + m.ns) * sizeof(float);
    const int n16 = h->n_tiles * CT;
    int block = 0;
    // the net-split pair's rich tape replaces the pre-activation tape on small blocks and is 4x its size per column (13,824 floats per
    // 16-column record against 216 per column): it is part of the fit estimate, not an afterthought behind the 3 GB margin
    const char* ezt0 = getenv("COLNDE_T16_ZTAPE");
    const char* er0 = getenv("COLNDE_T16_SPLIT_RICH");
    const bool rich_possible = h->adj_split && !(ezt0 && atoi(ezt0) == 0) && !(er0 && atoi(er0) == 0);
    const size_t per_col_rich = (size_t)n_steps * m.nst * (R + rt_split_rich_record_floats() / CT + m.ns) * sizeof(float);
    bool want_rich = false;
    if (want) {
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        const size_t margin = ((size_t)3 << 30) + (size_t)h->n_tiles * (m.n_params + 8) * sizeof(float);
        const size_t budget = free_b > margin ? free_b - margin : 0;
        const size_t fit = budget / per_col;
        if (fit >= (size_t)n16) block = n16;
        else if (fit >= CT) {
            const int nb = (int)(((size_t)n16 + fit - 1) / fit);
            block = ((n16 + nb - 1) / nb + CT - 1) / CT * CT;
            if (block >= 4096) block = (block + 4095) / 4096 * 4096;
            while ((size_t)block > fit) block -= block > 4096 ? 4096 : CT;
        }
        const char* eb = getenv("COLNDE_T16_BLOCK");
        if (eb && atoi(eb) >= CT) block = std::min(n16, (atoi(eb) / CT) * CT);
        if (block <= 0) want = false;
        // rich tape: by default on blocks of at most 2,048 columns; forced on by COLNDE_T16_SPLIT_RICH=1.  It must fit WITH the other tapes:
        // a forced rich tape shrinks the block, an automatic one is dropped.
        want_rich = want && rich_possible && ((er0 && atoi(er0) != 0) || block <= 128 * CT);
        if (want_rich && (size_t)block * per_col_rich > budget) {
            const size_t fit_rich = budget / per_col_rich / CT * CT;
            if (er0 && atoi(er0) != 0 && fit_rich >= CT) block = (int)std::min<size_t>((size_t)block, fit_rich);
            else want_rich = false;
        }
    }
    if (!want) {
        if (h->ag_rows) return fail("the delta tape of this network (%zu B per column) does not fit in HBM beside the solve", per_col);
        h->t16_dwtape = 0;
        return 0;
    }
    const int tiles_b = block / CT;
    const size_t n_rec = (size_t)tiles_b * n_steps * h->m.nst;
    const size_t need = n_rec * CT * R * sizeof(float);
    h->t16_block = block;
    h->t16_nblocks = (n16 + block - 1) / block;
    std::vector<DwMacro> mac;
    build_dw_macros(h, n_rec, mac);
    h->t16_rows = h->n_tiles + h->t16_nblocks * h->dw_slices;
    const int stride = m.n_params + 8;
    hipError_t e = hipMalloc((void**)&h->d_dwtape, need);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_macros, mac.size() * sizeof(DwMacro));
    if (e == hipSuccess) e = hipMemcpy(h->d_macros, mac.data(), mac.size() * sizeof(DwMacro), hipMemcpyHostToDevice);
    if (e == hipSuccess && !h->d_slab) e = hipMalloc((void**)&h->d_slab, (size_t)h->t16_rows * stride * sizeof(float));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (h->d_dwtape) { (void)hipFree(h->d_dwtape); h->d_dwtape = nullptr; }
        h->t16_dwtape = 0;
        if (h->ag_rows) return fail("allocating the delta tape (%zu B) failed: %s", need, hipGetErrorString(e));
        return 0;
    }
    h->t16_dwtape = 1;
    // the block's stage tape, and (COLNDE_T16_ZTAPE=0 disables) the hidden pre-activations the forward kernel tapes for the adjoint
    if (!h->d_tape) {
        e = hipMalloc((void**)&h->d_tape, n_rec * CT * m.ns * sizeof(float));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(h->d_dwtape); h->d_dwtape = nullptr;
            h->t16_dwtape = 0;
            if (h->ag_rows) return fail("allocating the stage tape failed: %s", hipGetErrorString(e));
            return 0;
        }
    }
    const char* ezt = getenv("COLNDE_T16_ZTAPE");
    // net-split kernels on a small block (<= 2,048 columns, where it pays: 64-step iteration with the four-wave kernels 2.36 vs 2.48 ms at 1,024
    // columns, 2.69 vs 2.77 at 2,048, 3.99 vs 3.53 at 4,096): the rich tape in place of the pre-activations
    {
        if (want_rich && hipMalloc((void**)&h->d_t16_ztape, n_rec * rt_split_rich_record_floats() * sizeof(float)) == hipSuccess) {
            h->split_rich = true;
            return 0;
        }
        (void)hipGetLastError();
        h->d_t16_ztape = nullptr;
    }
    if (!(ezt && atoi(ezt) == 0) && hipMalloc((void**)&h->d_t16_ztape, n_rec * CT * t16_ztape_col_floats(m) * sizeof(float)) != hipSuccess) {
        (void)hipGetLastError();
        h->d_t16_ztape = nullptr;
    }
    if ((h->ag_rows || need_z) && !h->d_t16_ztape)
        return fail("this network's taped adjoint needs the pre-activation tape (COLNDE_T16_ZTAPE=0 or its allocation failed): no gradient path for it without one");
    return 0;
}

extern "C" int colnde_loss_grad_dev(colnde_handle* h, const float* d_weights, const float scalings[6], float* d_out) {
    if (!h) return fail("null handle");
    if (!d_weights || !scalings || !d_out) return fail("null pointer argument");
    if (!h->have_truth) return fail("no truth trajectories: pass truth to colnde_set_problem");
    if (h->m.inplace) return fail("the in-place NDE! variant is an evaluation RHS; gradients use the training RHS (inplace_variant = 0)");
    HIPCHK(hipSetDevice(h->device));
    if (h->auto_substeps) {
        if (!h->have_problem) return fail("colnde_set_problem has not been called");
        if (refuse_auto_on_a_shard(h)) return 1;
        h->auto_substeps = false;
        if (choose_substeps_impl(h, d_weights, h->cfg.reltol, nullptr, nullptr)) { h->auto_substeps = true; return 1; }
    }
    if (check_stability(h)) return 1;
    const int stride = h->m.n_params + 8;
    if (h->use_rt) {
        if (!h->have_problem) return fail("colnde_set_problem has not been called");
        if (rt_plan_tapes(h)) return 1;
        const int n_steps = (h->cfg.n_save - 1) * h->cfg.substeps;
        const int n_wt = rt_n_wtiles(h->n_col), n_dw = rt_dw1_waves(h->rt_block, n_steps);
        if (!h->d_rt_slab) {
            h->rt_rows = n_wt + h->rt_nblocks * n_dw;
            hipError_t e = hipMalloc((void**)&h->d_rt_slab, (size_t)h->rt_rows * stride * sizeof(float));
            if (e != hipSuccess) return fail("hipMalloc of the partial-gradient slab failed: %s", hipGetErrorString(e));
        }
        LossWeights lw;
        loss_weights(h, scalings, &lw);
        hipError_t e = rt_launch_pack(h->m, d_weights, h->d_wimg, h->stream);
        if (e != hipSuccess) return fail("rt pack launch failed: %s", hipGetErrorString(e));
        HIPCHK(hipMemsetAsync(h->d_rt_slab, 0, (size_t)h->rt_rows * stride * sizeof(float), h->stream));
        const size_t ns = h->m.ns;
        for (int b = 0; b < h->rt_nblocks; b++) {
            const int c0 = b * h->rt_block, nc = std::min(h->rt_block, h->n_col - c0);
            if (nc <= 0) break;
            if (rt_forward_range(h, h->d_sol, true, c0, nc)) return 1;
            {
                Timed tm(h, K_ADJOINT);
                e = rt_launch_adjoint(h->m, h->d_wimg, h->d_bcs + (size_t)c0 * h->m.n_bc, h->d_times, h->cfg.n_save, h->cfg.substeps,
                                      h->d_sol + (size_t)c0 * h->cfg.n_save * ns, h->d_truth + (size_t)c0 * h->cfg.n_save * ns, h->d_rt_tape,
                                      h->d_rt_tape2, h->rt_ztape ? h->d_rt_tapez : nullptr, lw,
                                      h->d_rt_slab + (size_t)(c0 / 32) * stride, nc, h->sp_adj, h->stream);
                if (e != hipSuccess) return fail("rt adjoint launch failed: %s", hipGetErrorString(e));
            }
            {
                Timed tm(h, K_DW1);
                e = rt_launch_dw1(h->m, h->d_rt_tape, h->d_rt_tape2, nc, n_steps, h->d_rt_slab + ((size_t)n_wt + (size_t)b * n_dw) * stride,
                                  h->sp_dw, h->stream);
                if (e != hipSuccess) return fail("rt dW1 launch failed: %s", hipGetErrorString(e));
            }
        }
        {
            Timed tm(h, K_REDUCE);
            e = launch_reduce(h->d_rt_slab, h->rt_rows, h->m.n_params, stride, lw, d_out, h->stream);
            if (e != hipSuccess) return fail("reduce launch failed: %s", hipGetErrorString(e));
        }
        return 0;
    }
    if (h->use_fc) {
        if (!h->have_problem) return fail("colnde_set_problem has not been called");
        if (fc_plan_tapes(h)) return 1;
        LossWeights lw;
        loss_weights(h, scalings, &lw);
        const size_t ns = h->m.ns;
        hipError_t e = fc_launch_pack(h->m, h->fc_cw, d_weights, h->d_fc_imgf, h->d_fc_imgb, h->d_fc_bias, h->d_fc_simgf, h->d_fc_simgb, h->stream);
        if (e != hipSuccess) return fail("fc32 pack launch failed: %s", hipGetErrorString(e));
        HIPCHK(hipMemsetAsync(h->d_slab, 0, (size_t)h->fc_rows * stride * sizeof(float), h->stream));
        const int cw = h->fc_cw;
        const int n_wg = (h->n_col + cw - 1) / cw, n_iv = h->cfg.n_save - 1, nseg = h->fc_nseg;
        const size_t gemm_rows0 = (size_t)n_wg * nseg;                      // slab: [tile][segment] adjoint rows, then [block][segment][slice] GEMM rows
        for (int b = 0; b < h->fc_nblocks; b++) {
            const int c0 = b * h->fc_block, nc = std::min(h->fc_block, h->n_col - c0);
            if (nc <= 0) break;
            const size_t tiles_b = ((size_t)nc + cw - 1) / cw;
            // time segments: the states at the save points first (tape-less), then segment by segment from the end of the axis
            // (that first pass tapes the LAST segment on its way, which is the first one the backward sweep needs)
            if (nseg > 1 && fc_forward_range(h, h->d_sol, true, c0, nc, 0, n_iv, (nseg - 1) * h->fc_seg)) return 1;
            for (int sg = nseg - 1; sg >= 0; sg--) {
                const int iv0 = sg * h->fc_seg, iv1 = std::min(n_iv, iv0 + h->fc_seg);
                if (!(nseg > 1 && sg == nseg - 1) && fc_forward_range(h, h->d_sol, true, c0, nc, iv0, iv1)) return 1;
                {
                    Timed tm(h, K_ADJOINT);
                    e = fc_launch_adjoint(h->m, h->fc_cw, h->d_fc_imgb, (h->sp_adj && fc_split_supported(h->fc_cw)) ? h->d_fc_simgb : nullptr, h->d_times, h->cfg.n_save, iv0, iv1, h->cfg.substeps, h->d_sol + (size_t)c0 * h->cfg.n_save * ns,
                                          h->d_truth + (size_t)c0 * h->cfg.n_save * ns, h->d_dwtape, h->d_fc_masks, h->d_fc_switch, lw.w[2],
                                          nseg > 1 ? h->d_fc_lam + (size_t)c0 * h->m.Nz : nullptr,
                                          h->d_slab + ((size_t)sg * n_wg + (size_t)(c0 / cw)) * stride, nc, h->stream);
                    if (e != hipSuccess) return fail("fc32 adjoint launch failed: %s", hipGetErrorString(e));
                }
                {
                    Timed tm(h, K_DW1);
                    if (h->sp_dw && !h->dw_split.passes.empty())
                        e = launch_dw_gemm_split(h->d_dwtape, tiles_b * (cw / 16) * (size_t)(iv1 - iv0) * h->cfg.substeps * h->m.nst, (int)dwtape_row_floats(h->m), h->dw_split,
                                                 h->dw_slices, h->d_slab + (gemm_rows0 + ((size_t)b * nseg + sg) * h->dw_slices) * stride, stride, h->stream);
                    else
                    e = launch_dw_gemm(h->d_dwtape, tiles_b * (cw / 16) * (size_t)(iv1 - iv0) * h->cfg.substeps * h->m.nst, (int)dwtape_row_floats(h->m), h->d_macros,
                                       h->n_macros, h->dw_slices, h->d_slab + (gemm_rows0 + ((size_t)b * nseg + sg) * h->dw_slices) * stride, stride, h->stream);
                    if (e != hipSuccess) return fail("dW GEMM launch failed: %s", hipGetErrorString(e));
                }
            }
        }
        {
            Timed tm(h, K_REDUCE);
            e = launch_reduce(h->d_slab, h->fc_rows, h->m.n_params, stride, lw, d_out, h->stream);
            if (e != hipSuccess) return fail("reduce launch failed: %s", hipGetErrorString(e));
        }
        return 0;
    }
    if (t16_plan_dwtape(h)) return 1;
    if (h->t16_dwtape == 1) {
        if (!h->have_problem) return fail("colnde_set_problem has not been called");
        LossWeights lw;
        loss_weights(h, scalings, &lw);
        const int n_steps = (h->cfg.n_save - 1) * h->cfg.substeps;
        const size_t ns = h->m.ns;
        if (pack(h, d_weights)) return 1;
        HIPCHK(hipMemsetAsync(h->d_slab, 0, (size_t)h->t16_rows * stride * sizeof(float), h->stream));
        for (int b = 0; b < h->t16_nblocks; b++) {
            const int c0 = b * h->t16_block, nc = std::min(h->t16_block, h->n_col - c0);
            if (nc <= 0) break;
            const int tiles_b = (nc + CT - 1) / CT;
            if (t16_forward_range(h, d_weights, h->d_sol, true, c0, nc)) return 1;
            {
                Timed tm(h, K_ADJOINT);
                AdjointGeom g = {512, 1, 3, 0};
                hipError_t e;
                if (h->adj_split && h->d_t16_ztape)
                    // the companion of the split forward: one wavefront per flux net (+ a helper) per tile, writing tile16's delta tape
                    e = rt_launch_adjoint_split(h->m, h->d_wimg, h->d_times, h->cfg.n_save, h->cfg.substeps,
                                                h->d_sol + (size_t)c0 * h->cfg.n_save * ns, h->d_truth + (size_t)c0 * h->cfg.n_save * ns, h->d_tape,
                                                h->d_t16_ztape, lw, h->d_slab + (size_t)(c0 / CT) * stride, nc, h->d_dwtape, h->split_rich, h->adj_helper, h->sp_adj, h->stream);
                else
                e = launch_adjoint(h->m, h->pk, d_weights, h->d_wf, h->d_wb, h->d_tiles, h->d_bias_zoff, h->d_bias_goff,
                                              h->d_bcs + (size_t)c0 * h->m.n_bc, h->d_times, h->cfg.n_save, h->cfg.substeps,
                                              h->d_sol + (size_t)c0 * h->cfg.n_save * ns, h->d_truth + (size_t)c0 * h->cfg.n_save * ns, h->d_tape,
                                              lw, h->d_slab + (size_t)(c0 / CT) * stride, nc, g,
                                              (MODEL_FLOATS + (h->ag_rows ? lds_floats_adjoint_ag(h->m) : (h->d_t16_ztape ? lds_floats_adjoint_noA(h->m) : lds_floats_adjoint(h->m)))) * sizeof(float),
                                              h->stream, h->d_dwtape, h->d_t16_ztape);
                if (e != hipSuccess) return fail("adjoint (taped dW) launch failed: %s", hipGetErrorString(e));
            }
            {
                Timed tm(h, K_DW1);
                hipError_t e = (h->sp_dw && !h->dw_split.passes.empty())
                    ? launch_dw_gemm_split(h->d_dwtape, (size_t)tiles_b * n_steps * h->m.nst, (int)dwtape_row_floats(h->m), h->dw_split,
                                           h->dw_slices, h->d_slab + ((size_t)h->n_tiles + (size_t)b * h->dw_slices) * stride, stride, h->stream)
                    : launch_dw_gemm(h->d_dwtape, (size_t)tiles_b * n_steps * h->m.nst, (int)dwtape_row_floats(h->m), h->d_macros, h->n_macros,
                                              h->dw_slices, h->d_slab + ((size_t)h->n_tiles + (size_t)b * h->dw_slices) * stride, stride, h->stream);
                if (e != hipSuccess) return fail("dW GEMM launch failed: %s", hipGetErrorString(e));
            }
        }
        {
            Timed tm(h, K_REDUCE);
            hipError_t e = launch_reduce(h->d_slab, h->t16_rows, h->m.n_params, stride, lw, d_out, h->stream);
            if (e != hipSuccess) return fail("reduce launch failed: %s", hipGetErrorString(e));
        }
        return 0;
    }
    if (forward_impl(h, d_weights, h->d_sol, true)) return 1;
    if (!h->geo_ok)
        return fail("network too large for the tile engine's in-register adjoint (%d weight-gradient tiles, %zu B of LDS) and its "
                    "delta tape does not fit in HBM (or COLNDE_T16_DWTAPE=0)", h->m.n_tiles, h->lds_adj);
    if (!h->d_slab) {
        hipError_t e = hipMalloc((void**)&h->d_slab, (size_t)h->n_tiles * stride * sizeof(float));
        if (e != hipSuccess) return fail("hipMalloc of the partial-gradient slab failed: %s", hipGetErrorString(e));
    }
    LossWeights lw;
    loss_weights(h, scalings, &lw);
    {
        Timed tm(h, K_ADJOINT);
        hipError_t e = launch_adjoint(h->m, h->pk, d_weights, h->d_wf, h->d_wb, h->d_tiles, h->d_bias_zoff, h->d_bias_goff,
                                      h->d_bcs, h->d_times, h->cfg.n_save, h->cfg.substeps, h->d_sol, h->d_truth, h->d_tape,
                                      lw, h->d_slab, h->n_col, h->geo, h->lds_adj, h->stream);
        if (e != hipSuccess) return fail("adjoint launch failed: %s", hipGetErrorString(e));
    }
    {
        Timed tm(h, K_REDUCE);
        hipError_t e = launch_reduce(h->d_slab, h->n_tiles, h->m.n_params, stride, lw, d_out, h->stream);
        if (e != hipSuccess) return fail("reduce launch failed: %s", hipGetErrorString(e));
    }
    return 0;
}

extern "C" int colnde_loss_grad(colnde_handle* h, const float* weights, const float scalings[6], float terms[6],
                                float* total, float* grad) {
    if (!h) return fail("null handle");
    if (!weights || !scalings || !terms || !total || !grad) return fail("null pointer argument");
    HIPCHK(hipSetDevice(h->device));
    const int np = h->m.n_params;
    HIPCHK(hipMemcpyAsync(h->d_w, weights, sizeof(float) * np, hipMemcpyHostToDevice, h->stream));
    if (colnde_loss_grad_dev(h, h->d_w, scalings, h->d_out)) return 1;
    float o[8];
    HIPCHK(hipMemcpyAsync(grad, h->d_out, sizeof(float) * np, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(o, h->d_out + np, sizeof(o), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int q = 0; q < 6; q++) terms[q] = o[q];
    *total = o[6];
    if (!std::isfinite(o[6])) return fail("the loss is not finite (%g): the solve left the stable regime (time step, weights or inputs)", o[6]);
    return 0;
}

// ---- error-controlled time stepping: what `reltol` means at this boundary ------------------------------------------------------------
// The reference hands the solve to an adaptive integrator (solve(prob, ROCK4(); reltol=1f-3, ...): NDE_training.jl:291,304,403; reltol=1e-4 at
// free_convection/src/solve.jl:4).  This path steps at a fixed `substeps` per save interval, so the tolerance is enforced a posteriori: a second
// forward solve at 2 x substeps and Richardson's estimate of the error of the first, e = (u_S - u_2S) 2^p / (2^p - 1) (p = 4: RK4, 2: RKC2), in
// the integrator's own norm: per column and save point the RMS over the state's components of e_i / (abstol / reltol + |u_i|), abstol / reltol = 1e-3
// (OrdinaryDiffEq's default abstol 1e-6 over the reference's reltol; its accept test is rms(err / (abstol + reltol |u|)) <= 1), then the maximum over
// columns and save points.
static int forward_at(colnde_handle* h, const float* d_weights, float* d_sol, int substeps) {
    const int keep = h->cfg.substeps;
    h->cfg.substeps = substeps;
    const bool au = h->auto_substeps;
    h->auto_substeps = false;
    int rc = refresh_rkc(h);
    if (!rc) rc = forward_impl(h, d_weights, d_sol, false);
    h->cfg.substeps = keep;
    h->auto_substeps = au;
    if (refresh_rkc(h)) rc = 1;
    return rc;
}

static int tapes_planned(const colnde_handle* h) { return h->d_rt_tape || h->d_dwtape || h->d_tape; }

// estimate at `substeps` given its solution d_a; d_b receives the solution at 2 x substeps; *d_est (device) the estimate
static int estimate_from(colnde_handle* h, const float* d_weights, const float* d_a, float* d_b, int substeps, float* d_est, float floor_) {
    if (forward_at(h, d_weights, d_b, 2 * substeps)) return 1;
    hipError_t e = launch_rel_diff_max(d_a, d_b, (long)h->n_col * h->cfg.n_save, h->m.ns, floor_, h->d_partial, d_est, h->stream);
    if (e != hipSuccess) return fail("error-estimate launch failed: %s", hipGetErrorString(e));
    return 0;
}

// 2^p / (2^p - 1) with the order the right-hand side lets the stepper reach: a SWITCHING right-hand side (ConvectiveAdjustmentNDE's min(0, K dT/dz),
// the wind-mixing convective-adjustment branch) converges at about first order through its kinks whatever the stepper's nominal order (measured:
// profiles/r05_rkc2_conditioning.json), so the conservative factor 2 is used there (p = 4 / 2 understated the error by up to 2x);
// smooth closures keep p = 4 (RK4) and p = 2 (RKC2).
static float richardson_factor(const colnde_handle* h) {
    if (h->m.ca || h->cfg.model == COLNDE_MODEL_CONV_ADJ_NDE) return 2.0f;
    return h->cfg.stepper == COLNDE_STEPPER_RKC2 ? 4.0f / 3.0f : 16.0f / 15.0f;
}
// abstol / reltol of the integrator's accept test rms(err / (abstol + reltol |u|)) <= 1, OrdinaryDiffEq's default abstol = 1e-6: 1e-3 at the wind-mixing
// reltol = 1e-3 (NDE_training.jl:291), 1e-2 at free convection's 1e-4 (solve.jl:4)
static float norm_floor(float reltol) { return 1e-6f / (reltol > 0.0f ? reltol : 1e-3f); }

extern "C" int colnde_error_estimate_dev(colnde_handle* h, const float* d_weights, float* max_rel_err) {
    if (!h) return fail("null handle");
    if (!d_weights || !max_rel_err) return fail("null pointer argument");
    if (!h->have_problem) return fail("colnde_set_problem has not been called");
    HIPCHK(hipSetDevice(h->device));
    const size_t n = (size_t)h->n_col * h->cfg.n_save * h->m.ns;
    float* d_b = nullptr;
    HIPCHK(hipMalloc((void**)&d_b, (n + 4) * sizeof(float)));
    int rc = forward_at(h, d_weights, h->d_sol, h->cfg.substeps);
    if (!rc) rc = estimate_from(h, d_weights, h->d_sol, d_b, h->cfg.substeps, d_b + n, norm_floor(h->cfg.reltol));
    float est = 0.0f;
    if (!rc && (hipMemcpyAsync(&est, d_b + n, sizeof(float), hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess))
        rc = fail("error estimate: device-to-host copy failed");
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(d_b);
    if (rc) return 1;
    *max_rel_err = est * richardson_factor(h);
    return 0;
}

extern "C" int colnde_error_estimate(colnde_handle* h, const float* weights, float* max_rel_err) {
    if (!h) return fail("null handle");
    if (!weights || !max_rel_err) return fail("null pointer argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(h->d_w, weights, sizeof(float) * h->m.n_params, hipMemcpyHostToDevice, h->stream));
    return colnde_error_estimate_dev(h, h->d_w, max_rel_err);
}

// The least power-of-two sub-step count (not below the stability bound) whose estimate meets reltol; the handle keeps it.
static int choose_substeps_impl(colnde_handle* h, const float* d_weights, float reltol, int* chosen, float* estimate) {
    if (tapes_planned(h)) return fail("the sub-step count sizes the tapes: choose it before the first colnde_loss_grad of this handle");
    if (!(reltol > 0.0f)) return fail("reltol must be > 0");
    const size_t n = (size_t)h->n_col * h->cfg.n_save * h->m.ns;
    float* d_b = nullptr;
    HIPCHK(hipMalloc((void**)&d_b, (2 * n + 4) * sizeof(float)));
    float* buf[2] = {d_b, d_b + n};
    int S = 1;
    while (S < h->min_substeps) S *= 2;
    int rc = forward_at(h, d_weights, buf[0], S);
    float est = -1.0f, prev = -1.0f;
    int cur = 0;
    bool floor_hit = false;
    while (!rc) {
        rc = estimate_from(h, d_weights, buf[cur], buf[cur ^ 1], S, d_b + 2 * n, norm_floor(reltol));
        if (rc) break;
        float e = 0.0f;
        if (hipMemcpyAsync(&e, d_b + 2 * n, sizeof(float), hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) {
            rc = fail("choose_substeps: device-to-host copy failed");
            break;
        }
        est = e * richardson_factor(h);
        if (!(est == est) || est > 3.0e38f) { rc = fail("the solve is not finite at %d sub-steps per save interval: no sub-step count can be chosen", S); break; }
        if (est <= reltol || S >= 4096) break;
        // halving a fourth-order step divides the error by ~16 (second order: 4; a switch in the right-hand side — ConvectiveAdjustmentNDE — makes it ~2);
        // an estimate that falls by less than a fifth is float32 round-off of the two solves it compares (more steps only add to it): the tolerance is
        // below what this arithmetic resolves
        if (prev > 0.0f && est > 0.8f * prev) { floor_hit = true; break; }
        prev = est;
        S *= 2;                      // the solution at 2S is already there: it is the next candidate's own
        cur ^= 1;
    }
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(d_b);
    if (rc) return 1;
    if (floor_hit) return fail("reltol = %g is below the float32 round-off floor of this solve: the estimate stopped falling at %g (%d sub-steps per save interval, "
                               "%g at half as many)", reltol, est, S, prev);
    if (est > reltol) return fail("reltol = %g is not met with %d sub-steps per save interval (estimate %g): the right-hand side is too stiff for this stepper", reltol, S, est);
    h->cfg.substeps = S;
    if (refresh_rkc(h)) return 1;
    h->last_estimate = est;
    h->substeps_chosen = true;
    if (chosen) *chosen = S;
    if (estimate) *estimate = est;
    return 0;
}

extern "C" int colnde_choose_substeps(colnde_handle* h, const float* weights, float reltol, int* substeps, float* estimate) {
    if (!h) return fail("null handle");
    if (!weights) return fail("null weights");
    if (!h->have_problem) return fail("colnde_set_problem has not been called");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(h->d_w, weights, sizeof(float) * h->m.n_params, hipMemcpyHostToDevice, h->stream));
    const bool au = h->auto_substeps;
    h->auto_substeps = false;
    const int rc = choose_substeps_impl(h, h->d_w, reltol > 0.0f ? reltol : h->cfg.reltol, substeps, estimate);
    if (rc) h->auto_substeps = au;
    return rc;
}

extern "C" int colnde_substeps(const colnde_handle* h) { return h ? h->cfg.substeps : -1; }

// Impose a sub-step count (e.g. the MAX over ranks of what colnde_choose_substeps chose on each shard).  Refused once the tapes are planned (the count
// sizes them) and outside the stepper's stability bound; clears a pending substeps = 0.
extern "C" int colnde_set_substeps(colnde_handle* h, int substeps) {
    if (!h) return fail("null handle");
    if (substeps < 1) return fail("substeps must be >= 1");
    if (substeps == h->cfg.substeps && !h->auto_substeps) return 0;
    if (tapes_planned(h)) return fail("the sub-step count sizes the tapes: set it before the first colnde_loss_grad of this handle");
    colnde_config c = h->cfg;
    c.substeps = substeps;
    const int need = colnde_min_substeps(&c);
    if (need < 0) return 1;
    if (substeps < need && !getenv("COLNDE_ALLOW_UNSTABLE_DT"))
        return fail("substeps = %d is outside the stepper's stability bound: need >= %d (colnde_min_substeps)", substeps, need);
    h->cfg.substeps = substeps;
    h->auto_substeps = false;
    h->substeps_chosen = false;
    h->last_estimate = -1.0f;
    return refresh_rkc(h);
}

// ---- flux diagnostics: predict_flux and loss_per_tstep ---------------------------------------------------------------------------------
extern "C" int colnde_flux_dev(colnde_handle* h, const float* d_x, const float* d_weights, const float* d_bcs, float t, float* d_flux, int n_columns) {
    if (!h) return fail("null handle");
    if (!d_x || !d_weights || !d_bcs || !d_flux) return fail("null pointer argument");
    if (n_columns < 1) return fail("n_columns must be >= 1");
    HIPCHK(hipSetDevice(h->device));
    if (pack(h, d_weights)) return 1;
    if (ensure_ag(h, ((size_t)n_columns + CT - 1) / CT)) return 1;
    Timed tm(h, K_RHS);
    hipError_t e = launch_rhs(h->m, h->pk, d_weights, h->d_wf, d_x, d_bcs, t, nullptr, n_columns, 256, h->lds_fwd, h->stream, d_flux);
    if (e != hipSuccess) return fail("flux launch failed: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int colnde_flux(colnde_handle* h, const float* x, const float* weights, const float* bcs, float t, float* flux, int n_columns) {
    if (!h) return fail("null handle");
    if (!x || !weights || !bcs || !flux) return fail("null pointer argument");
    if (n_columns < 1) return fail("n_columns must be >= 1");
    HIPCHK(hipSetDevice(h->device));
    if (ensure_tmp(h, (size_t)n_columns)) return 1;
    const DevModel& m = h->m;
    const size_t nfl = (size_t)n_columns * m.n_nets * (m.Nz + 1);
    float* d_fl = nullptr;
    HIPCHK(hipMalloc((void**)&d_fl, nfl * sizeof(float)));
    int rc = 1;
    do {
        if (hipMemcpyAsync(h->d_w, weights, sizeof(float) * m.n_params, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipMemcpyAsync(h->d_tmp_a, x, sizeof(float) * (size_t)n_columns * m.ns, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipMemcpyAsync(h->d_tmp_b, bcs, sizeof(float) * (size_t)n_columns * m.n_bc, hipMemcpyHostToDevice, h->stream) != hipSuccess) {
            fail("flux: host-to-device copy failed");
            break;
        }
        if (colnde_flux_dev(h, h->d_tmp_a, h->d_w, h->d_tmp_b, t, d_fl, n_columns)) break;
        if (hipMemcpyAsync(flux, d_fl, nfl * sizeof(float), hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) {
            fail("flux: device-to-host copy failed");
            break;
        }
        rc = 0;
    } while (0);
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(d_fl);
    return rc;
}

extern "C" int colnde_loss_per_tstep_dev(colnde_handle* h, const float* d_weights, float* d_out) {
    if (!h) return fail("null handle");
    if (!d_weights || !d_out) return fail("null pointer argument");
    if (!h->have_truth) return fail("no truth trajectories: pass truth to colnde_set_problem");
    HIPCHK(hipSetDevice(h->device));
    if (forward_impl(h, d_weights, h->d_sol, false)) return 1;
    hipError_t e = launch_loss_per_tstep(h->d_sol, h->d_truth, h->n_col, h->cfg.n_save, h->m.Nz, h->m.ns / h->m.Nz, d_out, h->stream);
    if (e != hipSuccess) return fail("loss_per_tstep launch failed: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int colnde_loss_per_tstep(colnde_handle* h, const float* weights, float* out) {
    if (!h) return fail("null handle");
    if (!weights || !out) return fail("null pointer argument");
    HIPCHK(hipSetDevice(h->device));
    const size_t n = (size_t)h->n_col * 6 * h->cfg.n_save;
    float* d_o = nullptr;
    HIPCHK(hipMalloc((void**)&d_o, n * sizeof(float)));
    int rc = 1;
    do {
        if (hipMemcpyAsync(h->d_w, weights, sizeof(float) * h->m.n_params, hipMemcpyHostToDevice, h->stream) != hipSuccess) { fail("loss_per_tstep: host-to-device copy failed"); break; }
        if (colnde_loss_per_tstep_dev(h, h->d_w, d_o)) break;
        if (hipMemcpyAsync(out, d_o, n * sizeof(float), hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) {
            fail("loss_per_tstep: device-to-host copy failed");
            break;
        }
        rc = 0;
    } while (0);
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(d_o);
    return rc;
}

// ---- embedded inference --------------------------------------------------------------------------------
// sign = +1: the forcing -dz(wT); -1: +dz(wT), what the reference stores in params.∂z_wT_NN (double_gyre_nn.jl:165)
static int infer_impl(colnde_handle* h, const float* d_weights, const float* d_T, const float* d_top_flux, float Lz, float* d_out, int n_columns, float sign);
extern "C" int colnde_infer_forcing_dev(colnde_handle* h, const float* d_weights, const float* d_T, const float* d_top_flux,
                                        float Lz, float* d_out, int n_columns) {
    return infer_impl(h, d_weights, d_T, d_top_flux, Lz, d_out, n_columns, 1.0f);
}
extern "C" int colnde_infer_dz_wT_dev(colnde_handle* h, const float* d_weights, const float* d_T, const float* d_top_flux,
                                      float Lz, float* d_out, int n_columns) {
    return infer_impl(h, d_weights, d_T, d_top_flux, Lz, d_out, n_columns, -1.0f);
}
static int infer_impl(colnde_handle* h, const float* d_weights, const float* d_T, const float* d_top_flux, float Lz, float* d_out, int n_columns, float sign) {
    if (!h) return fail("null handle");
    if (!d_weights || !d_T || !d_top_flux || !d_out) return fail("null pointer argument");
    if (h->m.model == COLNDE_MODEL_WIND_MIXING) return fail("infer_forcing needs a single T-only network (free-convection model)");
    if (n_columns < 1 || !(Lz > 0.0f)) return fail("n_columns >= 1 and Lz > 0 required");
    HIPCHK(hipSetDevice(h->device));
    if (h->use_fc && h->m.model == COLNDE_MODEL_FREE_CONVECTION) {
        // the reference's forcing network IS the fc32 shape (32-128-128-31 in double_gyre_nn.jl): the 32-column engine's sections, one evaluation
        const int cw = fc_tile_width(n_columns);                 // (the images are packed per call: this call's own tile width)
        hipError_t ef = fc_launch_pack(h->m, cw, d_weights, h->d_fc_imgf, h->d_fc_imgb, h->d_fc_bias, nullptr, nullptr, h->stream);
        if (ef != hipSuccess) return fail("fc32 pack launch failed: %s", hipGetErrorString(ef));
        Timed tm(h, K_INFER);
        ef = fc_launch_infer(h->m, cw, h->d_fc_imgf, h->d_fc_bias, d_T, d_top_flux, sign * (float)h->m.Nz / Lz, d_out, n_columns, h->stream);
        if (ef != hipSuccess) return fail("fc32 infer launch failed: %s", hipGetErrorString(ef));
        return 0;
    }
    if (pack(h, d_weights)) return 1;
    Timed tm(h, K_INFER);
    hipError_t e = launch_infer(h->m, h->pk, d_weights, h->d_wf, d_T, d_top_flux, sign * (float)h->m.Nz / Lz, d_out, n_columns, 256,
                                h->lds_fwd, h->stream);
    if (e != hipSuccess) return fail("infer launch failed: %s", hipGetErrorString(e));
    return 0;
}

static int infer_host(colnde_handle* h, const float* weights, const float* T, const float* top_flux, float Lz, float* out, int n_columns, float sign);
extern "C" int colnde_infer_forcing(colnde_handle* h, const float* weights, const float* T, const float* top_flux, float Lz,
                                    float* out, int n_columns) {
    return infer_host(h, weights, T, top_flux, Lz, out, n_columns, 1.0f);
}
extern "C" int colnde_infer_dz_wT(colnde_handle* h, const float* weights, const float* T, const float* top_flux, float Lz,
                                  float* out, int n_columns) {
    return infer_host(h, weights, T, top_flux, Lz, out, n_columns, -1.0f);
}
static int infer_host(colnde_handle* h, const float* weights, const float* T, const float* top_flux, float Lz, float* out, int n_columns, float sign) {
    if (!h) return fail("null handle");
    if (!weights || !T || !top_flux || !out) return fail("null pointer argument");
    if (n_columns < 1) return fail("n_columns must be >= 1");
    HIPCHK(hipSetDevice(h->device));
    if (ensure_tmp(h, (size_t)n_columns)) return 1;
    const int Nz = h->m.Nz;
    HIPCHK(hipMemcpyAsync(h->d_w, weights, sizeof(float) * h->m.n_params, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_tmp_a, T, sizeof(float) * (size_t)n_columns * Nz, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_tmp_b, top_flux, sizeof(float) * (size_t)n_columns, hipMemcpyHostToDevice, h->stream));
    if (infer_impl(h, h->d_w, h->d_tmp_a, h->d_tmp_b, Lz, h->d_tmp_c, n_columns, sign)) return 1;
    HIPCHK(hipMemcpyAsync(out, h->d_tmp_c, sizeof(float) * (size_t)n_columns * Nz, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// ---- the steps either side of the hot path ---------------------------------------------------
extern "C" int colnde_convective_adjustment_dev(colnde_handle* h, const float* d_T, const float* d_halo_bottom,
                                                const float* d_halo_top, float dt, float dz, float K, float* d_out, int n_columns) {
    if (!h) return fail("null handle");
    if (!d_T || !d_out) return fail("null pointer argument");
    if (n_columns < 1) return fail("n_columns must be >= 1");
    if (!(dt > 0.0f) || !(dz > 0.0f) || !(K >= 0.0f)) return fail("dt > 0, dz > 0 and K >= 0 required");
    if (h->m.Nz < 2 || h->m.Nz > 128) return fail("convective adjustment supports 2 <= Nz <= 128 (Nz = %d)", h->m.Nz);
    HIPCHK(hipSetDevice(h->device));
    Timed tm(h, K_CONVADJ);
    hipError_t e = launch_convective_adjustment(d_T, d_halo_bottom, d_halo_top, dt / (dz * dz), K, d_out, h->m.Nz, n_columns, h->stream);
    if (e != hipSuccess) return fail("convective adjustment launch failed: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int colnde_convective_adjustment(colnde_handle* h, const float* T, const float* halo_bottom, const float* halo_top,
                                            float dt, float dz, float K, float* out, int n_columns) {
    if (!h) return fail("null handle");
    if (!T || !out) return fail("null pointer argument");
    if (n_columns < 1) return fail("n_columns must be >= 1");
    HIPCHK(hipSetDevice(h->device));
    if (ensure_tmp(h, (size_t)n_columns)) return 1;
    const int Nz = h->m.Nz;
    HIPCHK(hipMemcpyAsync(h->d_tmp_a, T, sizeof(float) * (size_t)n_columns * Nz, hipMemcpyHostToDevice, h->stream));
    float* d_hb = nullptr;
    float* d_ht = nullptr;
    if (halo_bottom) {
        d_hb = h->d_tmp_b;
        HIPCHK(hipMemcpyAsync(d_hb, halo_bottom, sizeof(float) * (size_t)n_columns, hipMemcpyHostToDevice, h->stream));
    }
    if (halo_top) {
        d_ht = h->d_tmp_b + n_columns;
        HIPCHK(hipMemcpyAsync(d_ht, halo_top, sizeof(float) * (size_t)n_columns, hipMemcpyHostToDevice, h->stream));
    }
    if (colnde_convective_adjustment_dev(h, h->d_tmp_a, d_hb, d_ht, dt, dz, K, h->d_tmp_c, n_columns)) return 1;
    HIPCHK(hipMemcpyAsync(out, h->d_tmp_c, sizeof(float) * (size_t)n_columns * Nz, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// modified_pacanowski_philander! (wind_mixing/src/NDE_oceananigans.jl:61-101): one implicit diffusion step of u, v, T per column
static int impl_diff_check(colnde_handle* h, const void* a, const void* b, const void* c, const void* d, const void* e, const void* f,
                           float dt, float dz, const float params[7], int n_columns) {
    if (!h) return fail("null handle");
    if (!a || !b || !c || !d || !e || !f || !params) return fail("null pointer argument");
    if (n_columns < 1) return fail("n_columns must be >= 1");
    if (!(dt > 0.0f) || !(dz > 0.0f)) return fail("dt > 0 and dz > 0 required");
    if (!(params[0] >= 0.0f) || !(params[1] >= 0.0f)) return fail("nu0 >= 0 and nu_minus >= 0 required (the tridiagonal must stay diagonally dominant)");
    if (!(params[2] != 0.0f) || !(params[4] > 0.0f)) return fail("dRi != 0 and Pr > 0 required");
    if (h->m.Nz < 2 || h->m.Nz > 128) return fail("implicit diffusion supports 2 <= Nz <= 128 (Nz = %d)", h->m.Nz);
    return 0;
}

extern "C" int colnde_implicit_diffusion_dev(colnde_handle* h, const float* d_u, const float* d_v, const float* d_T,
                                             const float* d_halo_bottom, float dt, float dz, const float params[7],
                                             int convective_adjustment, float* d_u_out, float* d_v_out, float* d_T_out, int n_columns) {
    if (impl_diff_check(h, d_u, d_v, d_T, d_u_out, d_v_out, d_T_out, dt, dz, params, n_columns)) return 1;
    HIPCHK(hipSetDevice(h->device));
    Timed tm(h, K_IMPLDIFF);
    hipError_t e = launch_mpp_diffusion(d_u, d_v, d_T, d_halo_bottom, dt, dz, params, convective_adjustment, d_u_out, d_v_out, d_T_out,
                                        h->m.Nz, n_columns, h->stream);
    if (e != hipSuccess) return fail("implicit diffusion launch failed: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int colnde_implicit_diffusion(colnde_handle* h, const float* u, const float* v, const float* T, const float* halo_bottom,
                                         float dt, float dz, const float params[7], int convective_adjustment, float* u_out,
                                         float* v_out, float* T_out, int n_columns) {
    if (impl_diff_check(h, u, v, T, u_out, v_out, T_out, dt, dz, params, n_columns)) return 1;
    HIPCHK(hipSetDevice(h->device));
    const size_t nf = (size_t)n_columns * h->m.Nz, nh = (size_t)n_columns;
    float* d = nullptr;       // [u | v | T | halo(3 n_col)]: a scratch of its own — the handle's are sized for its own state vector
    HIPCHK(hipMalloc((void**)&d, (3 * nf + 3 * nh) * sizeof(float)));
    int rc = 1;
    do {
        const float* srcs[3] = {u, v, T};
        bool ok = true;
        for (int f = 0; f < 3 && ok; f++)
            ok = hipMemcpyAsync(d + f * nf, srcs[f], nf * sizeof(float), hipMemcpyHostToDevice, h->stream) == hipSuccess;
        if (ok && halo_bottom) ok = hipMemcpyAsync(d + 3 * nf, halo_bottom, 3 * nh * sizeof(float), hipMemcpyHostToDevice, h->stream) == hipSuccess;
        if (!ok) { fail("implicit diffusion: host-to-device copy failed"); break; }
        if (colnde_implicit_diffusion_dev(h, d, d + nf, d + 2 * nf, halo_bottom ? d + 3 * nf : nullptr, dt, dz, params,
                                          convective_adjustment, d, d + nf, d + 2 * nf, n_columns)) break;
        float* dsts[3] = {u_out, v_out, T_out};
        for (int f = 0; f < 3 && ok; f++)
            ok = hipMemcpyAsync(dsts[f], d + f * nf, nf * sizeof(float), hipMemcpyDeviceToHost, h->stream) == hipSuccess;
        if (!ok || hipStreamSynchronize(h->stream) != hipSuccess) { fail("implicit diffusion: device-to-host copy failed"); break; }
        rc = 0;
    } while (0);
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(d);
    return rc;
}

extern "C" int colnde_adam_step_dev(colnde_handle* h, float* d_weights, const float* d_grad, float* d_m, float* d_v, float eta,
                                    float beta1, float beta2, float eps, float beta1_t, float beta2_t, int n) {
    if (!h) return fail("null handle");
    if (!d_weights || !d_grad || !d_m || !d_v) return fail("null pointer argument");
    if (n < 1) return fail("n must be >= 1");
    if (!(beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f)) return fail("0 <= beta < 1 required");
    if (!(beta1_t < 1.0f && beta2_t < 1.0f)) return fail("running powers beta^t must be < 1");
    HIPCHK(hipSetDevice(h->device));
    Timed tm(h, K_ADAM);
    hipError_t e = launch_adam_step(d_weights, d_grad, d_m, d_v, eta, beta1, beta2, eps, beta1_t, beta2_t, n, h->stream);
    if (e != hipSuccess) return fail("ADAM launch failed: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int colnde_coarse_grain_dev(colnde_handle* h, const float* d_in, int n_rows, int N, int n, int location, float* d_out) {
    if (!h) return fail("null handle");
    if (!d_in || !d_out) return fail("null pointer argument");
    if (n_rows < 1 || N < 2 || n < 2 || n > N) return fail("need n_rows >= 1 and 2 <= n <= N (n_rows = %d, N = %d, n = %d)", n_rows, N, n);
    if (location != 0 && location != 1) return fail("location must be 0 (Center) or 1 (Face)");
    if (location == 0 && N % n != 0) return fail("coarse_grain(Center): n = %d must divide N = %d", n, N);
    HIPCHK(hipSetDevice(h->device));
    hipError_t e = launch_coarse_grain(d_in, n_rows, N, n, location, d_out, h->stream);
    if (e != hipSuccess) return fail("coarse_grain launch failed: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int colnde_zscore_stats_dev(colnde_handle* h, const float* d_x, int64_t count, float* d_mu_sigma) {
    if (!h) return fail("null handle");
    if (!d_x || !d_mu_sigma) return fail("null pointer argument");
    if (count < 2) return fail("need at least two values for a standard deviation");
    HIPCHK(hipSetDevice(h->device));
    hipError_t e = launch_zscore_stats(d_x, (long)count, d_mu_sigma, h->stream);
    if (e != hipSuccess) return fail("zscore_stats launch failed: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int colnde_scale_dev(colnde_handle* h, const float* d_x, int64_t count, const float* d_mu_sigma, float* d_out) {
    if (!h) return fail("null handle");
    if (!d_x || !d_mu_sigma || !d_out) return fail("null pointer argument");
    if (count < 1) return fail("count must be >= 1");
    HIPCHK(hipSetDevice(h->device));
    hipError_t e = launch_zscore_scale(d_x, (long)count, d_mu_sigma, d_out, h->stream);
    if (e != hipSuccess) return fail("scale launch failed: %s", hipGetErrorString(e));
    return 0;
}

// [grad; terms; total; 0] of this rank summed over the communicator, on the handle's stream (comm.hip)
extern "C" int colnde_comm_allreduce_dev(colnde_comm* c, float* d_buf, int64_t n, int op, void* hip_stream);
extern "C" int colnde_allreduce_result_dev(colnde_handle* h, colnde_comm* comm, float* d_out) {
    if (!h || !comm || !d_out) return fail("null argument");
    return colnde_comm_allreduce_dev(comm, d_out, (int64_t)h->m.n_params + 8, 0, (void*)h->stream);
}

// ---- flux-MLP pre-training (second half) ---------------------------------------------------------------------
extern "C" int colnde_pretrain_flux_dev(colnde_handle* h, int flux_type, float* d_theta, float* d_m, float* d_v, const float* d_profiles,
                                        const float* d_bcs, const float* d_flux, const int32_t* d_order, int n_samples,
                                        float gradient_scaling, float eta, float beta1, float beta2, float eps, double beta_t[2],
                                        int update, float* mean_loss) {
    if (!h) return fail("null handle");
    if (!d_theta || !d_profiles || !d_bcs || !d_flux || !beta_t || !mean_loss) return fail("null pointer argument");
    if (update && (!d_m || !d_v)) return fail("the ADAM moments are needed when update != 0");
    if (n_samples < 1) return fail("n_samples must be >= 1");
    const bool wm = h->m.model == COLNDE_MODEL_WIND_MIXING;
    if (flux_type < 0 || flux_type > 2 || (!wm && flux_type != 2)) return fail("flux_type: 0 = uw, 1 = vw, 2 = wT (T-only models: 2)");
    if (h->m.smooth_NN || h->m.smooth_Ri) return fail("flux pre-training does not cover the smoothing options");
    if (h->m.inplace) return fail("flux pre-training uses the training arithmetic (inplace_variant = 0)");
    if (!(beta_t[0] < 1.0 && beta_t[1] < 1.0)) return fail("running powers beta^t must be < 1");
    HIPCHK(hipSetDevice(h->device));
    float* d_loss = nullptr;
    HIPCHK(hipMalloc((void**)&d_loss, sizeof(float) + 2 * sizeof(double) + 8));
    double* d_bt = reinterpret_cast<double*>(reinterpret_cast<char*>(d_loss) + 8);
    hipError_t e = launch_pretrain(h->m, flux_type, d_theta, d_m, d_v, d_profiles, d_bcs, d_flux, d_order, n_samples, gradient_scaling,
                                   eta, beta1, beta2, eps, beta_t[0], beta_t[1], update, d_loss, d_bt, h->stream);
    float loss = 0.0f;
    double bt[2] = {beta_t[0], beta_t[1]};
    if (e == hipSuccess) e = hipMemcpyAsync(&loss, d_loss, sizeof(float), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(bt, d_bt, sizeof(bt), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d_loss);
    if (e != hipSuccess) return fail("flux pre-training failed: %s", hipGetErrorString(e));
    *mean_loss = loss / (float)n_samples;
    if (update) { beta_t[0] = bt[0]; beta_t[1] = bt[1]; }
    return 0;
}

extern "C" int colnde_plan(const colnde_handle* h, int info[8]) {
    if (!h || !info) return fail("null argument");
    for (int i = 0; i < 8; i++) info[i] = 0;
    info[0] = h->use_rt ? COLNDE_ENGINE_MFMA : (h->use_fc ? COLNDE_ENGINE_FC32 : COLNDE_ENGINE_GENERIC);
    // RKC2 on a model with a convective-adjustment switch: the gradient is the one-switch-pattern pullback (include/colnde.h)
    info[7] = (h->cfg.stepper == COLNDE_STEPPER_RKC2 && (h->m.model == COLNDE_MODEL_CONV_ADJ_NDE || (h->m.ca && !h->m.mpp))) ? 1 : 0;
    // bits 1..3: which kernel families run the exact three-way bf16 split (where the engine has a split kernel for this shape)
    bool bf_fwd = false, bf_adj = false, bf_dw = false;
    if (h->use_rt) {
        bf_fwd = h->sp_fwd && !h->rt_fwd32;
        bf_adj = h->sp_adj && (h->d_rt_tape ? h->rt_ztape : !h->rt_fwd32);     // (needs the Z1 tape; before the tapes are planned: the expectation)
        bf_dw = h->sp_dw;
    } else if (h->use_fc) {
        bf_fwd = h->sp_fwd && fc_split_supported(h->fc_cw);
        bf_adj = h->sp_adj && fc_split_supported(h->fc_cw);
        bf_dw = h->sp_dw && (h->d_dwtape ? !h->dw_split.passes.empty() : true);
    } else {
        bf_fwd = (h->sp_fwd && h->fwd_split && h->fwd_helper) || h->m.sf != nullptr;
        bf_adj = (h->sp_adj && h->adj_split && rt_adjoint_split_has_bf16(h->m, h->adj_helper) && (h->t16_dwtape < 0 || (h->t16_dwtape == 1 && h->d_t16_ztape))) || h->m.sb != nullptr;
        bf_dw = h->sp_dw && h->t16_dwtape == 1 && !h->dw_split.passes.empty();
    }
    info[7] |= (bf_fwd ? 2 : 0) | (bf_adj ? 4 : 0) | (bf_dw ? 8 : 0);
    if (h->use_fc) {
        info[1] = h->fc_block;
        info[2] = h->fc_nblocks;
        info[3] = h->fc_nseg > 1 ? h->fc_nseg : 0;       // fc32: time segments of the tapes (0: the tapes hold the whole axis)
        info[4] = h->d_dwtape ? 1 : 0;
        info[5] = h->d_dwtape ? h->dw_slices : 0;
    } else if (h->use_rt) {
        info[1] = h->rt_block;
        info[2] = h->rt_nblocks;
        info[3] = h->rt_ztape ? 1 : 0;
    } else {
        info[1] = h->t16_dwtape == 1 ? h->t16_block : 0;
        info[2] = h->t16_dwtape == 1 ? h->t16_nblocks : 0;
        info[4] = h->t16_dwtape == 1 ? 1 : 0;
        info[5] = h->t16_dwtape == 1 ? h->dw_slices : 0;
        info[6] = (h->fwd_split ? 1 : 0) | ((h->adj_split && h->t16_dwtape == 1 && h->d_t16_ztape) ? 2 : 0) | (h->split_rich ? 4 : 0);
    }
    return 0;
}

// Every environment variable some part of the library reads (api.hip, engine_*.hip): the list colnde_describe reports from.
static const char* const COLNDE_ENV_SWITCHES[] = {
    "COLNDE_FWD_SPLIT", "COLNDE_ADJ_SPLIT", "COLNDE_DW_SPLIT", "COLNDE_ADJ_GEOM", "COLNDE_FWD_WLDS", "COLNDE_FWD_THREADS", "COLNDE_T16_FWD_HELPER",
    "COLNDE_T16_ADJ_HELPER", "COLNDE_T16_FWD_SPLIT", "COLNDE_T16_ADJ_SPLIT", "COLNDE_FC", "COLNDE_FC_CW", "COLNDE_FC_BLOCK", "COLNDE_FC_SEG",
    "COLNDE_RT_ZTAPE", "COLNDE_RT_BLOCK", "COLNDE_RT_FWD", "COLNDE_ALLOW_UNSTABLE_DT", "COLNDE_T16_DWTAPE", "COLNDE_T16_ZTAPE", "COLNDE_T16_SPLIT_RICH",
    "COLNDE_T16_BLOCK", "COLNDE_T16_DWLDS", "COLNDE_T16_TAPE_THREADS", "COLNDE_T16_TAPE_WLDS"};

extern "C" int colnde_describe(const colnde_handle* h, char* buf, int capacity) {
    if (!h) { fail("null argument"); return -1; }
    int info[8];
    if (colnde_plan(h, info)) return -1;
    std::string s;
    char t[256];
    const char* eng = info[0] == COLNDE_ENGINE_MFMA ? "regtile" : info[0] == COLNDE_ENGINE_FC32 ? "fc32" : ((info[6] & 1) ? "tile16+net-split" : "tile16");
    char why[96] = "";
    if (h->auto_substeps) snprintf(why, sizeof why, "(stability bound; pending: the first solve chooses from reltol=%g)", h->cfg.reltol);
    else if (h->substeps_chosen) snprintf(why, sizeof why, "(chosen from reltol, estimate=%.3g)", h->last_estimate);
    snprintf(t, sizeof t, "engine=%s columns=%d stepper=%s substeps=%d%s", eng, h->cfg.n_columns, h->cfg.stepper == COLNDE_STEPPER_RKC2 ? "rkc2" : "rk4",
             h->cfg.substeps, why);
    s += t;
    if (h->cfg.stepper == COLNDE_STEPPER_RKC2) { snprintf(t, sizeof t, " rkc_stages=%d%s", h->m.nst, h->cfg.rkc_stages ? "" : "(automatic)"); s += t; }
    snprintf(t, sizeof t, " matrix_arithmetic=%s forward=%s adjoint=%s dw=%s", h->cfg.matrix_arithmetic == COLNDE_MATRIX_BF16X3_EXACT ? "bf16x3_exact" : "f32_mfma",
             (info[7] & 2) ? "bf16x3" : "f32", (info[7] & 4) ? "bf16x3" : "f32", (info[7] & 8) ? "bf16x3" : "f32");
    s += t;
    if (info[1]) { snprintf(t, sizeof t, " block=%dx%d", info[1], info[2]); s += t; } else s += " block=(not planned yet)";
    if (info[0] == COLNDE_ENGINE_MFMA) { snprintf(t, sizeof t, " z1_tape=%d", info[3]); s += t; }
    if (info[0] == COLNDE_ENGINE_FC32) { snprintf(t, sizeof t, " time_segments=%d tile_width=%d dw_slices=%d", info[3], h->fc_cw, info[5]); s += t; }
    if (h->ag_rows) s += " activation_rows=global_memory(L2)";
    if (info[0] == COLNDE_ENGINE_GENERIC) {
        snprintf(t, sizeof t, " dw_taped=%d dw_slices=%d net_split_forward=%d net_split_adjoint=%d rich_tape=%d", info[4], info[5], info[6] & 1, (info[6] >> 1) & 1, (info[6] >> 2) & 1);
        s += t;
    }
    snprintf(t, sizeof t, " gradient=%s", (info[7] & 1) ? "one-switch-pattern RKC2 pullback (approximate)" : "exact discrete adjoint");
    s += t;
    s += " | env";
    bool any = false;
    for (const char* name : COLNDE_ENV_SWITCHES) {
        const char* e = getenv(name);
        if (e && *e) { s += " "; s += name; s += "="; s += e; any = true; }
    }
    if (!any) s += " (none set)";
    // an environment override that contradicts the arithmetic selected through the API (colnde_config.matrix_arithmetic / colnde_set_matrix_arithmetic) wins —
    // say so, by name (it used to outrank the API call silently)
    {
        const bool split = h->cfg.matrix_arithmetic == COLNDE_MATRIX_BF16X3_EXACT;
        const struct { const char* name; bool v; } fam[3] = {{"COLNDE_FWD_SPLIT", h->sp_fwd}, {"COLNDE_ADJ_SPLIT", h->sp_adj}, {"COLNDE_DW_SPLIT", h->sp_dw}};
        for (const auto& f : fam)
            if (f.v != split) { snprintf(t, sizeof t, " | WARNING %s=%d overrides matrix_arithmetic=%s for its kernels", f.name, f.v ? 1 : 0, split ? "bf16x3_exact" : "f32_mfma"); s += t; }
    }
    const int need = (int)s.size() + 1;
    if (buf && capacity > 0) {
        const int n = need <= capacity ? need - 1 : capacity - 1;
        memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return need;
}

// Diagnostic builds (-DCOLNDE_STAMPS) only; not part of include/colnde.h.  Returns zeros in the shipped library.
extern "C" int colnde_debug_stamps(colnde_handle* h, unsigned long long* out16) {
    if (!h || !out16) return fail("null argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->use_rt || h->adj_split) {
        for (int i = 0; i < 16; i++) out16[i] = 0;
        HIPCHK(rt_debug_read_stamps(out16));
        return 0;
    }
    HIPCHK(debug_read_stamps(out16));
    return 0;
}
