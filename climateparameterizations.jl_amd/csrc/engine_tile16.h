// engine_tile16.h — host/device interface of the MFMA tile engine.
#pragma once
#include <vector>
#include "colnde_dev.h"

struct PackInfo {
    int pf_off[COLNDE_MAX_LAYERS + 1];   // per-layer offsets inside one net's forward A-operand image
    int pb_off[COLNDE_MAX_LAYERS + 1];   // ... backward (transposed) image
    int pf_net, pb_net;                  // floats per net
};

struct LossWeights { float w[8]; };       // w[q] = loss_scaling[q] / (global element count of term q)

// launch geometry chosen by the host for the adjoint kernel
struct AdjointGeom { int nthreads, maxt, maxr, wlds; };

#define MODEL_FLOATS ((int)((sizeof(DevModel) / 4 + 3) & ~3))   // the model description at the head of the adjoint kernel's LDS

static inline int lds_pad(int n) { int p = (n + 3) & ~3; return p + 2; }   // == 2 (mod 4)

static inline size_t lds_floats_forward(const DevModel& m) {
    return (size_t)2 * CT * m.ld_x + (size_t)m.n_nets * CT * m.ld_a + (size_t)4 * CT * m.ld_f + CT * 8;
}
static inline size_t lds_floats_adjoint(const DevModel& m) {
    return (size_t)3 * CT * m.ld_x + (size_t)2 * m.n_nets * CT * m.ld_a + (size_t)5 * CT * m.ld_f + CT * 8 + 16 * 8;
}

// DevModel::ag set (rows in global memory): what remains in LDS
static inline size_t lds_floats_forward_ag(const DevModel& m) { return lds_floats_forward(m) - (size_t)m.n_nets * CT * m.ld_a; }
static inline size_t lds_floats_adjoint_ag(const DevModel& m) { return lds_floats_adjoint(m) - (size_t)2 * m.n_nets * CT * m.ld_a; }
static inline size_t ag_floats_per_tile(const DevModel& m) { return (size_t)m.n_nets * CT * m.ld_a; }

// taped mode with the hidden pre-activations taped: no activation array on chip
static inline size_t lds_floats_adjoint_noA(const DevModel& m) { return lds_floats_adjoint(m) - (size_t)m.n_nets * CT * m.ld_a; }

hipError_t set_kernel_attributes(size_t max_lds_bytes);
hipError_t launch_pack(const DevModel& m, const PackInfo& pk, const float* w, float* wf, float* wb, hipStream_t stream);
// bf16 plane images of the dense chains (DevModel::sf / sb; offsets and sizes in m, 16-byte units): the weights split exactly, once per call
hipError_t launch_pack_planes(const DevModel& m, const float* w, unsigned* sf, unsigned* sb, hipStream_t stream);
// dx [n_col][ns] tendencies and / or flux [n_col][n_nets][Nz + 1] face fluxes (predict_flux); either may be null
hipError_t launch_rhs(const DevModel& m, const PackInfo& pk, const float* w, const float* wf, const float* x,
                      const float* bcs, float t, float* dx, int n_col, int nthreads, size_t lds_bytes, hipStream_t stream, float* flux = nullptr);
hipError_t launch_forward(const DevModel& m, const PackInfo& pk, const float* w, const float* wf, const float* x0,
                          const float* bcs, const float* save_times, int n_save, int substeps, float* sol, float* tape,
                          int n_col, int nthreads, bool wlds, size_t lds_bytes, hipStream_t stream, float* ztape = nullptr);
hipError_t launch_loss(const DevModel& m, const float* sol, const float* truth, int n_save, int n_col, float* partial,
                       int n_blocks, hipStream_t stream);
hipError_t launch_adjoint(const DevModel& m, const PackInfo& pk, const float* w, const float* wf, const float* wb,
                          const TileDesc* tiles, const int* bias_zoff, const int* bias_goff, const float* bcs,
                          const float* save_times, int n_save, int substeps, const float* sol, const float* truth,
                          const float* tape, const LossWeights& lw, float* slab, int n_col, const AdjointGeom& geo,
                          size_t lds_bytes, hipStream_t stream, float* dwtape = nullptr, const float* ztape = nullptr);
// hidden pre-activation tape of the taped-dW mode: floats per column
static inline size_t t16_ztape_col_floats(const DevModel& m) { return (size_t)((m.n_nets * m.act_off[m.n_layers - 1] + 3) & ~3); }
// taped-dW mode (dwtape != nullptr above): floats per 16-column tile and stage, and the contraction kernel
static inline int dwtape_ns4(const DevModel& m) { return (m.ns + 3) & ~3; }
static inline int dwtape_act4(const DevModel& m) { return (m.act_total + 3) & ~3; }
static inline size_t dwtape_row_floats(const DevModel& m) { return (size_t)dwtape_ns4(m) + (size_t)2 * m.n_nets * dwtape_act4(m); }
bool dw_gemm_lds_fits(int row_floats, int n_macros);     // the LDS-staged dW kernel applies (else the L2-streaming one)
hipError_t launch_dw_gemm(const float* dwtape, size_t n_records, int row_floats, const DwMacro* macros, int n_macros, int n_slices,
                          float* slab_rows, int slab_stride, hipStream_t stream);
// dW GEMM on the bf16 pipe with exact three-way operand splitting (COLNDE_MATRIX_BF16X3_EXACT; the plan is built with the tapes whenever the records fit LDS): the blocks are
// dealt to passes by layer so that a pass's operand features (split ONCE per record into LDS planes) and its accumulators fit one workgroup
struct DwSeg { int src, len, dst; };                                  // floats of a record row [src, src + len) -> compact features [dst, dst + len)
struct DwPassDesc { DwSeg seg[8]; int n_seg, Fc, m0, n_macros, maxm, nit; };
struct DwSplitPlan { std::vector<DwPassDesc> passes; DwMacro* d_macros = nullptr; };
bool dw_split_build(const std::vector<DwMacro>& mac, const std::vector<int>& matrix_of, int row_floats, DwSplitPlan& plan);
void dw_split_free(DwSplitPlan& plan);
hipError_t launch_dw_gemm_split(const float* dwtape, size_t n_records, int row_floats, const DwSplitPlan& plan, int n_slices,
                                float* slab_rows, int slab_stride, hipStream_t stream);
hipError_t launch_reduce(const float* slab, int n_tiles, int n_params, int stride, const LossWeights& lw, float* out,
                         hipStream_t stream);
hipError_t launch_infer(const DevModel& m, const PackInfo& pk, const float* w, const float* wf, const float* T,
                        const float* top_flux, float inv_dz, float* out, int n_col, int nthreads, size_t lds_bytes,
                        hipStream_t stream);
// flux-MLP pre-training (train_NN): one pass over the data set, one ADAM update per sample (update = 0: mean-loss evaluation only)
hipError_t launch_pretrain(const DevModel& m, int flux_type, float* theta, float* mom, float* vel, const float* X, const float* BC,
                           const float* Y, const int* order, int n_samples, float gs, float eta, float b1, float b2, float eps,
                           double bt1, double bt2, int update, float* loss_out, double* bt_out, hipStream_t stream);
bool pick_adjoint_geom(const DevModel& m, AdjointGeom* geo, int force);
size_t lds_floats_adjoint_geom(const DevModel& m, const AdjointGeom& g);
hipError_t debug_read_stamps(unsigned long long* out16);
