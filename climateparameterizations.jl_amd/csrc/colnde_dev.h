// colnde_dev.h — device-side model description shared by the kernels and the host API (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/colnde.h"

#define CT 16            // columns per workgroup tile = N of v_mfma_f32_16x16x4_f32
#define MAX_TILE_DESC 4096

typedef float f32x4 __attribute__((ext_vector_type(4)));

// One 16x16 weight-gradient tile owned by one wave for the whole adjoint kernel.
struct TileDesc {
    int a_off;    // offset (floats) of the A-operand feature block inside a column's input row
    int a_src;    // 0: layer input is the state row xs; 1: activation row of `net`
    int d_off;    // offset of the delta (dz) feature block inside the column's Z row
    int net;
    int ni_rem;   // valid input features in this tile (<=16)
    int no_rem;   // valid output features in this tile (<=16)
    int g_off;    // offset in the flat gradient of W[out j0][in i0] (element (i,j) at g_off + i*no + j)
    int no;       // leading dimension (layer outputs)
};

// One 64x64 block of a layer's weight gradient, contracted by one wavefront of dw_gemm_kernel from the taped rows
// [xs | A of every net | dZ of every net] (see adjoint_kernel<..., TAPEDW>).
struct DwMacro {
    int a_feat;   // offset of the block's first input feature inside a taped row
    int d_feat;   // offset of the block's first delta feature inside a taped row
    int ni_rem;   // valid input features (<= 64)
    int no_rem;   // valid output features (<= 64)
    int g_off;    // offset in the flat gradient of W[out j0][in i0] (element (i, j) at g_off + i*no + j)
    int no;       // leading dimension (layer outputs)
};

struct DevModel {
    int model, Nz, ns, n_nets, n_bc, n_layers;
    int sizes[COLNDE_MAX_LAYERS + 1];
    int acts[COLNDE_MAX_LAYERS];
    int w_off[COLNDE_MAX_LAYERS], b_off[COLNDE_MAX_LAYERS];   // inside one net, Flux.destructure order
    int act_off[COLNDE_MAX_LAYERS + 1];                       // act_off[l]: layer l (1-based) output offset in a row
    int net_size, n_params, act_total, n_bias;
    int ld_x, ld_a, ld_f;                                     // LDS row strides (floats), == 2 (mod 4)
    int mpp, ca, zero_w, smooth_NN, smooth_Ri, diurnal, inplace;
    int n_tiles, tiles_per_wave;
    float cs[3], A[3], s0[3], B, cor_u, cor_v, C_fc;
    float sig_u, sig_v, mu_u, mu_v, mu_wT, sig_wT, mu_T, sig_T;
    float nu0, nu_minus, Ric, dRi, inv_dRi, Pr, inv_Pr, c_rib, kappa, eps, ca_K, tau, alpha_g;
    // time stepper: nst = RHS evaluations (= taped stage inputs) per step: 4 for classical RK4, s for the s-stage RKC2 step;
    // rkc = device table [6][RKC_LD] of (mu, nu, mu~, gamma~, c, 1 - mu - nu), index j = 0..s (nullptr: RK4)
    int nst;
    const float* rkc;
    // Networks whose per-tile activation rows [net][CT][ld_a] do not fit the CU's LDS (the reference's wide wind-mixing architectures,
    // 3 x 96-400-400-31: 160 KB of rows alone): the tile16 kernels then keep THAT array in a per-workgroup slab of global memory (L2-resident:
    // written and read by the same workgroup between its own barriers) and everything else in LDS as before.  nullptr: rows in LDS.
    float* ag;
    // ... and, for those networks under COLNDE_MATRIX_BF16X3_EXACT, the dense chains of the tile16 kernels run on the bf16 pipe from pre-split weight planes (round 5):
    // sf / sb = forward / transposed A-operand images [net][layer][16-row tile][32-deep k-block][plane h, m, l][64 lanes] of 16 bytes (8 bf16: k = 32 S + 8 (lane >> 4) + e),
    // offsets in 16-byte units.  nullptr: f32 chains (packed f32 images).
    const unsigned* sf;
    const unsigned* sb;
    int sf_off[COLNDE_MAX_LAYERS], sb_off[COLNDE_MAX_LAYERS], sf_net, sb_net;
};
#define RKC_LD 260

// a / b on the reciprocal unit (v_rcp_f32, 1 ulp).  NOTE: HIP's __fdividef(a, b) compiles to the full IEEE division
// sequence (v_div_scale / v_div_fmas / v_div_fixup, ~10 instructions) unless fast-math is on: not used here.
__device__ __forceinline__ float fast_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

// ---- activations (NNlib 0.7: relu, mish, swish, tanh, leakyrelu) ------------------------------------
__device__ __forceinline__ float dev_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// mish(x) = x tanh(softplus(x));  tanh(log(1+e^x)) = ((1+e^x)^2 - 1)/((1+e^x)^2 + 1) = n/(n+2), n = e^x (e^x + 2)
__device__ __forceinline__ float dev_mish_t(float x) {
    float e = __expf(fminf(x, 20.0f));
    float n = e * (e + 2.0f);
    return fast_div(n, n + 2.0f);
}

// tanh(z) = 1 - 2 / (1 + e^{2z}) on the exponential and reciprocal units
__device__ __forceinline__ float dev_tanh(float z) {
    const float e = __expf(2.0f * fminf(fmaxf(z, -15.0f), 15.0f));
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}

__device__ __forceinline__ float dev_act(int a, float z) {
    switch (a) {
        case COLNDE_ACT_RELU: return fmaxf(z, 0.0f);
        case COLNDE_ACT_MISH: return z * dev_mish_t(z);
        case COLNDE_ACT_SWISH: return z * dev_sigmoid(z);
        case COLNDE_ACT_TANH: return dev_tanh(z);
        case COLNDE_ACT_LEAKYRELU: return z > 0.0f ? z : 0.01f * z;
        default: return z;
    }
}

__device__ __forceinline__ float dev_act_grad(int a, float z) {
    switch (a) {
        case COLNDE_ACT_RELU: return z > 0.0f ? 1.0f : 0.0f;
        case COLNDE_ACT_MISH: {      // e w / (n + 2)^2, w = 4 (z + 1) + 4 e^2 + e^3 + e (4 z + 6): one exponential, one reciprocal
            const float e = __expf(fminf(z, 20.0f)), n = e * (e + 2.0f), r = __builtin_amdgcn_rcpf(n + 2.0f), p = 4.0f * z + 4.0f;
            return (e * r) * (fmaf(e, (n + 2.0f * e) + (p + 2.0f), p) * r);
        }
        case COLNDE_ACT_SWISH: { float s = dev_sigmoid(z); return s + z * s * (1.0f - s); }
        case COLNDE_ACT_TANH: { float t = dev_tanh(z); return 1.0f - t * t; }
        case COLNDE_ACT_LEAKYRELU: return z > 0.0f ? 1.0f : 0.01f;
        default: return 1.0f;
    }
}
