// column_ops.hip — gfx950 kernels for the steps either side of the NDE hot path.
#include "column_ops.h"

// ------------------------------------------------------------------------------------------------
// convective_adjustment!(model, Δt, K): free_convection/double_gyre_nn.jl:27-62 (3-D), free_convection/src/oceananigans_nn.jl:13-40
// (1-D).  Per column: the centred vertical gradient (zero-gradient halos) marks the statically unstable cells,
// κ_k = K there and 0 elsewhere, and T' = L \ T with the reference's tridiagonal
//     lower_k = -c κ_k (k >= 1),  diag_k = 1 + c (κ_k + κ_{k+1}) (k < Nz-1),  diag_{Nz-1} = 1 + c κ_{Nz-1},  upper_k = -c κ_{k+1},
// c = Δt/Δz².  L is strictly diagonally dominant by rows and columns, so the reference's pivoted LU (`Tridiagonal \`)
// never pivots and the Thomas recurrence below is the same elimination.
//
// HBM-bound (2·4·Nz bytes per column, ~8 flop per level).  A workgroup stages its columns through LDS with coalesced float4 traffic
// (rows padded to NZ+1 floats: conflict-free per-thread column walks), one thread solves one column.
// ------------------------------------------------------------------------------------------------
// Round 3: ONE WAVE per workgroup (64 columns, 8.4 KB of LDS at Nz = 32) instead of four — what took the three-field kernel below to 70 % of the
// HBM peak: many small workgroups per CU are in different phases, one's solve hides under the others' float4 traffic (256-thread workgroups: 4.6–4.9 TB/s).
template <int NZ>
__global__ void __launch_bounds__(64) convadj_kernel(const float* T, const float* __restrict__ halo_bottom,
                                                     const float* __restrict__ halo_top, float c, float K, float* out, int n_col) {
    extern __shared__ float cs_smem[];
    constexpr int LD = NZ + 1, Q = NZ / 4;
    const int lane = threadIdx.x;
    const int col0 = blockIdx.x * 64;
    const int ncol = min(64, n_col - col0);
    typedef float ca_f32x4 __attribute__((ext_vector_type(4)));
    const ca_f32x4* src = reinterpret_cast<const ca_f32x4*>(T + (size_t)col0 * NZ);
    if (ncol == 64) {
        ca_f32x4 r[Q];
#pragma unroll
        for (int i = 0; i < Q; i++) r[i] = __builtin_nontemporal_load(src + i * 64 + lane);
#pragma unroll
        for (int i = 0; i < Q; i++) {
            const int e = i * 64 + lane, cl = e / Q, k = (e % Q) * 4;
            float* d = cs_smem + cl * LD + k;
            d[0] = r[i].x; d[1] = r[i].y; d[2] = r[i].z; d[3] = r[i].w;
        }
    } else {
        for (int e = lane; e < ncol * Q; e += 64) {
            const ca_f32x4 v = src[e];
            const int cl = e / Q, k = (e % Q) * 4;
            float* d = cs_smem + cl * LD + k;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    }
    __syncthreads();
    if (lane < ncol) {
        float* t = cs_smem + lane * LD;
        float x[NZ], cp[NZ];
#pragma unroll
        for (int k = 0; k < NZ; k++) x[k] = t[k];
        const float ck = c * K;
        // κ of cell k as c·κ_k: statically unstable where T[k+1] - T[k-1] < 0; the halo cells are the caller's (they carry
        // the field's boundary conditions) or, absent, the nearest interior value (zero-gradient fill)
        const float below = halo_bottom ? halo_bottom[col0 + lane] : x[0];
        const float above = halo_top ? halo_top[col0 + lane] : x[NZ - 1];
        float kk[NZ];
#pragma unroll
        for (int k = 0; k < NZ; k++) kk[k] = ((k + 1 < NZ ? x[k + 1] : above) - (k > 0 ? x[k - 1] : below)) < 0.0f ? ck : 0.0f;
        // forward elimination
        float inv = 1.0f / (1.0f + kk[0] + kk[1]);
        cp[0] = -kk[1] * inv;
        x[0] = x[0] * inv;
#pragma unroll
        for (int k = 1; k < NZ; k++) {
            const float a = -kk[k];
            const float b = 1.0f + kk[k] + (k < NZ - 1 ? kk[k + 1] : 0.0f);
            inv = 1.0f / (b - a * cp[k - 1]);
            cp[k] = (k < NZ - 1 ? -kk[k + 1] : 0.0f) * inv;
            x[k] = (x[k] - a * x[k - 1]) * inv;
        }
        // back substitution
#pragma unroll
        for (int k = NZ - 2; k >= 0; k--) x[k] -= cp[k] * x[k + 1];
#pragma unroll
        for (int k = 0; k < NZ; k++) t[k] = x[k];
    }
    __syncthreads();
    ca_f32x4* dst = reinterpret_cast<ca_f32x4*>(out + (size_t)col0 * NZ);
    if (ncol == 64) {
#pragma unroll
        for (int i = 0; i < Q; i++) {
            const int e = i * 64 + lane, cl = e / Q, k = (e % Q) * 4;
            const float* d = cs_smem + cl * LD + k;
            const ca_f32x4 q = {d[0], d[1], d[2], d[3]};
            __builtin_nontemporal_store(q, dst + e);
        }
    } else {
        for (int e = lane; e < ncol * Q; e += 64) {
            const int cl = e / Q, k = (e % Q) * 4;
            const float* d = cs_smem + cl * LD + k;
            const ca_f32x4 q = {d[0], d[1], d[2], d[3]};
            dst[e] = q;
        }
    }
}

// any Nz <= 128 (not a multiple of 4, or none of the instantiated sizes): one thread per column straight from HBM
__global__ void __launch_bounds__(256) convadj_generic_kernel(const float* __restrict__ T, const float* __restrict__ halo_bottom,
                                                               const float* __restrict__ halo_top, float c, float K,
                                                               float* __restrict__ out, int Nz, int n_col) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= n_col) return;
    const float* t = T + (size_t)col * Nz;
    float* o = out + (size_t)col * Nz;
    float x[128], cp[128], kk[128];
    for (int k = 0; k < Nz; k++) x[k] = t[k];
    const float ck = c * K;
    const float below = halo_bottom ? halo_bottom[col] : x[0], above = halo_top ? halo_top[col] : x[Nz - 1];
    for (int k = 0; k < Nz; k++) kk[k] = ((k + 1 < Nz ? x[k + 1] : above) - (k > 0 ? x[k - 1] : below)) < 0.0f ? ck : 0.0f;
    float inv = 1.0f / (1.0f + kk[0] + (Nz > 1 ? kk[1] : 0.0f));
    cp[0] = (Nz > 1 ? -kk[1] : 0.0f) * inv;
    x[0] *= inv;
    for (int k = 1; k < Nz; k++) {
        const float a = -kk[k];
        const float b = 1.0f + kk[k] + (k < Nz - 1 ? kk[k + 1] : 0.0f);
        inv = 1.0f / (b - a * cp[k - 1]);
        cp[k] = (k < Nz - 1 ? -kk[k + 1] : 0.0f) * inv;
        x[k] = (x[k] - a * x[k - 1]) * inv;
    }
    for (int k = Nz - 2; k >= 0; k--) x[k] -= cp[k] * x[k + 1];
    for (int k = 0; k < Nz; k++) o[k] = x[k];
}

hipError_t launch_convective_adjustment(const float* T, const float* halo_bottom, const float* halo_top, float c, float K, float* out,
                                        int Nz, int n_col, hipStream_t stream) {
    if (Nz < 2 || Nz > 128 || n_col < 1) return hipErrorInvalidValue;
    const dim3 grid((n_col + 63) / 64), block(64);
#define CA_LAUNCH(N) hipLaunchKernelGGL(convadj_kernel<N>, grid, block, 64 * (N + 1) * sizeof(float), stream, T, halo_bottom, halo_top, c, K, out, n_col)
    const bool aligned = (((uintptr_t)T | (uintptr_t)out) & 15) == 0;
    if (aligned && Nz == 16) CA_LAUNCH(16);
    else if (aligned && Nz == 32) CA_LAUNCH(32);
    else if (aligned && Nz == 64) CA_LAUNCH(64);
    else hipLaunchKernelGGL(convadj_generic_kernel, dim3((n_col + 255) / 256), dim3(256), 0, stream, T, halo_bottom, halo_top, c, K, out, Nz, n_col);
#undef CA_LAUNCH
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Flux.Optimise.ADAM `apply!` followed by `update!` (Flux 0.11.6, src/optimise/optimisers.jl; call sites
// wind_mixing/src/NDE_training.jl:340-372, free_convection/src/training.jl:71): one fused pass over the parameter vector.
// beta1_t / beta2_t are the running powers β₁ᵗ, β₂ᵗ the optimiser state carries (β on the first step).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, float eta, float b1, float b2, float eps, float c1, float c2,
                                                    int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    w[i] -= mi * c1 / (sqrtf(vi * c2) + eps) * eta;      // c1 = 1/(1-β₁ᵗ), c2 = 1/(1-β₂ᵗ)
}

hipError_t launch_adam_step(float* w, const float* grad, float* m, float* v, float eta, float beta1, float beta2, float eps,
                            float beta1_t, float beta2_t, int n, hipStream_t stream) {
    if (n < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(adam_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, w, grad, m, v, eta, beta1, beta2, eps,
                       1.0f / (1.0f - beta1_t), 1.0f / (1.0f - beta2_t), n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// data preparation: wind_mixing/src/data_containers.jl:343-427 coarse-grains every LES profile
// 128 -> 32 cells / 129 -> 33 faces and z-scores each variable before training.
// ------------------------------------------------------------------------------------------------
// coarse_grain(Φ, n, Center), src/DataWrangling/coarse_graining.jl:8-16: block means, Δ = N / n.
// coarse_grain_linear_interpolation(Φ, n, Face), :47-62: end points kept, interior point i (1-based) at position
// p = 1 + (i-1)(N-1)/(n-1): (⌊p⌋ + 1 - p) Φ[⌊p⌋] + (p - ⌊p⌋) Φ[⌊p⌋ + 1].  One thread per output value; rows are profiles.
__global__ void __launch_bounds__(256) coarse_grain_kernel(const float* __restrict__ in, int n_rows, int N, int n, int face,
                                                           float* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)n_rows * n) return;
    const int row = (int)(idx / n), i = (int)(idx - (long)row * n);
    const float* src = in + (size_t)row * N;
    float v;
    if (!face) {
        const int d = N / n;
        float acc = 0.0f;
        for (int k = 0; k < d; k++) acc += src[i * d + k];
        v = acc / (float)d;
    } else if (i == 0) v = src[0];
    else if (i == n - 1) v = src[N - 1];
    else {
        const double p = 1.0 + (double)i * ((double)(N - 1) / (double)(n - 1));     // 1-based position, as the reference computes it
        const double fl = floor(p);
        const int k = (int)fl - 1;                                                    // 0-based index of Φ[⌊p⌋]
        v = (float)((fl + 1.0 - p) * (double)src[k] + (p - fl) * (double)src[k + 1 < N ? k + 1 : N - 1]);
    }
    out[idx] = v;
}

// ZeroMeanUnitVarianceScaling(data) (src/DataWrangling/feature_scaling.jl:17-20): μ = mean, σ = std (n - 1 in the denominator),
// accumulated in float64 by one workgroup in a fixed order (deterministic); out[0] = μ, out[1] = σ.
__global__ void __launch_bounds__(1024) zscore_stats_kernel(const float* __restrict__ x, long count, float* __restrict__ out2) {
    __shared__ double red[1024];
    double s = 0.0;
    for (long i = threadIdx.x; i < count; i += 1024) s += (double)x[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) { if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w]; __syncthreads(); }
    const double mu = red[0] / (double)count;
    __syncthreads();
    double q = 0.0;
    for (long i = threadIdx.x; i < count; i += 1024) { const double d = (double)x[i] - mu; q += d * d; }
    red[threadIdx.x] = q;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) { if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w]; __syncthreads(); }
    if (threadIdx.x == 0) { out2[0] = (float)mu; out2[1] = (float)sqrt(red[0] / (double)(count > 1 ? count - 1 : 1)); }
}

// scale(x, s) = (x - μ) / σ (feature_scaling.jl:22); μ, σ read from device memory (the output of zscore_stats_kernel)
__global__ void __launch_bounds__(256) zscore_scale_kernel(const float* __restrict__ x, long count, const float* __restrict__ mu_sigma,
                                                           float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < count) out[i] = (x[i] - mu_sigma[0]) / mu_sigma[1];
}

hipError_t launch_coarse_grain(const float* in, int n_rows, int N, int n, int face, float* out, hipStream_t stream) {
    if (n_rows < 1 || N < 2 || n < 2 || n > N) return hipErrorInvalidValue;
    if (!face && N % n != 0) return hipErrorInvalidValue;
    const long total = (long)n_rows * n;
    hipLaunchKernelGGL(coarse_grain_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, in, n_rows, N, n, face, out);
    return hipGetLastError();
}

hipError_t launch_zscore_stats(const float* x, long count, float* out2, hipStream_t stream) {
    if (count < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(zscore_stats_kernel, dim3(1), dim3(1024), 0, stream, x, count, out2);
    return hipGetLastError();
}

hipError_t launch_zscore_scale(const float* x, long count, const float* mu_sigma, float* out, hipStream_t stream) {
    if (count < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(zscore_scale_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, x, count, mu_sigma, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// modified_pacanowski_philander!(model, constants, Δt, p, convective_adjustment): wind_mixing/src/NDE_oceananigans.jl:61-101, with
// the face diffusivities of modified_pacanowski_philander_diffusivity (:17-58).  Per column, one backward-Euler diffusion step of
// u, v (shared matrix) and T:
//     Ri_k = gα (T_k − T_{k−1}) Δz / ((u_k − u_{k−1})² + (v_k − v_{k−1})²)          (Oceanostics 0.3.2 richardson_number_ccf!, the
//            (Center, Center, Face) ratio ∂z b / ((∂z u)² + (∂z v)²) with b = gαT; third-party, pinned in wind_mixing/Manifest.toml:1279)
//     ν_k  = ν₀ + ν₋ tanh_step((Ri_k − Riᶜ)/ΔRi) on the interior faces, 0 on face 0 (:45-47);   tanh_step(x) = (1 − tanh x)/2
//     νT_k = convective_adjustment ? (Ri_k > 0 ? ν_k/Pr : 1) : ν_k/Pr  on EVERY face (:49-55; face 0 sees the halo cells)
//     lower_k = −c ν_k,  diag_k = 1 + c (ν_k + ν_{k+1}) (k < Nz−1),  diag_{Nz−1} = 1 + c ν_{Nz−1},  upper_k = −c ν_{k+1},  c = Δt/Δz²
//     u′ = L_ν \ u,  v′ = L_ν \ v,  T′ = L_νT \ T,  then T′_0 = T_0 (`T′[1] = T_bottom`, :94).
// The same strictly diagonally dominant tridiagonal as convadj_kernel: the Thomas recurrence is the reference's LU without pivoting.
// tanh_step is evaluated as 1/(1 + e^{2x}) (one v_exp + one v_rcp; exact limits 0 / 1 at Ri = ±∞, NaN stays NaN as in the reference,
// which a face with no shear AND no stratification produces: 0/0).
//
// HBM-bound: 2·3·4·Nz bytes per column (768 B at Nz = 32), ≈ 60 flop per level.  One WAVE per workgroup stages 64 columns of the three
// fields through LDS (rows padded to NZ+1: conflict-free column walks; 25 KB at Nz = 32, so six workgroups share a CU and one's solve
// hides under the others' float4 traffic); a lane solves one column in place in LDS with the elimination factors in registers.  The T
// system goes first: its sweep forms the face diffusivities from the still-unmodified u, v and keeps c·ν for the velocity sweep.
// ------------------------------------------------------------------------------------------------
typedef float co_f32x4 __attribute__((ext_vector_type(4)));
struct MppParams { float nu0, nu_minus, inv_dRi, Ric, inv_Pr, galpha_dz, c; int ca; };

// c·ν and c·νT of face k (1 <= k < Nz) from the level differences across it (du, dv, dT = upper − lower)
__device__ __forceinline__ void mpp_face(const MppParams& P, float du, float dv, float dT, float& kv, float& kT) {
    const float Ri = P.galpha_dz * dT / (du * du + dv * dv);
    const float x = (Ri - P.Ric) * P.inv_dRi;
    const float nu = P.nu0 + P.nu_minus * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x));
    kv = P.c * nu;
    kT = P.ca ? (Ri > 0.0f ? kv * P.inv_Pr : P.c) : kv * P.inv_Pr;
}

template <int NZ>
__global__ void __launch_bounds__(64) mpp_diffusion_kernel(const float* u, const float* v, const float* T,
                                                           const float* __restrict__ halo_bottom, MppParams P, float* uo, float* vo, float* To, int n_col) {
    extern __shared__ float cs_smem[];
    constexpr int LD = NZ + 1, Q = NZ / 4, FS = 64 * LD;
    const int lane = threadIdx.x;
    const int col0 = blockIdx.x * 64;
    const int ncol = min(64, n_col - col0);
    const float* srcs[3] = {u, v, T};
    float* dsts[3] = {uo, vo, To};
    if (ncol == 64) {
        co_f32x4 r[3][Q];
#pragma unroll
        for (int f = 0; f < 3; f++) {
            const co_f32x4* s = reinterpret_cast<const co_f32x4*>(srcs[f] + (size_t)col0 * NZ);
#pragma unroll
            for (int i = 0; i < Q; i++) r[f][i] = __builtin_nontemporal_load(s + i * 64 + lane);
        }
#pragma unroll
        for (int f = 0; f < 3; f++)
#pragma unroll
            for (int i = 0; i < Q; i++) {
                const int e = i * 64 + lane, cl = e / Q, k = (e % Q) * 4;
                float* d = cs_smem + f * FS + cl * LD + k;
                d[0] = r[f][i].x; d[1] = r[f][i].y; d[2] = r[f][i].z; d[3] = r[f][i].w;
            }
    } else {
        for (int f = 0; f < 3; f++) {
            const float4* s = reinterpret_cast<const float4*>(srcs[f] + (size_t)col0 * NZ);
            for (int e = lane; e < ncol * Q; e += 64) {
                const float4 q = s[e];
                const int cl = e / Q, k = (e % Q) * 4;
                float* d = cs_smem + f * FS + cl * LD + k;
                d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
            }
        }
    }
    __syncthreads();
    if (lane < ncol) {
        float* tu = cs_smem + lane * LD;
        float* tv = tu + FS;
        float* tT = tv + FS;
        float kvs[NZ], cp[NZ];
        // face 0: ν = 0; νT under convective adjustment from the Richardson number the halo cells give (absent: zero-gradient fill,
        // 0/0 = NaN, `NaN > 0` false: νT = 1 — what the reference computes for a flux-bounded field)
        float u_lo = tu[0], v_lo = tv[0], T_lo = tT[0];
        const float T_bottom = T_lo;
        float kT_k = 0.0f;
        kvs[0] = 0.0f;
        if (P.ca) {
            const size_t c = (size_t)col0 + lane;
            const float du = halo_bottom ? u_lo - halo_bottom[c] : 0.0f;
            const float dv = halo_bottom ? v_lo - halo_bottom[(size_t)n_col + c] : 0.0f;
            const float dT = halo_bottom ? T_lo - halo_bottom[2 * (size_t)n_col + c] : 0.0f;
            const float Ri0 = P.galpha_dz * dT / (du * du + dv * dv);
            kT_k = Ri0 > 0.0f ? 0.0f : P.c;
        }
        // ---- T system, forming the faces one ahead of the elimination
        float xT = 0.0f;
#pragma unroll
        for (int k = 0; k < NZ; k++) {
            float kT_n = 0.0f;
            const float T_k = T_lo;
            if (k + 1 < NZ) {
                const float u_hi = tu[k + 1], v_hi = tv[k + 1], T_hi = tT[k + 1];
                mpp_face(P, u_hi - u_lo, v_hi - v_lo, T_hi - T_lo, kvs[k + 1], kT_n);
                u_lo = u_hi; v_lo = v_hi; T_lo = T_hi;
            }
            const float a = -kT_k, b = 1.0f + kT_k + kT_n;
            const float inv = 1.0f / (k == 0 ? b : b - a * cp[k - 1]);
            cp[k] = -kT_n * inv;
            xT = (k == 0 ? T_k : T_k - a * xT) * inv;
            tT[k] = xT;
            kT_k = kT_n;
        }
#pragma unroll
        for (int k = NZ - 2; k >= 0; k--) { xT = tT[k] - cp[k] * xT; tT[k] = xT; }
        tT[0] = T_bottom;
        // ---- velocity system, two right-hand sides
        float xu = 0.0f, xv = 0.0f;
#pragma unroll
        for (int k = 0; k < NZ; k++) {
            const float kn = k + 1 < NZ ? kvs[k + 1] : 0.0f;
            const float a = -kvs[k], b = 1.0f + kvs[k] + kn;
            const float inv = 1.0f / (k == 0 ? b : b - a * cp[k - 1]);
            cp[k] = -kn * inv;
            xu = (k == 0 ? tu[k] : tu[k] - a * xu) * inv;
            xv = (k == 0 ? tv[k] : tv[k] - a * xv) * inv;
            tu[k] = xu; tv[k] = xv;
        }
#pragma unroll
        for (int k = NZ - 2; k >= 0; k--) {
            xu = tu[k] - cp[k] * xu; tu[k] = xu;
            xv = tv[k] - cp[k] * xv; tv[k] = xv;
        }
    }
    __syncthreads();
    if (ncol == 64) {
#pragma unroll
        for (int f = 0; f < 3; f++) {
            co_f32x4* dd = reinterpret_cast<co_f32x4*>(dsts[f] + (size_t)col0 * NZ);
#pragma unroll
            for (int i = 0; i < Q; i++) {
                const int e = i * 64 + lane, cl = e / Q, k = (e % Q) * 4;
                const float* d = cs_smem + f * FS + cl * LD + k;
                const co_f32x4 q = {d[0], d[1], d[2], d[3]};
                __builtin_nontemporal_store(q, dd + e);
            }
        }
    } else {
        for (int f = 0; f < 3; f++) {
            float4* dd = reinterpret_cast<float4*>(dsts[f] + (size_t)col0 * NZ);
            for (int e = lane; e < ncol * Q; e += 64) {
                const int cl = e / Q, k = (e % Q) * 4;
                const float* d = cs_smem + f * FS + cl * LD + k;
                dd[e] = make_float4(d[0], d[1], d[2], d[3]);
            }
        }
    }
}

// any 2 <= Nz <= 128 (not a multiple of 4, unaligned pointers, or none of the instantiated sizes): one thread per column from HBM
__global__ void __launch_bounds__(64) mpp_diffusion_generic_kernel(const float* u, const float* v, const float* T, const float* __restrict__ halo_bottom,
                                                                   MppParams P, float* uo, float* vo, float* To, int Nz, int n_col) {
    const int col = blockIdx.x * 64 + threadIdx.x;
    if (col >= n_col) return;
    const float* pu = u + (size_t)col * Nz;
    const float* pv = v + (size_t)col * Nz;
    const float* pT = T + (size_t)col * Nz;
    float kv[128], kT[128], cp[128], xa[128], xb[128];
    kv[0] = 0.0f;
    kT[0] = 0.0f;
    if (P.ca) {
        const float du = halo_bottom ? pu[0] - halo_bottom[col] : 0.0f;
        const float dv = halo_bottom ? pv[0] - halo_bottom[(size_t)n_col + col] : 0.0f;
        const float dT = halo_bottom ? pT[0] - halo_bottom[2 * (size_t)n_col + col] : 0.0f;
        const float Ri0 = P.galpha_dz * dT / (du * du + dv * dv);
        kT[0] = Ri0 > 0.0f ? 0.0f : P.c;
    }
    for (int k = 1; k < Nz; k++) mpp_face(P, pu[k] - pu[k - 1], pv[k] - pv[k - 1], pT[k] - pT[k - 1], kv[k], kT[k]);
    const float T_bottom = pT[0];
    // T
    for (int k = 0; k < Nz; k++) {
        const float kn = k + 1 < Nz ? kT[k + 1] : 0.0f;
        const float a = -kT[k], b = 1.0f + kT[k] + kn;
        const float inv = 1.0f / (k == 0 ? b : b - a * cp[k - 1]);
        cp[k] = -kn * inv;
        xa[k] = (k == 0 ? pT[k] : pT[k] - a * xa[k - 1]) * inv;
    }
    for (int k = Nz - 2; k >= 0; k--) xa[k] -= cp[k] * xa[k + 1];
    xa[0] = T_bottom;
    for (int k = 0; k < Nz; k++) To[(size_t)col * Nz + k] = xa[k];
    // u, v
    for (int k = 0; k < Nz; k++) {
        const float kn = k + 1 < Nz ? kv[k + 1] : 0.0f;
        const float a = -kv[k], b = 1.0f + kv[k] + kn;
        const float inv = 1.0f / (k == 0 ? b : b - a * cp[k - 1]);
        cp[k] = -kn * inv;
        xa[k] = (k == 0 ? pu[k] : pu[k] - a * xa[k - 1]) * inv;
        xb[k] = (k == 0 ? pv[k] : pv[k] - a * xb[k - 1]) * inv;
    }
    for (int k = Nz - 2; k >= 0; k--) { xa[k] -= cp[k] * xa[k + 1]; xb[k] -= cp[k] * xb[k + 1]; }
    for (int k = 0; k < Nz; k++) { uo[(size_t)col * Nz + k] = xa[k]; vo[(size_t)col * Nz + k] = xb[k]; }
}

hipError_t launch_mpp_diffusion(const float* u, const float* v, const float* T, const float* halo_bottom, float dt, float dz,
                                const float params[7], int convective_adjustment, float* uo, float* vo, float* To, int Nz, int n_col,
                                hipStream_t stream) {
    if (Nz < 2 || Nz > 128 || n_col < 1) return hipErrorInvalidValue;
    MppParams P;
    P.nu0 = params[0]; P.nu_minus = params[1]; P.inv_dRi = 1.0f / params[2]; P.Ric = params[3]; P.inv_Pr = 1.0f / params[4];
    P.galpha_dz = params[5] * params[6] * dz;       // ∂z b / ((∂z u)² + (∂z v)²) = gα ΔT Δz / (Δu² + Δv²)
    P.c = dt / (dz * dz);
    P.ca = convective_adjustment ? 1 : 0;
    const dim3 grid((n_col + 63) / 64), block(64);
    const bool aligned = (((uintptr_t)u | (uintptr_t)v | (uintptr_t)T | (uintptr_t)uo | (uintptr_t)vo | (uintptr_t)To) & 15) == 0;
    // in-place use is allowed field by field (uo == u etc.); any other overlap between the six arrays is the caller's error
#define MPP_LAUNCH(N) hipLaunchKernelGGL(mpp_diffusion_kernel<N>, grid, block, 3 * 64 * (N + 1) * sizeof(float), stream, u, v, T, halo_bottom, P, uo, vo, To, n_col)
    if (aligned && Nz == 16) MPP_LAUNCH(16);
    else if (aligned && Nz == 32) MPP_LAUNCH(32);
    else if (aligned && Nz == 64) MPP_LAUNCH(64);
    else hipLaunchKernelGGL(mpp_diffusion_generic_kernel, grid, block, 0, stream, u, v, T, halo_bottom, P, uo, vo, To, Nz, n_col);
#undef MPP_LAUNCH
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// loss_per_tstep (wind_mixing/src/loss.jl:44-46; called on the six profile matrices of one simulation at training_postprocessing.jl:311-316)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) loss_per_tstep_kernel(const float* __restrict__ sol, const float* __restrict__ truth, int n_col, int n_save, int Nz,
                                                             int n_var, float* __restrict__ out) {
    const long total = (long)n_col * n_save * n_var;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
        const int v = (int)(it % n_var);
        const long cs = it / n_var;                        // column * n_save + save point
        const int sp = (int)(cs % n_save);
        const long c = cs / n_save;
        const float* a = sol + (cs * n_var + v) * Nz;
        const float* b = truth + (cs * n_var + v) * Nz;
        float sp2 = 0.0f, sg2 = 0.0f, prev = 0.0f;
        for (int k = 0; k < Nz; k++) {
            const float d = a[k] - b[k];
            sp2 += d * d;
            if (k > 0) { const float g = (d - prev) * (float)Nz; sg2 += g * g; }      // D^f: (x[k] - x[k-1]) Nz on the interior faces, 0 on the two end faces
            prev = d;
        }
        const int term = n_var == 3 ? v : 2;
        out[(c * 6 + term) * n_save + sp] = sp2 / (float)Nz;
        out[(c * 6 + 3 + term) * n_save + sp] = sg2 / (float)(Nz + 1);
    }
}

hipError_t launch_loss_per_tstep(const float* sol, const float* truth, int n_col, int n_save, int Nz, int n_var, float* out, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(out, 0, (size_t)n_col * 6 * n_save * sizeof(float), stream);
    if (e != hipSuccess) return e;
    const long total = (long)n_col * n_save * n_var;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(loss_per_tstep_kernel, dim3(blocks), dim3(256), 0, stream, sol, truth, n_col, n_save, Nz, n_var, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// The adaptive integrator's error norm of a difference (the Richardson estimate of colnde_error_estimate): per state vector (row) the RMS over its
// components of (a_i - b_i) / (floor + |b_i|) — OrdinaryDiffEq's internalnorm of err / (abstol + reltol |u|), divided through by reltol —, then the
// maximum over the rows (columns x save points)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) rel_diff_max_kernel(const float* __restrict__ a, const float* __restrict__ b, long n_rows, int row, float floor,
                                                           float* __restrict__ partial) {
    float mx = 0.0f;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < n_rows; r += (long)gridDim.x * 256) {
        float s2 = 0.0f;
        for (int i = 0; i < row; i++) {
            const float q = (a[r * row + i] - b[r * row + i]) / (floor + fabsf(b[r * row + i]));
            s2 += q * q;
        }
        const float v = sqrtf(s2 / (float)row);
        mx = (v <= 3.0e38f) ? fmaxf(mx, v) : __int_as_float(0x7f800000);              // a NaN or Inf anywhere: +inf
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_down(mx, off));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__global__ void __launch_bounds__(64) rel_diff_finish_kernel(const float* __restrict__ partial, int nb, float* __restrict__ out) {
    float mx = 0.0f;
    for (int i = threadIdx.x; i < nb; i += 64) mx = fmaxf(mx, partial[i]);
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_down(mx, off));
    if (threadIdx.x == 0) out[0] = mx;
}

hipError_t launch_rel_diff_max(const float* a, const float* b, long n_rows, int row, float floor, float* partial, float* out, hipStream_t stream) {
    const int blocks = (int)((n_rows + 255) / 256 < 1024 ? (n_rows + 255) / 256 : 1024);
    hipLaunchKernelGGL(rel_diff_max_kernel, dim3(blocks), dim3(256), 0, stream, a, b, n_rows, row, floor, partial);
    hipLaunchKernelGGL(rel_diff_finish_kernel, dim3(1), dim3(64), 0, stream, partial, blocks, out);
    return hipGetLastError();
}
