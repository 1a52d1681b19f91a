// Probe: the error of fp32 matrix products on the bf16 pipe from exact three-way operand splitting (csrc/split_bf16.h), beside the fp32 MFMA's own.
// One wave computes C (32 x 32) = A (32 x K) B (K x 32) three ways:
//   (1) v_mfma_f32_32x32x2_f32 chain            (what the default kernels do; bitwise an fmaf chain in k order)
//   (2) six bf16 products per 16-deep k-block    (hh, hm, mh, hl, lh, mm — the opt-in kernels)
//   (3) all nine                                 (adds ml, lm, ll)
// and the host computes it in float64.  Reported: max and rms of |C - C64| / sum_k |a||b| (the natural scale of the rounding error bound), for
// several operand distributions (uniform, wide dynamic range, cancelling sums, tiny and huge magnitudes) and K = 16 .. 4096.
// build: hipcc -O3 --offload-arch=gfx950 -I ../../climateparameterizations.jl_amd/csrc split_error.hip -o split_error_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "split_bf16.h"

// A row-major [32][K], B row-major [K][32]; lane (m = lane & 31, kh = lane >> 5)
__global__ void __launch_bounds__(64) k_fp32(const float* A, const float* B, int K, float* C) {
    const int lane = threadIdx.x, m = lane & 31, kh = lane >> 5;
    sp_f32x16 acc = (sp_f32x16)(0.0f);
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m * K + k + kh], B[(k + kh) * 32 + m], acc, 0, 0, 0);
    for (int r = 0; r < 16; r++) C[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + m] = acc[r];
}

template <int NINE>
__global__ void __launch_bounds__(64) k_split(const float* A, const float* B, int K, float* C) {
    const int lane = threadIdx.x, m = lane & 31, kh = lane >> 5;
    sp_f32x16 acc = (sp_f32x16)(0.0f);
    for (int k = 0; k < K; k += 16) {
        float a8[8], b8[8];
        for (int i = 0; i < 8; i++) { a8[i] = A[m * K + k + 8 * kh + i]; b8[i] = B[(k + 8 * kh + i) * 32 + m]; }
        const Bf3 a = bf3_split8(a8), b = bf3_split8(b8);
        if (NINE) {
            acc = mfma_bf(a.l, b.l, acc);
            acc = mfma_bf(a.m, b.l, acc);
            acc = mfma_bf(a.l, b.m, acc);
        }
        acc = mfma_bf3(a, b, acc);
    }
    for (int r = 0; r < 16; r++) C[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + m] = acc[r];
}

static double urand() { return (rand() + 0.5) / (RAND_MAX + 1.0); }

int main() {
    const char* names[] = {"uniform [-1, 1]", "log-uniform magnitudes 1e-6 .. 1e6, random sign", "cancelling: uniform [-1, 1] with every row of A summing to ~0 against B = 1 + 1e-3 u",
                           "tiny: 1e-30 x uniform", "huge: 1e15 x uniform", "weights ~ N(0, 0.1) against activations in [0, 4] (an MLP layer)"};
    printf("%-88s %5s | %-23s | %-23s | %-23s\n", "operands", "K", "fp32 MFMA  max / rms", "split, 6 products", "split, 9 products");
    for (int d = 0; d < 6; d++)
        for (int K : {16, 96, 256, 1024, 4096}) {
            std::vector<float> A(32 * K), B(K * 32);
            srand(1234 + 17 * d + K);
            for (int i = 0; i < 32 * K; i++) {
                double a = 2 * urand() - 1, b = 2 * urand() - 1;
                if (d == 1) { a = (urand() < 0.5 ? -1 : 1) * pow(10.0, 12 * urand() - 6); b = (urand() < 0.5 ? -1 : 1) * pow(10.0, 12 * urand() - 6); }
                if (d == 2) b = 1.0 + 1e-3 * b;
                if (d == 3) { a *= 1e-30; }
                if (d == 4) { a *= 1e15; b *= 1e15; }
                if (d == 5) { a = 0.1 * sqrt(-2 * log(urand())) * cos(6.283185307179586 * urand()); b = 4 * urand(); }
                A[i] = (float)a; B[i] = (float)b;
            }
            if (d == 2)
                for (int m = 0; m < 32; m++) {
                    double s = 0;
                    for (int k = 0; k < K; k++) s += A[m * K + k];
                    for (int k = 0; k < K; k++) A[m * K + k] = (float)(A[m * K + k] - s / K);
                }
            float *dA, *dB, *dC;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 3 * 1024 * 4);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k_fp32, dim3(1), dim3(64), 0, 0, dA, dB, K, dC);
            hipLaunchKernelGGL(k_split<0>, dim3(1), dim3(64), 0, 0, dA, dB, K, dC + 1024);
            hipLaunchKernelGGL(k_split<1>, dim3(1), dim3(64), 0, 0, dA, dB, K, dC + 2048);
            std::vector<float> C(3 * 1024);
            hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
            double mx[3] = {0, 0, 0}, sq[3] = {0, 0, 0};
            for (int i = 0; i < 32; i++)
                for (int j = 0; j < 32; j++) {
                    double ref = 0, scale = 0;
                    for (int k = 0; k < K; k++) { const double p = (double)A[i * K + k] * (double)B[k * 32 + j]; ref += p; scale += fabs(p); }
                    for (int v = 0; v < 3; v++) {
                        const double e = fabs((double)C[v * 1024 + i * 32 + j] - ref) / scale;
                        mx[v] = fmax(mx[v], e); sq[v] += e * e;
                    }
                }
            printf("%-88s %5d | %.3e / %.3e | %.3e / %.3e | %.3e / %.3e\n", names[d], K, mx[0], sqrt(sq[0] / 1024), mx[1], sqrt(sq[1] / 1024), mx[2], sqrt(sq[2] / 1024));
            hipFree(dA); hipFree(dB); hipFree(dC);
        }
    printf("(2^-24 = %.3e: one fp32 rounding)\n", pow(2.0, -24));
    return 0;
}
