// Shared by the TN GEMM kernels (gemm_bf16.hip: 128x128 tile, gemm_big_bf16.hip: 256x256 tile): problem descriptor and
// the fused epilogue applied to 4 consecutive output columns of one row.
#pragma once
#include "common.h"

struct GemmParams {
    const bf16_t* A;
    const bf16_t* W;
    void* C;
    const bf16_t* bias;       // [N] or null
    const bf16_t* R;          // residual [*, ldr] (EPI_GATED_RES)
    const float* gate_txt;    // fp32 gate for rows with (m % S) <  St, indexed [b*gate_bstride + n]
    const float* gate_vid;    // fp32 gate for rows with (m % S) >= St
    bf16_t* C2;               // second output: pre-activation (EPI_BIAS_GELU)
    const bf16_t* U;          // saved pre-activation (EPI_DGELU)
    int M, N, K;
    int lda, ldw, ldc, ldr, ldc2, ldu;
    int S, St, gate_bstride;  // rows per sample, text rows per sample, gate batch stride (elements)
    int r_mod;                // if > 0 the residual row is (m % r_mod)  (positional table add)
};

enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_GATED_RES = 2, EPI_DGELU = 3 };


// The epilogues that read a second operand (EPI_GATED_RES: residual R, EPI_DGELU: saved pre-activation U) take its four
// bf16 values of (row m, columns n..n+3) as `aux`, so that a kernel can fetch them a slab ahead of their use.
template <int EPI>
__device__ __forceinline__ u32x2 gemm_epilogue_aux_load(const GemmParams& p, int m, int n) {
    if (EPI == EPI_GATED_RES) {
        const int rr = p.r_mod > 0 ? (m % p.r_mod) : m;
        return *(const u32x2*)(p.R + (size_t)rr * p.ldr + n);
    } else if (EPI == EPI_DGELU) {
        return *(const u32x2*)(p.U + (size_t)m * p.ldu + n);
    }
    return (u32x2){0u, 0u};
}

// v = accumulators of (row m, columns n..n+3), bias4 = bias of those columns (zeros if none).  Caller guarantees
// m < M and n < N (N % 4 == 0).
template <int EPI, bool OUT_F32>
__device__ __forceinline__ void gemm_epilogue_store_aux(const GemmParams& p, int m, int n, f32x4 v, const float* bias4, u32x2 aux) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = v[j] + bias4[j];
                if (EPI == EPI_BIAS_GELU) {
                    u32x2 u2;
                    u2[0] = pack2(o[0], o[1]);
                    u2[1] = pack2(o[2], o[3]);
                    *(u32x2*)(p.C2 + (size_t)m * p.ldc2 + n) = u2;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = gelu_tanh_f(o[j]);
                } else if (EPI == EPI_GATED_RES) {
                    if (p.C2 != nullptr) {           // branch output before gating (full fine-tune: d gate = sum dh * branch)
                        u32x2 u2;
                        u2[0] = pack2(o[0], o[1]);
                        u2[1] = pack2(o[2], o[3]);
                        *(u32x2*)(p.C2 + (size_t)m * p.ldc2 + n) = u2;
                    }
                    const u32x2 r2 = aux;
                    float r[4] = {__uint_as_float(r2[0] << 16), __uint_as_float(r2[0] & 0xffff0000u),
                                  __uint_as_float(r2[1] << 16), __uint_as_float(r2[1] & 0xffff0000u)};
                    if (p.gate_vid != nullptr) {
                        const int b = m / p.S;
                        const int s = m - b * p.S;
                        const float* g = (s < p.St ? p.gate_txt : p.gate_vid) + (size_t)b * p.gate_bstride + n;
                        f32x4 g4 = *(const f32x4*)g;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = r[j] + g4[j] * o[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = r[j] + o[j];
                    }
                } else if (EPI == EPI_DGELU) {
                    const u32x2 u2 = aux;
                    float u[4] = {__uint_as_float(u2[0] << 16), __uint_as_float(u2[0] & 0xffff0000u),
                                  __uint_as_float(u2[1] << 16), __uint_as_float(u2[1] & 0xffff0000u)};
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = o[j] * gelu_tanh_grad_f(u[j]);
                }
                if (OUT_F32) {
                    *(f32x4*)((float*)p.C + (size_t)m * p.ldc + n) = (f32x4){o[0], o[1], o[2], o[3]};
                } else {
                    u32x2 c2;
                    c2[0] = pack2(o[0], o[1]);
                    c2[1] = pack2(o[2], o[3]);
                    *(u32x2*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = c2;
                }
}

template <int EPI, bool OUT_F32>
__device__ __forceinline__ void gemm_epilogue_store(const GemmParams& p, int m, int n, f32x4 v, const float* bias4) {
    gemm_epilogue_store_aux<EPI, OUT_F32>(p, m, n, v, bias4, gemm_epilogue_aux_load<EPI>(p, m, n));
}
