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
    int splits;               // producer / consumer kernel only: > 1 = split-K, partial tiles added into an fp32 C with atomics
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

// EPI_GATED_RES gates of one output tile, hoisted out of the per-row work: a tile of `rows` <= S rows touches at most the
// samples b0 = row0 / S and b0 + 1, so their (text, video) gate vectors for this thread's 4 columns are loaded once per
// tile and a row only selects among them (no per-row integer division, no dependent load).  fast = 0 (S smaller than a
// tile, or no gates): the generic per-row lookup runs.
struct GateCtx {
    f32x4 g[2][2];      // [sample b0, b0+1][text, video]
    int base0;          // b0 * S
    int fast;
};
__device__ __forceinline__ GateCtx gate_ctx_load(const GemmParams& p, int row0, int rows, int n) {
    GateCtx c;
    c.fast = (p.gate_vid != nullptr && p.S >= rows && n < p.N) ? 1 : 0;
    c.base0 = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) c.g[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (c.fast) {
        const int b0 = row0 / p.S;
        const int bl = (p.M - 1) / p.S;                  // last sample
        const int b1 = b0 + 1 <= bl ? b0 + 1 : bl;
        c.base0 = b0 * p.S;
        c.g[0][0] = *(const f32x4*)(p.gate_txt + (size_t)b0 * p.gate_bstride + n);
        c.g[0][1] = *(const f32x4*)(p.gate_vid + (size_t)b0 * p.gate_bstride + n);
        c.g[1][0] = *(const f32x4*)(p.gate_txt + (size_t)b1 * p.gate_bstride + n);
        c.g[1][1] = *(const f32x4*)(p.gate_vid + (size_t)b1 * p.gate_bstride + n);
    }
    return c;
}

// v = accumulators of (row m, columns n..n+3), bias4 = bias of those columns (zeros if none).  Caller guarantees
// m < M and n < N (N % 4 == 0).
template <int EPI, bool OUT_F32>
__device__ __forceinline__ void gemm_epilogue_store_aux(const GemmParams& p, int m, int n, f32x4 v, const float* bias4, u32x2 aux,
                                                        const GateCtx* gc = nullptr) {
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
                    if (gc != nullptr && gc->fast) {
                        int sr = m - gc->base0;
                        const bool second = sr >= p.S;
                        sr -= second ? p.S : 0;
                        const bool txt = sr < p.St;
                        const f32x4 ga = txt ? gc->g[0][0] : gc->g[0][1], gb = txt ? gc->g[1][0] : gc->g[1][1];
                        const f32x4 g4 = second ? gb : ga;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = r[j] + g4[j] * o[j];
                    } else if (p.gate_vid != nullptr) {
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
                    if (p.splits > 1) {            // split-K partial tile: C was zeroed by the host entry point
                        float* c = (float*)p.C + (size_t)m * p.ldc + n;
#pragma unroll
                        for (int j = 0; j < 4; ++j) atomicAdd(c + j, o[j]);
                        return;
                    }
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
