// Weight-gradient GEMM for gfx950:  C[P,Q] (+)= sum_m A[m,P] * B[m,Q]   (bf16 operands, fp32 result).
//
// This is dW = dY^T X of every nn.Linear on the finetune path (full fine-tuning, BASELINE config 3): what
// loss.backward() computes for the Linear weights of diffusers' CogVideoXBlock, reached by the reference through
// videotuna/models/cogvideo_hf/cogvideo_pl.py:865-871 (SURVEY 8(a) a4,a5,a9).  The reduction runs over the TOKEN
// axis m (the slow index of both operands), so neither operand is K-contiguous: tiles are staged row-major as they
// lie in HBM ([64 tokens][128 columns], 256-byte rows, LDS-DMA with the swizzle on the source side) and both MFMA
// fragments are produced by transposed LDS reads (ds_read_b64_tr_b16) -- no transposed copy of dY or X ever exists.
//
// 128x128 output tile per workgroup, 4 waves (2x2) x 4x4 tiles of v_mfma_f32_16x16x32_bf16, 64 tokens per K-tile,
// double-buffered.  Tokens past M read zeros (buffer bounds check).  P and Q must be multiples of 128.
#include "common.h"
#include <cstdlib>

struct GemmNtParams {
    const bf16_t* A;   // [M, lda]  (dY)
    const bf16_t* B;   // [M, ldb]  (X)
    float* C;          // [P, ldc]
    int M, P, Q, lda, ldb, ldc;
    int accumulate;    // 1: C += result, 0: C = result
    float alpha;
    int splits;        // > 1: the token axis is cut into `splits` ranges of m_chunk tokens (blockIdx.y), partial tiles are added
    int m_chunk;       //      to C with fp32 atomics (C already holds the value to accumulate onto, or zeros)
};

typedef __attribute__((ext_vector_type(8))) short short8nt;

// 256-byte-row image, 16-byte chunk ch of row `row`; the XOR keeps the transposed reads of both MFMA operands
// conflict-free (a half-wave reads two 4-row blocks that are 8 rows apart in the same 16 columns)
__device__ __forceinline__ int nt_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ bf16x8 nt_tr_pair(const char* p0, const char* p1) {
    short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(p0));
    short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(p1));
    short8nt v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v8);
}

__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmNtParams p) {
    __shared__ __attribute__((aligned(16))) char smem[65536];     // 2 stages x (A 16 KiB + B 16 KiB)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave & 1, wq = wave >> 1;
    const int nbp = p.P / 128, nbq = p.Q / 128;
    const int id = xcd_remap(blockIdx.x, nbp * nbq);
    const int tile_p = id % nbp, tile_q = id / nbp;
    const int p0 = tile_p * 128, q0 = tile_q * 128;

    // this workgroup's token range (split-K over the token axis: dW has few output tiles -- 675 for the fused QKV weight --
    // and a 35 552-long reduction, so one workgroup per tile leaves a third of the chip idle in the last round)
    const int m_lo = (int)blockIdx.y * p.m_chunk;
    const int m_cnt = min(p.M - m_lo, p.m_chunk);
    const bf16_t* Ab = p.A + (size_t)m_lo * p.lda;
    const bf16_t* Bb = p.B + (size_t)m_lo * p.ldb;
    const long long a_bytes = (long long)m_cnt * p.lda * 2, b_bytes = (long long)m_cnt * p.ldb * 2;
    __amdgpu_buffer_rsrc_t ra = make_rsrc(Ab + p0, (unsigned)((a_bytes - p0 * 2) > 0x7fffffffLL ? 0x7fffffffLL : (a_bytes - p0 * 2)));
    __amdgpu_buffer_rsrc_t rb = make_rsrc(Bb + q0, (unsigned)((b_bytes - q0 * 2) > 0x7fffffffLL ? 0x7fffffffLL : (b_bytes - q0 * 2)));

    // LDS-DMA: a tile is 64 rows x 256 B = 16 blocks of 1 KiB (4 rows each); wave w moves blocks w, w+4, w+8, w+12 of A and B.
    // lane l lands at (row l>>4, physical chunk l&15) -> it fetches logical chunk (l&15) ^ swz(row).
    const int drl = lane >> 4, dcp = lane & 15;
    int a_voff[4], b_voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 4 * (wave + 4 * j) + drl;
        const int ch = dcp ^ nt_swz(row);
        a_voff[j] = row * p.lda * 2 + ch * 16;
        b_voff[j] = row * p.ldb * 2 + ch * 16;
    }
    auto dma = [&](int kt, int buf) {
        // token offset of this K-tile; kept in 32-bit byte offsets (host checks M*ld*2 < 2^31)
        const int sa = kt * 64 * p.lda * 2, sb = kt * 64 * p.ldb * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            char* dst = smem + buf * 32768 + (wave + 4 * j) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)dst, 16, a_voff[j], sa, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(dst + 16384), 16, b_voff[j], sb, 0, 0);
        }
    };

    // transposed-read offsets: lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3 of its block
    const int g = lane >> 4, ql = (lane & 15) >> 2, pl = lane & 3;
    auto tr_addr = [&](const char* img, int slab_col0, int ks, int sec, int t) {
        const int row = ks * 32 + 8 * g + ql + 4 * sec;
        const int col = slab_col0 + t * 16 + 4 * pl;
        return img + row * 256 + (((col >> 3) ^ nt_swz(row)) << 4) + (col & 7) * 2;
    };

    f32x4 acc[4][4];     // [tp][tq]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (m_cnt + 63) / 64;
    dma(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) dma(kt + 1, buf ^ 1);
        const char* As = smem + buf * 32768;
        const char* Bs = As + 16384;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[t] = nt_tr_pair(tr_addr(As, wp * 64, ks, 0, t), tr_addr(As, wp * 64, ks, 1, t));
                bfr[t] = nt_tr_pair(tr_addr(Bs, wq * 64, ks, 0, t), tr_addr(Bs, wq * 64, ks, 1, t));
            }
#pragma unroll
            for (int tp = 0; tp < 4; ++tp)
#pragma unroll
                for (int tq = 0; tq < 4; ++tq)
                    acc[tp][tq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tp], bfr[tq], acc[tp][tq], 0, 0, 0);
        }
        __syncthreads();
    }
    // epilogue: D[i = 4g + reg][j = lane & 15]; each register row is 16 consecutive fp32 columns (64 B) of C
    const int fr = lane & 15;
#pragma unroll
    for (int tp = 0; tp < 4; ++tp)
#pragma unroll
        for (int tq = 0; tq < 4; ++tq)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int pr = p0 + wp * 64 + tp * 16 + 4 * g + rg;
                const int qc = q0 + wq * 64 + tq * 16 + fr;
                float* c = p.C + (size_t)pr * p.ldc + qc;
                const float v = p.alpha * acc[tp][tq][rg];
                if (p.splits > 1) atomicAdd(c, v);
                else *c = p.accumulate ? (*c + v) : v;
            }
}

extern "C" int vt_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int P, int Q,
                               float alpha, int accumulate, void* stream) {
    if (M <= 0 || P <= 0 || Q <= 0 || (P % 128) || (Q % 128) || (lda % 8) || (ldb % 8) || lda < P || ldb < Q || ldc < Q)
        return VT_ERR_BAD_SHAPE;
    if ((long long)M * lda * 2 >= 0x7fffffffLL || (long long)M * ldb * 2 >= 0x7fffffffLL) return VT_ERR_BAD_SHAPE;
    if ((((uintptr_t)A) | ((uintptr_t)B)) & 15) return VT_ERR_BAD_ALIGN;
    GemmNtParams p{(const bf16_t*)A, (const bf16_t*)B, C, M, P, Q, lda, ldb, ldc, accumulate, alpha, 1, M};
    // split the token axis so that the grid fills whole rounds of the chip (2 workgroups per CU); ranges stay >= 2048 tokens
    const int tiles = (P / 128) * (Q / 128);
    static int wg_slots = 0;
    if (wg_slots == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        wg_slots = cus > 0 ? 2 * cus : 512;
    }
    int best = 1;
    double best_eff = 0.0;
    for (int sp = 1; sp <= 8; ++sp) {
        if (sp > 1 && M / sp < 2048) break;
        const long long wgs = (long long)tiles * sp;
        const double eff = (double)wgs / (double)(((wgs + wg_slots - 1) / wg_slots) * wg_slots);
        if (eff > best_eff + 0.03) { best_eff = eff; best = sp; }
    }
    if (const char* e = getenv("VT_NT_SPLITS")) { const int v = atoi(e); if (v >= 1 && v <= 16) best = v; }
    p.splits = best;
    p.m_chunk = ((M + best - 1) / best + 63) / 64 * 64;
    hipStream_t st = (hipStream_t)stream;
    if (best > 1 && !accumulate) {
        // partial tiles are ADDED: start from zeros
        if (hipMemset2DAsync(C, (size_t)ldc * 4, 0, (size_t)Q * 4, (size_t)P, st) != hipSuccess) return VT_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(gemm_nt_kernel, dim3((P / 128) * (Q / 128), best), dim3(256), 0, st, p);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}
