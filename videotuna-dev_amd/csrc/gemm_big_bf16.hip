// bf16 MFMA GEMM for gfx950, large-tile version:  C[M,N] = A[M,K] * W[N,K]^T (+ fused epilogue), fp32 accumulate.
//
// Same contract and epilogues as gemm_bf16.hip (the nn.Linear layers of diffusers' CogVideoXBlock reached through
// videotuna/models/cogvideo_hf/cogvideo_pl.py:865-871, SURVEY 8(a) a4,a5); used for the token-sized problems.
//
// Why a second tiling: a wave that owns a 64x64 output tile reads (64+64) x 64 bf16 = 16 KiB of LDS per 64-deep
// K-tile for 512 Ki flop -- 32 flop per LDS byte, exactly the ratio of the CU's MFMA rate (4096 flop/clk) to its LDS
// bandwidth (128 B/clk), so the 128x128 kernel can never run the matrix cores above ~50 %.  Here the workgroup tile is
// 256x256x64 with 8 waves (2 x 4), each wave owning 128x64 = 4x2 tiles of v_mfma_f32_32x32x16_bf16 (128 accumulator
// registers, two waves per SIMD): 24 KiB of LDS reads per wave per K-tile for 1 Mi flop -> the LDS pipe is ~75 % busy
// at full MFMA rate.  Operands are staged by LDS-DMA (buffer_load ... lds, XOR swizzle applied to the source chunk),
// double-buffered (2 x 64 KiB); the accumulators leave through LDS in four 64-row slabs so that the epilogue reads
// residuals / writes C in row-contiguous 512-B segments.
#include "gemm_epilogue.h"

#ifndef VT_SUFFIX
#define VT_SUFFIX
#endif
#define VT_CAT_(a, b) a##b
#define VT_CAT(a, b) VT_CAT_(a, b)
#define GEMM_BIG_KERNEL VT_CAT(gemm_tn_big_kernel, VT_SUFFIX)
#define GEMM_BIG_ENTRY VT_CAT(vt_gemm_bf16_big, VT_SUFFIX)

#define GB_BM 256
#define GB_BN 256
#define GB_BK 64
#define GB_STAGE 65536      // A 32 KiB | W 32 KiB
#define GB_CS_LD 260        // fp32 row stride of the epilogue staging slab (64 rows x 260 floats = 65 KiB)

template <int EPI, bool OUT_F32>
__global__ __launch_bounds__(512, 1) void GEMM_BIG_KERNEL(GemmParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * GB_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;                  // 2 x 4 waves: rows 128 wm .., columns 64 wn ..
    const int nbm = (p.M + GB_BM - 1) / GB_BM, nbn = (p.N + GB_BN - 1) / GB_BN;
    // XCD-aware id, then grouped ordering (4 row-tiles per group, column-tiles outer): the 32 workgroups of an XCD
    // that run together share A row-panels / W column-panels in its L2
    const int id = xcd_remap(blockIdx.x, nbm * nbn);
    const int GM = 4;
    const int in_group = GM * nbn;
    const int group = id / in_group;
    const int first_m = group * GM;
    const int gsz = min(nbm - first_m, GM);
    const int tile_m = first_m + (id % in_group) % gsz;
    const int tile_n = (id % in_group) / gsz;
    const int row0 = tile_m * GB_BM, col0 = tile_n * GB_BN;

    const long long a_rem = (long long)(p.M - row0) * p.lda * 2;
    const long long w_rem = (long long)(p.N - col0) * p.ldw * 2;
    __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A + (size_t)row0 * p.lda, (unsigned)(a_rem > 0x7fffffffLL ? 0x7fffffffLL : a_rem));
    __amdgpu_buffer_rsrc_t rw = make_rsrc(p.W + (size_t)col0 * p.ldw, (unsigned)(w_rem > 0x7fffffffLL ? 0x7fffffffLL : w_rem));

    // ---- LDS-DMA staging: a K-tile of A (and of W) is 32 blocks of 1 KiB = 8 rows x 128 B; wave w moves blocks
    // w, w+8, w+16, w+24 of each operand.  Lane l lands at (row l>>3, physical chunk l&7) of its block and fetches
    // logical chunk (l&7) ^ (row&7); rows past M / N read zeros (bounds check).
    const int drl = lane >> 3, dcp = lane & 7;
    int a_voff[4], w_voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * (wave + 8 * j) + drl;
        a_voff[j] = row * p.lda * 2 + ((dcp ^ drl) << 4);
        w_voff[j] = row * p.ldw * 2 + ((dcp ^ drl) << 4);
    }
    auto dma = [&](int kt, int buf) {
        const int soff = kt * GB_BK * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            char* dst = smem + buf * GB_STAGE + (wave + 8 * j) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)dst, 16, a_voff[j], soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(dst + 32768), 16, w_voff[j], soff, 0, 0);
        }
    };

    // acc[tn][tm] = D[n][m] of (W-fragment, A-fragment): lane holds m = lane & 31 and, in register r, column
    // n = 8 (r>>2) + 4 (lane>>5) + (r&3): four consecutive output columns per register quad
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = p.K / GB_BK;
    dma(0, 0);
    __syncthreads();
    const int fr = lane & 31;             // row inside a 32-row fragment
    const int fh = lane >> 5;             // which 8-element half of a 16-deep k-step
    const int fx = lane & 7;              // == (row & 7) of every fragment row this lane reads
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) dma(kt + 1, buf ^ 1);
        const char* As = smem + buf * GB_STAGE + (wm * 128 + fr) * 128;
        const char* Ws = smem + buf * GB_STAGE + 32768 + (wn * 64 + fr) * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int coff = (((ks * 2 + fh) ^ fx) << 4);
            bf16x8 af[4], wf[2];
#pragma unroll
            for (int t = 0; t < 4; ++t) af[t] = *(const bf16x8*)(As + t * 4096 + coff);
#pragma unroll
            for (int t = 0; t < 2; ++t) wf[t] = *(const bf16x8*)(Ws + t * 4096 + coff);
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tn], af[tm], acc[tn][tm], 0, 0, 0);
        }
        __syncthreads();       // its vmcnt(0) also retires the next tile's LDS-DMA
    }

    // ---------------- epilogue: four 64-row slabs through LDS ----------------
    float* Cs = (float*)smem;
    const int er = tid >> 6;              // 0..7 : row inside an 8-row pass
    const int ec = (tid & 63) * 4;        // first of this thread's 4 columns
    const int n = col0 + ec;
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr && n < p.N) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bias4[j] = bf2f(p.bias[n + j]);
    }
#pragma unroll
    for (int slab = 0; slab < 4; ++slab) {
        if (wm == (slab >> 1)) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const int tm = 2 * (slab & 1) + t2;
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int ml = t2 * 32 + fr;
                        const int nl = wn * 64 + tn * 32 + 8 * q + 4 * fh;
                        *(f32x4*)(Cs + ml * GB_CS_LD + nl) =
                            (f32x4){acc[tn][tm][4 * q], acc[tn][tm][4 * q + 1], acc[tn][tm][4 * q + 2], acc[tn][tm][4 * q + 3]};
                    }
            }
        }
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int ml = pass * 8 + er;
            const int m = row0 + slab * 64 + ml;
            if (m < p.M && n < p.N) {
                const f32x4 v = *(const f32x4*)(Cs + ml * GB_CS_LD + ec);
                gemm_epilogue_store<EPI, OUT_F32>(p, m, n, v, bias4);
            }
        }
        __syncthreads();
    }
}

template <int EPI, bool F32>
static int VT_CAT(launch_big, VT_SUFFIX)(const GemmParams& p, hipStream_t st) {
    const int nbm = (p.M + GB_BM - 1) / GB_BM, nbn = (p.N + GB_BN - 1) / GB_BN;
    hipLaunchKernelGGL((GEMM_BIG_KERNEL<EPI, F32>), dim3(nbm * nbn), dim3(512), 0, st, p);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}

// same validation as vt_gemm_bf16 (gemm_bf16.hip), which calls this for large problems
int VT_CAT(vt_gemm_big_dispatch, VT_SUFFIX)(const GemmParams& p, int epilogue, int out_fp32, hipStream_t st) {
    switch (epilogue) {
        case EPI_BIAS:
            return out_fp32 ? VT_CAT(launch_big, VT_SUFFIX)<EPI_BIAS, true>(p, st) : VT_CAT(launch_big, VT_SUFFIX)<EPI_BIAS, false>(p, st);
        case EPI_BIAS_GELU: return VT_CAT(launch_big, VT_SUFFIX)<EPI_BIAS_GELU, false>(p, st);
        case EPI_GATED_RES: return VT_CAT(launch_big, VT_SUFFIX)<EPI_GATED_RES, false>(p, st);
        case EPI_DGELU: return VT_CAT(launch_big, VT_SUFFIX)<EPI_DGELU, false>(p, st);
        default: return VT_ERR_UNSUPPORTED;
    }
}
