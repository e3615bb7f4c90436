// Causal 3x3x3 convolution over channels-last video activations as an implicit GEMM on the matrix cores, gfx950.
//   y[n,t,h,w,co] = bias[co] + sum_{dt,dh,dw in 0..2} sum_ci x[n, max(t+dt-2, 0), h+dh-1, w+dw-1, ci] * Wk[co, (dt,dh,dw), ci]
// (zero padding in h / w, the FIRST FRAME REPLICATED in front of t -- the causal padding of the CogVideoX VAE:
// `ContextParallelCausalConv3d`, videotuna/models/cogvideo_sat/vae_modules/cp_enc_dec.py:356-433 with
// `_fake_cp_pass_from_previous_rank` :228-273; diffusers' CogVideoXCausalConv3d behind cogvideo_pl.py:792-813).  Second kernel of the
// next scope row (the VAE encoder: ~150 TFLOP per 49x480x720 sample, nearly all of it in these convolutions).
//
// GEMM view: M = N*T*H*W output positions, N = Cout, K = 27 * Cin; the K loop walks the 27 taps, Cin/64 K-tiles each.  Same
// 128x128x64 tile, LDS-DMA staging, swizzle and MFMA loop as gemm_bf16.hip; the only difference is the A operand: the row a lane
// fetches for output position m and tap (dt,dh,dw) is the input position m + delta(tap) -- or nothing (an out-of-range buffer
// offset reads zeros) where the tap leaves the image.  No im2col buffer exists; each input row is fetched 27 times, from the L2
// (a tile's halo is ~0.9 MB, shared with its neighbours on the same XCD).
#include "gemm_epilogue.h"
#include <stdlib.h>

struct Conv3dParams {
    GemmParams g;          // C = y, ldc, bias, R (residual), M = output positions, N = Cout (the epilogue's view); A / W = x / packed weight
    int T, H, W, Cin;      // input frames / lines / pixels per sample, input channels
    int Ho, Wo;            // output lines / pixels (H, W at stride 1; H/2, W/2 for the stride-2 downsample)
    long long ldx;         // position stride of x in elements (>= Cin)
    long long x_rows;      // N*T*H*W
};

#define CV_BM 128
#define CV_BN 128
#define CV_BK 64
#define CV_CS_LD 132
#define CV_OOB 0x80000000u   // buffer offset that is always out of range (num_records is clamped to < 2^31): the fetch returns zeros

// KT = temporal taps (3: causal 3x3x3, 1: per-frame 3x3), STRIDE = spatial stride (1: padding 1 on all sides, 2: the VAE's
// downsample -- no padding on top / left, one zero line / column at the bottom / right), EPI = EPI_BIAS or EPI_GATED_RES (+ residual).
// IN8: the first convolution of the encoder (RGB input): x has 8 channels per position (3 used) = ONE 16-byte chunk, and a K-tile
// of 64 holds 8 TAPS x 8 channels instead of 64 channels of one tap -- 4 K-tiles (32 tap slots, 27 used) instead of 27.
template <int KT, int STRIDE, int EPI, bool IN8 = false>
__global__ __launch_bounds__(256, 2) void conv3d_cl_kernel(Conv3dParams cp) {
    constexpr int PAD = STRIDE == 1 ? 1 : 0;
    __shared__ __attribute__((aligned(16))) char smem[65536];
    const GemmParams& p = cp.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int nbm = (p.M + CV_BM - 1) / CV_BM, nbn = (p.N + CV_BN - 1) / CV_BN;
    // XCD-aware id, then consecutive row tiles (neighbouring positions, shared halos) on the same XCD, column tiles inner
    const int id = xcd_remap(blockIdx.x, nbm * nbn);
    const int tile_m = id / nbn, tile_n = id % nbn;
    const int row0 = tile_m * CV_BM, col0 = tile_n * CV_BN;
    const int HW = cp.H * cp.W, HWo = cp.Ho * cp.Wo;

    // A descriptor rooted at the first input FRAME the tile can touch (frame max(t0 - (KT-1), 0) of the first output position's
    // sample): later output positions only reach the same or later frames, so all byte offsets are non-negative and, the host
    // having bounded the span, fit 31 bits
    long long base_row;
    {
        const long long nt = (long long)row0 / HWo;
        const int t0 = (int)(nt % cp.T);
        int tt = t0 - (KT - 1); tt = tt < 0 ? 0 : tt;
        base_row = ((nt / cp.T) * cp.T + tt) * (long long)HW;
    }
    const long long a_rem = (cp.x_rows - base_row) * cp.ldx * 2;
    __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A + base_row * cp.ldx, (unsigned)(a_rem > 0x7fffffffLL ? 0x7fffffffLL : a_rem));
    const long long w_rem = (long long)(p.N - col0) * p.ldw * 2;
    __amdgpu_buffer_rsrc_t rw = make_rsrc(p.W + (size_t)col0 * p.ldw, (unsigned)(w_rem > 0x7fffffffLL ? 0x7fffffffLL : w_rem));

    // ---- LDS-DMA staging (see gemm_bf16.hip): wave w moves blocks w, w+4, w+8, w+12 (8 rows x 128 B each) of each operand ----
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int drl = lane >> 3, dcp = lane & 7;
    const int chunk_off = (dcp ^ drl) << 4;
    int w_voff[4];
    int rt[4], rh[4], rw_[4];            // (t, ho, wo) of this lane's four output rows; rt < 0: row beyond M
    long long rfr[4];                    // first input row of the sample's frame 0, relative to base_row: (n * T) * H * W - base_row
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * (wv + 4 * j) + drl;
        w_voff[j] = row * p.ldw * 2 + chunk_off;
        const long long m = (long long)row0 + row;
        if (m < p.M) {
            const int sp = (int)(m % HWo);
            const long long nt = m / HWo;
            rt[j] = (int)(nt % cp.T);
            rh[j] = sp / cp.Wo;
            rw_[j] = sp - rh[j] * cp.Wo;
            rfr[j] = (nt / cp.T) * cp.T * (long long)HW - base_row;
        } else {
            rt[j] = -1; rh[j] = 0; rw_[j] = 0; rfr[j] = 0;
        }
    }
    const int kpt = IN8 ? 1 : cp.Cin / CV_BK;      // K-tiles per tap (IN8: per group of 8 taps)
    unsigned a_voff[4];
    auto set_tap = [&](int tap) {
        if (IN8) tap = tap * 8 + (dcp ^ drl);       // this lane's chunk of the K-tile IS a tap: group `tap`, slot = logical chunk
        const int dt = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int tt = rt[j] + dt - (KT - 1);
            tt = tt < 0 ? 0 : tt;                                   // causal: frames before the first one are the first one
            const int hh = rh[j] * STRIDE + dh - PAD, ww = rw_[j] * STRIDE + dw - PAD;
            const bool ok = rt[j] >= 0 && hh >= 0 && hh < cp.H && ww >= 0 && ww < cp.W && (!IN8 || tap < KT * 9);
            const long long r = rfr[j] + (long long)tt * HW + (long long)hh * cp.W + ww;
            a_voff[j] = ok ? (unsigned)(r * cp.ldx * 2 + (IN8 ? 0 : chunk_off)) : CV_OOB;
        }
    };
    int d_tap = 0, d_kc = 0;             // (tap, K-tile inside the tap) of the next K-tile to fetch
    auto dma = [&](int buf) {
        if (d_kc == 0) set_tap(d_tap);
        const int a_soff = IN8 ? 0 : d_kc * CV_BK * 2;
        const int w_soff = IN8 ? d_tap * CV_BK * 2 : (d_tap * cp.Cin + d_kc * CV_BK) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            char* dst = smem + buf * 32768 + (wv + 4 * j) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)dst, 16, (int)a_voff[j], a_soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(dst + 16384), 16, w_voff[j], w_soff, 0, 0);
        }
        if (++d_kc == kpt) { d_kc = 0; ++d_tap; }
    };

    f32x4 acc[4][4];   // [tn][tm]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = IN8 ? (KT * 9 + 7) / 8 : KT * 9 * kpt;
    dma(0);
    __syncthreads();
    const int frow = lane & 15, fq = lane >> 4, fx = lane & 7;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) dma(buf ^ 1);
        const char* As = smem + buf * 32768;
        const char* Ws = As + 16384;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = (((ks * 4 + fq) ^ fx) << 4);
            bf16x8 af[4], wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[t] = *(const bf16x8*)(As + (wm * 64 + t * 16 + frow) * 128 + coff);
                wf[t] = *(const bf16x8*)(Ws + (wn * 64 + t * 16 + frow) * 128 + coff);
            }
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tn], af[tm], acc[tn][tm], 0, 0, 0);
        }
        __syncthreads();       // its vmcnt(0) also retires the next tile's LDS-DMA
    }

    // ---------------- epilogue: two 64-row halves through LDS, bias, bf16 store ----------------
    float* Cs = (float*)smem;
    const int er = tid >> 5, ec = (tid & 31) * 4;
    const int n = col0 + ec;
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr && n < p.N) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bias4[j] = bf2f(p.bias[n + j]);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (wm == half) {
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
                    *(f32x4*)(Cs + (tm * 16 + frow) * CV_CS_LD + wn * 64 + tn * 16 + fq * 4) = acc[tn][tm];
        }
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int ml = pass * 8 + er;
            const int m = row0 + half * 64 + ml;
            if (m < p.M && n < p.N) {
                const f32x4 v = *(const f32x4*)(Cs + ml * CV_CS_LD + ec);
                gemm_epilogue_store<EPI, false>(p, m, n, v, bias4);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Producer / consumer version (the structure of gemm_pc_bf16.hip: 256 x 128 tile, waves 0..3 multiply 128 x 64 each with
// v_mfma_f32_32x32x16_bf16, waves 4..7 only issue LDS-DMA two K-tiles ahead in a three-stage ring, persistent over the output
// tiles).  The address generation of the implicit GEMM -- decode of the output positions, per-tap validity and offsets -- runs in
// the loader waves, which have nothing else to do, so the multipliers see an ordinary GEMM.
#ifndef CP_MFMA16
#define CP_MFMA16 1        // 16x16x32 MFMA body in the producer / consumer kernel (see gemm_big_bf16.hip GB_MFMA16)
#endif
#define CP_BM 256
#define CP_STAGE 49152      // A 32 KiB | W 16 KiB
#define CP_NS 3
#define CP_PIECES 12        // LDS-DMA operations per loader wave and K-tile: 8 of A + 4 of W
static __device__ __forceinline__ void cp_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int KT, int STRIDE, int EPI>
__global__ __launch_bounds__(512, 1) void conv3d_pc_kernel(Conv3dParams cp) {
    constexpr int PAD = STRIDE == 1 ? 1 : 0;
    __shared__ __attribute__((aligned(16))) char smem[CP_NS * CP_STAGE];
    const GemmParams& p = cp.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbm = (p.M + CP_BM - 1) / CP_BM, nbn = (p.N + CV_BN - 1) / CV_BN;
    const int ntiles = nbm * nbn;
    const int spx = gridDim.x >> 3;
    const int slot = (int)(blockIdx.x & 7) * spx + (int)(blockIdx.x >> 3);      // an XCD's slots are consecutive tiles: shared halos
    const int kpt = cp.Cin / CV_BK;
    const int nk = KT * 9 * kpt;                 // K-tiles per output tile
    if (slot >= ntiles) return;
    const int my_tiles = (ntiles - slot + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total_kt = my_tiles * nk;
    const int HW = cp.H * cp.W, HWo = cp.Ho * cp.Wo;

    if (wave >= 4) {
        // =================================== loader waves ===================================
        const int lw = wave - 4;
        const int drl = lane >> 3, dcp = lane & 7;
        const int row0p = 8 * lw + drl;                       // this lane's row inside piece j: row0p + 32 j
        const int chunk_off = (dcp ^ ((row0p >> 1) & 7)) << 4; // swz(row) = (row >> 1) & 7, unchanged by the 32-row piece step
        const int w_voff0 = row0p * p.ldw * 2 + chunk_off;
        const int w_pstep = 32 * p.ldw * 2;
        int tile = slot;
        int rt[8], rh[8], rw_[8];
        long long rfr[8];
        unsigned a_voff[8];
        __amdgpu_buffer_rsrc_t ra, rw;
        auto set_tile = [&](int tl) {
            const int row0 = (tl / nbn) * CP_BM, col0 = (tl % nbn) * CV_BN;
            const long long nt0 = (long long)row0 / HWo;
            int tt = (int)(nt0 % cp.T) - (KT - 1); tt = tt < 0 ? 0 : tt;
            const long long base_row = ((nt0 / cp.T) * cp.T + tt) * (long long)HW;         // first frame the tile can touch
            const long long a_rem = (cp.x_rows - base_row) * cp.ldx * 2;
            ra = make_rsrc(p.A + base_row * cp.ldx, (unsigned)(a_rem > 0x7fffffffLL ? 0x7fffffffLL : a_rem));
            const long long w_rem = (long long)(p.N - col0) * p.ldw * 2;
            rw = make_rsrc(p.W + (size_t)col0 * p.ldw, (unsigned)(w_rem > 0x7fffffffLL ? 0x7fffffffLL : w_rem));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long long m = (long long)row0 + row0p + 32 * j;
                if (m < p.M) {
                    const int sp = (int)(m % HWo);
                    const long long nt = m / HWo;
                    rt[j] = (int)(nt % cp.T);
                    rh[j] = sp / cp.Wo;
                    rw_[j] = sp - rh[j] * cp.Wo;
                    rfr[j] = (nt / cp.T) * cp.T * (long long)HW - base_row;
                } else {
                    rt[j] = -1; rh[j] = 0; rw_[j] = 0; rfr[j] = 0;
                }
            }
        };
        auto set_tap = [&](int tap) {
            const int dt = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int tt = rt[j] + dt - (KT - 1);
                tt = tt < 0 ? 0 : tt;
                const int hh = rh[j] * STRIDE + dh - PAD, ww = rw_[j] * STRIDE + dw - PAD;
                const bool ok = rt[j] >= 0 && hh >= 0 && hh < cp.H && ww >= 0 && ww < cp.W;
                const long long r = rfr[j] + (long long)tt * HW + (long long)hh * cp.W + ww;
                a_voff[j] = ok ? (unsigned)(r * cp.ldx * 2 + chunk_off) : CV_OOB;
            }
        };
        set_tile(tile);
        int d_tap = 0, d_kc = 0;
        auto issue = [&](int g) {                // K-tile number g of the stream -> stage g % 3
            if (d_kc == 0) set_tap(d_tap);
            const int a_soff = d_kc * CV_BK * 2;
            const int w_soff = (d_tap * cp.Cin + d_kc * CV_BK) * 2;
            char* st = smem + (g % CP_NS) * CP_STAGE + lw * 1024;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(st + j * 4096), 16, (int)a_voff[j], a_soff, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(st + 32768 + j * 4096), 16, w_voff0, w_soff + j * w_pstep, 0, 0);
            if (++d_kc == kpt) {
                d_kc = 0;
                if (++d_tap == KT * 9) {         // next K-tile belongs to the next output tile
                    d_tap = 0;
                    tile += gridDim.x;
                    if (tile < ntiles) set_tile(tile);
                }
            }
        };
        int issued = 0;
        for (; issued < 2 && issued < total_kt; ++issued) issue(issued);
        for (int g = 0; g < total_kt; ++g) {
            if (issued > g + 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CP_PIECES) : "memory");     // K-tile g landed, g+1 may fly
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            cp_barrier();                        // (A) stage g % 3 is ready / stage (g-1) % 3 has been consumed
            if ((g + 1) % nk == 0) {
#pragma unroll 1
                for (int i = 0; i < 17; ++i) cp_barrier();      // mirror the multipliers' epilogue barriers
            }
            if (issued < total_kt) { issue(issued); ++issued; }
        }
        return;
    }

    // =================================== multiplier waves (an ordinary GEMM from here on) ===================================
    const int wm = wave & 1, wn = wave >> 1;
    const int fr = lane & 31, fh = lane >> 5, fx = (fr >> 1) & 7;
    const int fr16 = lane & 15, fq = lane >> 4, fx16 = (fr16 >> 1) & 7;
    const int er = tid >> 5, ec = (tid & 31) * 4;
    int g = 0;
    for (int tile = slot; tile < ntiles; tile += gridDim.x) {
        const int row0 = (tile / nbn) * CP_BM, col0 = (tile % nbn) * CV_BN;
#if CP_MFMA16
        // v_mfma_f32_16x16x32_bf16 body (gemm_pc_bf16.hip): 8 x 4 tiles of 16 x 16 per wave, two 32-deep k-steps per K-tile, reads one sub-step ahead
        f32x4 acc16[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc16[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt, ++g) {
            cp_barrier();                    // (A) the loaders have landed stage g % 3
            const char* As = smem + (g % CP_NS) * CP_STAGE + (wm * 128 + fr16) * 128;
            const char* Ws = smem + (g % CP_NS) * CP_STAGE + 32768 + (wn * 64 + fr16) * 128;
            bf16x8 af[2][8], wf[2][4];
            auto frags_w = [&](int ks) {
                const int coff = (((ks * 4 + fq) ^ fx16) << 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) wf[ks][t] = *(const bf16x8*)(Ws + t * 2048 + coff);
            };
            auto frags_a = [&](int ks, int h) {
                const int coff = (((ks * 4 + fq) ^ fx16) << 4);
#pragma unroll
                for (int t = h * 4; t < (h + 1) * 4; ++t) af[ks][t] = *(const bf16x8*)(As + t * 2048 + coff);
            };
            auto mm = [&](int ks, int h) {
#pragma unroll
                for (int tm = h * 4; tm < (h + 1) * 4; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn)
                        acc16[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][tn], af[ks][tm], acc16[tn][tm], 0, 0, 0);
            };
            frags_w(0); frags_a(0, 0);
            frags_a(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            frags_w(1); frags_a(1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mm(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            frags_a(1, 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mm(1, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
#else
        f32x16 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int kt = 0; kt < nk; ++kt, ++g) {
            cp_barrier();                    // (A) the loaders have landed stage g % 3
            const char* As = smem + (g % CP_NS) * CP_STAGE + (wm * 128 + fr) * 128;
            const char* Ws = smem + (g % CP_NS) * CP_STAGE + 32768 + (wn * 64 + fr) * 128;
            bf16x8 af[2][4], wf[2][2];
            auto frags = [&](int ks) {
                const int coff = (((ks * 2 + fh) ^ fx) << 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) af[ks & 1][t] = *(const bf16x8*)(As + t * 4096 + coff);
#pragma unroll
                for (int t = 0; t < 2; ++t) wf[ks & 1][t] = *(const bf16x8*)(Ws + t * 4096 + coff);
            };
            frags(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) frags(ks + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
                        acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks & 1][tn], af[ks & 1][tm], acc[tn][tm], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#endif
        // ---------------- epilogue: eight 32-row slabs through the stage that was just consumed (17 barriers) ----------------
        float* Cs = (float*)(smem + ((g - 1) % CP_NS) * CP_STAGE);
        const int n = col0 + ec;
        float bias4[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr && n < p.N) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bias4[j] = bf2f(p.bias[n + j]);
        }
        cp_barrier();
#pragma unroll
        for (int slab = 0; slab < 8; ++slab) {
            if (wm == (slab >> 2)) {
#if CP_MFMA16
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn)
                        *(f32x4*)(Cs + (t2 * 16 + fr16) * CV_CS_LD + wn * 64 + tn * 16 + 4 * fq) = acc16[tn][(slab & 3) * 2 + t2];
#else
                const int tm = slab & 3;
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *(f32x4*)(Cs + fr * CV_CS_LD + wn * 64 + tn * 32 + 8 * q + 4 * fh) =
                            (f32x4){acc[tn][tm][4 * q], acc[tn][tm][4 * q + 1], acc[tn][tm][4 * q + 2], acc[tn][tm][4 * q + 3]};
#endif
            }
            cp_barrier();
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int ml = pass * 8 + er;
                const int m = row0 + slab * 32 + ml;
                if (m < p.M && n < p.N) {
                    const f32x4 v = *(const f32x4*)(Cs + ml * CV_CS_LD + ec);
                    gemm_epilogue_store<EPI, false>(p, m, n, v, bias4);
                }
            }
            cp_barrier();
        }
    }
}

template <int KT, int STRIDE>
static int conv_launch(const void* x, long long ldx, const void* wk, const void* bias, const void* res, long long ldr, void* y, long long ldy,
                       int N, int T, int H, int W, int Cin, int Cout, hipStream_t st) {
    if (N <= 0 || T <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (Cin % CV_BK) || (Cout % 4)) return VT_ERR_BAD_SHAPE;
    if (STRIDE == 2 && ((H % 2) || (W % 2))) return VT_ERR_BAD_SHAPE;
    if ((ldx % 8) || (ldy % 4) || ldx < Cin || ldy < Cout || (res != nullptr && ((ldr % 4) || ldr < Cout))) return VT_ERR_BAD_SHAPE;
    if ((((uintptr_t)x) | ((uintptr_t)wk) | ((uintptr_t)y) | ((uintptr_t)res)) & 15) return VT_ERR_BAD_ALIGN;
    const int Ho = H / STRIDE, Wo = W / STRIDE;
    const long long rows_in = (long long)N * T * H * W, rows_out = (long long)N * T * Ho * Wo;
    // 31-bit byte offsets from the start of the first frame a tile touches: at most KT + 2 frames (a tile may end one sample and
    // begin the next) plus the lines it covers
    const long long span = ((long long)(KT + 2) * H * W + ((long long)CV_BM / Wo + 4) * STRIDE * W + 256) * ldx * 2;
    if (rows_out >= 0x7fffffffLL || span >= 0x7fffffffLL || (long long)KT * 9 * Cin * 2 * CV_BN >= 0x7fffffffLL) return VT_ERR_BAD_SHAPE;
    Conv3dParams cp;
    GemmParams& p = cp.g;
    p.A = (const bf16_t*)x; p.W = (const bf16_t*)wk; p.C = y; p.bias = (const bf16_t*)bias;
    p.R = (const bf16_t*)res; p.gate_txt = nullptr; p.gate_vid = nullptr; p.C2 = nullptr; p.U = nullptr;
    p.M = (int)rows_out; p.N = Cout; p.K = KT * 9 * Cin; p.lda = (int)ldx; p.ldw = KT * 9 * Cin; p.ldc = (int)ldy; p.ldr = (int)ldr;
    p.ldc2 = 0; p.ldu = 0; p.S = 1; p.St = 0; p.gate_bstride = 0; p.r_mod = 0; p.splits = 1;
    cp.T = T; cp.H = H; cp.W = W; cp.Cin = Cin; cp.Ho = Ho; cp.Wo = Wo; cp.ldx = ldx; cp.x_rows = rows_in;
    // kernel: the producer / consumer one (256-row tiles, persistent) where it was measured faster; VT_CONV_KERNEL=1|2 forces the
    // 128-row / the producer-consumer kernel
    static int kmode = -1, slots = 0;
    if (kmode < 0) {
        const char* e = getenv("VT_CONV_KERNEL");
        kmode = e ? atoi(e) : 0;
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        slots = cus >= 8 ? cus / 8 * 8 : 8;
    }
    const int nbn = (Cout + CV_BN - 1) / CV_BN;
    const long long tiles_pc = ((rows_out + CP_BM - 1) / CP_BM) * nbn;
    const long long span_pc = ((long long)(KT + 2) * H * W + ((long long)CP_BM / Wo + 4) * STRIDE * W + 256) * ldx * 2;
    // measured (tools/kbench.py conv, VT_CONV_KERNEL=1 / 2): 128 -> 128 at 480x720 822 / 804 TFLOP/s, 256 -> 256 at 240x360 1024 / 1087,
    // at 120x180 1055 / 1048, 512 -> 512 at 60x90 1117 / 1065: the 256-row kernel pays only with two column tiles and many rounds
    const bool pc = span_pc < 0x7fffffffLL && (kmode == 2 || (kmode == 0 && nbn >= 2 && tiles_pc >= 16LL * slots));
    if (pc) {
        const int grid = tiles_pc < slots ? (int)((tiles_pc + 7) / 8 * 8) : slots;
        if (res != nullptr) hipLaunchKernelGGL((conv3d_pc_kernel<KT, STRIDE, EPI_GATED_RES>), dim3(grid), dim3(512), 0, st, cp);
        else hipLaunchKernelGGL((conv3d_pc_kernel<KT, STRIDE, EPI_BIAS>), dim3(grid), dim3(512), 0, st, cp);
        return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
    }
    const int nbm = (p.M + CV_BM - 1) / CV_BM;
    if (res != nullptr) hipLaunchKernelGGL((conv3d_cl_kernel<KT, STRIDE, EPI_GATED_RES>), dim3(nbm * nbn), dim3(256), 0, st, cp);
    else hipLaunchKernelGGL((conv3d_cl_kernel<KT, STRIDE, EPI_BIAS>), dim3(nbm * nbn), dim3(256), 0, st, cp);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}

// x: bf16 [N, T, H, W, Cin] channels-last (position stride ldx >= Cin, a multiple of 8); wk: bf16 [Cout, 27 * Cin], tap-major
// (dt, dh, dw, ci) -- i.e. torch's Conv3d weight [Cout, Cin, 3, 3, 3] permuted to [Cout, 3, 3, 3, Cin]; bias bf16 [Cout] or null;
// res: optional bf16 [N, T, H, W, Cout] (position stride ldr) added to the result (the ResNet block's skip); y: bf16
// [N, T, H, W, Cout] (position stride ldy).  Cin % 64 == 0, Cout % 4 == 0.
extern "C" int vt_causal_conv3d_cl(const void* x, long long ldx, const void* wk, const void* bias, const void* res, long long ldr,
                                   void* y, long long ldy, int N, int T, int H, int W, int Cin, int Cout, void* stream) {
    return conv_launch<3, 1>(x, ldx, wk, bias, res, ldr, y, ldy, N, T, H, W, Cin, Cout, (hipStream_t)stream);
}

// The VAE's spatial downsample: per frame, one zero line / column appended at the bottom / right, 3x3 convolution with stride 2 and
// no other padding (DownSample3D.forward, cp_enc_dec.py:660-666).  wk: bf16 [Cout, 9 * Cin] = Conv2d weight permuted to
// [Cout, 3, 3, Cin]; y: bf16 [N, T, H/2, W/2, Cout].  H, W even.
extern "C" int vt_downsample_conv2d_cl(const void* x, long long ldx, const void* wk, const void* bias, void* y, long long ldy,
                                       int N, int T, int H, int W, int Cin, int Cout, void* stream) {
    return conv_launch<1, 2>(x, ldx, wk, bias, nullptr, 0, y, ldy, N, T, H, W, Cin, Cout, (hipStream_t)stream);
}

// The encoder's first convolution (RGB input): x bf16 [N,T,H,W,8] (8 channels per position, the unused ones zero), wk bf16
// [Cout, 32 * 8] = the Conv3d weight [Cout, Cin<=8, 3,3,3] as (tap, channel) with taps 27..31 and channels >= Cin zero; y bf16
// [N,T,H,W,Cout].  Same arithmetic as vt_causal_conv3d_cl with 8 taps per K-tile (4 K-tiles instead of 27).
extern "C" int vt_causal_conv3d_in8_cl(const void* x, const void* wk, const void* bias, void* y, long long ldy,
                                       int N, int T, int H, int W, int Cout, void* stream) {
    if (N <= 0 || T <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (Cout % 4) || (ldy % 4) || ldy < Cout) return VT_ERR_BAD_SHAPE;
    if ((((uintptr_t)x) | ((uintptr_t)wk) | ((uintptr_t)y)) & 15) return VT_ERR_BAD_ALIGN;
    const long long rows = (long long)N * T * H * W;
    const long long span = (5LL * H * W + ((long long)CV_BM / W + 4) * W + 256) * 16;
    if (rows >= 0x7fffffffLL || span >= 0x7fffffffLL) return VT_ERR_BAD_SHAPE;
    Conv3dParams cp;
    GemmParams& p = cp.g;
    p.A = (const bf16_t*)x; p.W = (const bf16_t*)wk; p.C = y; p.bias = (const bf16_t*)bias;
    p.R = nullptr; p.gate_txt = nullptr; p.gate_vid = nullptr; p.C2 = nullptr; p.U = nullptr;
    p.M = (int)rows; p.N = Cout; p.K = 256; p.lda = 8; p.ldw = 256; p.ldc = (int)ldy; p.ldr = 0; p.ldc2 = 0; p.ldu = 0;
    p.S = 1; p.St = 0; p.gate_bstride = 0; p.r_mod = 0; p.splits = 1;
    cp.T = T; cp.H = H; cp.W = W; cp.Cin = 8; cp.Ho = H; cp.Wo = W; cp.ldx = 8; cp.x_rows = rows;
    const int nbm = (p.M + CV_BM - 1) / CV_BM, nbn = (Cout + CV_BN - 1) / CV_BN;
    hipLaunchKernelGGL((conv3d_cl_kernel<3, 1, EPI_BIAS, true>), dim3(nbm * nbn), dim3(256), 0, (hipStream_t)stream, cp);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}
