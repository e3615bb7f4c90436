// bf16 MFMA GEMM for gfx950, producer / consumer version:  C[M,N] = A[M,K] * W[N,K]^T (+ fused epilogue), fp32 accumulate.
//
// Same contract and epilogues as gemm_bf16.hip / gemm_big_bf16.hip (the nn.Linear layers of diffusers' CogVideoXBlock reached
// through videotuna/models/cogvideo_hf/cogvideo_pl.py:865-871, SURVEY 8(a) a4,a5).
//
// Why: the ablations of the 256x256 kernel (gemm_big_bf16.hip) show its MFMA / LDS-read loop running at 1.4-1.8 PFLOP/s once
// the operand staging is taken away and at 1.0-1.25 with it -- an LDS-DMA piece (1 KiB) blocks the wave that issues it for
// 100-185 cycles inside an MFMA phase, and every wave issues 8 of them per K-tile.  Here the two jobs are separate waves:
//   * waves 0..3 (one per SIMD) only multiply: 256x128 workgroup tile, 128x64 per wave = 4x2 tiles of
//     v_mfma_f32_32x32x16_bf16 (128 accumulator registers), fragments of the next k-step fetched while the current one is
//     multiplied;
//   * waves 4..7 (one per SIMD) only move data: 12 LDS-DMA pieces per K-tile each (8 of A, 4 of W), always two K-tiles
//     ahead of the multipliers in a THREE-stage ring (3 x 48 KiB), waiting with a counted vmcnt that leaves the newest
//     K-tile in flight.
// One barrier per K-tile hands a landed stage to the multipliers and a consumed one back to the loaders.  The kernel is
// persistent (one workgroup per CU) and the ring runs straight across output tiles: while the multipliers run the epilogue
// of tile T, the first two K-tiles of tile T+1 are already landing.
#include "gemm_epilogue.h"

#ifndef VT_SUFFIX
#define VT_SUFFIX
#endif
#define VT_CAT_(a, b) a##b
#define VT_CAT(a, b) VT_CAT_(a, b)
#define GEMM_PC_KERNEL VT_CAT(gemm_tn_pc_kernel, VT_SUFFIX)

#define GP_BM 256           // rows of the standard tile; the BM = 128 instantiation (four-stage ring) serves few-row GEMMs
#define GP_BN 128
#define GP_BK 64
#ifndef GP_MFMA16
#define GP_MFMA16 1         // 1: v_mfma_f32_16x16x32_bf16 (16-row fragments, two 32-deep k-steps per K-tile) instead of v_mfma_f32_32x32x16_bf16:
#endif                      // same LDS image and bytes per flop; the chip holds a higher clock on this shape under load (gemm_big_bf16.hip)
#ifndef GP_SUB
#define GP_SUB 1           // 1: fragment reads issued one sub-step (k-step x half of the A tiles) ahead: +1 % on the 72 008-row shapes; 0: one k-step ahead
#endif
#define GP_CS_LD 132        // fp32 row stride of the epilogue staging slab (32 rows x 132 floats = 16.5 KiB)
template <int BM> struct GpCfg {
    static constexpr int STAGE = BM * 128 + 16384;          // A (BM rows x 128 B) | W 16 KiB
    static constexpr int NS = BM == 256 ? 3 : 4;            // ring stages: 144 KiB / 128 KiB of LDS
    static constexpr int AHEAD = NS - 1;                    // K-tiles the loaders run ahead of the multipliers
    static constexpr int APIECES = BM / 32;                 // 1-KiB pieces of A per loader wave per K-tile
    static constexpr int PIECES = APIECES + 4;              // ... plus 4 of W
    static constexpr int SLABS = BM / 32;                   // 32-row epilogue slabs
    static constexpr int TM = BM / 64;                      // 32-row A fragments per multiplier wave
};

static __device__ __forceinline__ void gp_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct GpTile {
    int row0, col0;
    const bf16_t* a;
    const bf16_t* w;
    unsigned a_bytes, w_bytes;
};

template <int BM>
static __device__ __forceinline__ GpTile gp_tile(const GemmParams& p, int id, int nbm, int nbn) {
    // grouped ordering (4 row-tiles per group, column-tiles outer): the 32 workgroups of an XCD that run together share
    // A row-panels / W column-panels in its L2
    const int GM = 4;
    const int in_group = GM * nbn;
    const int group = id / in_group;
    const int first_m = group * GM;
    const int gsz = min(nbm - first_m, GM);
    GpTile t;
    t.row0 = (first_m + (id % in_group) % gsz) * BM;
    t.col0 = ((id % in_group) / gsz) * GP_BN;
    const long long a_rem = (long long)(p.M - t.row0) * p.lda * 2;
    const long long w_rem = (long long)(p.N - t.col0) * p.ldw * 2;
    t.a = p.A + (size_t)t.row0 * p.lda;
    t.w = p.W + (size_t)t.col0 * p.ldw;
    t.a_bytes = (unsigned)(a_rem > 0x7fffffffLL ? 0x7fffffffLL : a_rem);
    t.w_bytes = (unsigned)(w_rem > 0x7fffffffLL ? 0x7fffffffLL : w_rem);
    return t;
}

template <int EPI, bool OUT_F32, int BM>
__global__ __launch_bounds__(512, 1) void GEMM_PC_KERNEL(GemmParams p) {
    using Cfg = GpCfg<BM>;
    constexpr int GP_STAGE = Cfg::STAGE, GP_NS = Cfg::NS;
    __shared__ __attribute__((aligned(16))) char smem[GP_NS * GP_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + GP_BN - 1) / GP_BN;
    // split-K (p.splits > 1, fp32 C, EPI_BIAS only; the host guarantees (K / 64) % splits == 0): a work item is (output tile,
    // K range); item w = tile w % ntiles_mn of split w / ntiles_mn, so the workgroups that run together share one K range of A
    const int ntiles_mn = nbm * nbn;
    const int splits = p.splits > 1 ? p.splits : 1;
    const int ntiles = ntiles_mn * splits;       // work items
    const int spx = gridDim.x >> 3;
    const int slot = (int)(blockIdx.x & 7) * spx + (int)(blockIdx.x >> 3);
    const int nk = p.K / GP_BK / splits;         // K-tiles per work item
    if (slot >= ntiles) return;
    const int my_tiles = (ntiles - slot + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total_kt = my_tiles * nk;          // K-tiles this workgroup streams, over all its output tiles

    if (loader) {
        // =================================== loader waves ===================================
        // A K-tile of A is 32 pieces of 1 KiB (8 rows x 128 B), of W 16; wave lw moves pieces lw, lw+4, ...  Lane l lands at
        // (row l>>3, physical chunk l&7) and fetches logical chunk (l&7) ^ swz(row), swz(row) = (row>>1)&7 (conflict-free
        // 32-row fragment reads, see gemm_big_bf16.hip); the piece step (32 rows) does not change swz, so it is a scalar offset.
        const int lw = wave - 4;
        const int drl = lane >> 3, dcp = lane & 7;
        const int row0p = 8 * lw + drl;
        const int sw = (row0p >> 1) & 7;
        const int a_voff0 = row0p * p.lda * 2 + ((dcp ^ sw) << 4);
        const int w_voff0 = row0p * p.ldw * 2 + ((dcp ^ sw) << 4);
        const int a_pstep = 32 * p.lda * 2, w_pstep = 32 * p.ldw * 2;
        GpTile cur = gp_tile<BM>(p, slot % ntiles_mn, nbm, nbn);
        int tile = slot, kt = 0;
        int kbase = (slot / ntiles_mn) * nk;
        auto issue = [&](int g) {                // K-tile number g of the stream -> stage g % 3
            __amdgpu_buffer_rsrc_t ra = make_rsrc(cur.a, cur.a_bytes), rw = make_rsrc(cur.w, cur.w_bytes);
            const int soff = (kbase + kt) * GP_BK * 2;
            char* st = smem + (g % GP_NS) * GP_STAGE + lw * 1024;
#pragma unroll
            for (int j = 0; j < Cfg::APIECES; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(st + j * 4096), 16, a_voff0, soff + j * a_pstep, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(st + BM * 128 + j * 4096), 16, w_voff0, soff + j * w_pstep, 0, 0);
            if (++kt == nk) {                    // next K-tile belongs to the next output tile
                kt = 0;
                tile += gridDim.x;
                if (tile < ntiles) { cur = gp_tile<BM>(p, tile % ntiles_mn, nbm, nbn); kbase = (tile / ntiles_mn) * nk; }
            }
        };
        int issued = 0;
        for (; issued < Cfg::AHEAD && issued < total_kt; ++issued) issue(issued);
        for (int g = 0; g < total_kt; ++g) {
            // K-tile g must have landed before the barrier that hands it to the multipliers; the younger ones may stay in flight
            // (vmcnt retires in order: PIECES operations per K-tile and wave)
            const int younger = issued - g - 1;
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * Cfg::PIECES) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Cfg::PIECES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            gp_barrier();                        // (A) stage g % NS is ready / stage (g-1) % NS has been consumed
            const bool tile_end = (g + 1) % nk == 0;
            if (tile_end) {
                // the multipliers stage their accumulators through stage g % NS, the stages in flight are not touched; mirror their barriers
#pragma unroll 1
                for (int i = 0; i < 1 + 2 * Cfg::SLABS; ++i) gp_barrier();
            }
            if (issued < total_kt) { issue(issued); ++issued; }      // into stage (g+AHEAD) % NS = (g-1) % NS: consumed before barrier (A)
        }
        return;
    }

    // =================================== multiplier waves ===================================
    constexpr int TM = Cfg::TM;
    const int wm = wave & 1, wn = wave >> 1;                  // 2 x 2 waves: rows (BM/2) wm .., columns 64 wn ..
    const int fr = lane & 31;             // row inside a 32-row fragment
    const int fh = lane >> 5;             // which 8-element half of a 16-deep k-step
    const int fx = (fr >> 1) & 7;         // swz(row)
    const int fr16 = lane & 15, fq = lane >> 4, fx16 = (fr16 >> 1) & 7;   // 16-row fragments (GP_MFMA16)
    const int er = tid >> 5;              // epilogue (256 threads): 0..7, row inside an 8-row pass
    const int ec = (tid & 31) * 4;        // first of this thread's 4 columns
    int g = 0;
    for (int tile = slot; tile < ntiles; tile += gridDim.x) {
        const GpTile cur = gp_tile<BM>(p, tile % ntiles_mn, nbm, nbn);
        // acc[tn][tm] = D[n][m] of (W-fragment, A-fragment): lane holds m = lane & 31 and, in register r, column
        // n = 8 (r>>2) + 4 (lane>>5) + (r&3): four consecutive output columns per register quad
#if GP_MFMA16
        // acc16[tn][tm] = D[n][m] of a 16 x 16 tile: lane holds m = lane & 15 and the four consecutive columns n = 4 (lane >> 4) + r
        constexpr int TM16 = 2 * TM;
        f32x4 acc16[4][TM16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TM16; ++j) acc16[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int kt = 0; kt < nk; ++kt, ++g) {
            gp_barrier();                    // (A) the loaders have landed stage g % 3
            const char* As = smem + (g % GP_NS) * GP_STAGE + (wm * (BM / 2) + fr16) * 128;
            const char* Ws = smem + (g % GP_NS) * GP_STAGE + BM * 128 + (wn * 64 + fr16) * 128;
            bf16x8 af[2][TM16], wf[2][4];    // fragments of k-step ks in [ks]: the second set loads under the first set's MFMAs
#if GP_SUB
            // four sub-steps (k-step, half of the A tiles): the reads of a sub-step are issued one sub-step ahead, so only the first 4 + TM16/2
            // reads of a K-tile are exposed behind the barrier
            constexpr int HT = TM16 / 2;
            auto frags_w = [&](int ks) {
                const int coff = (((ks * 4 + fq) ^ fx16) << 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) wf[ks][t] = *(const bf16x8*)(Ws + t * 2048 + coff);
            };
            auto frags_a = [&](int ks, int h) {
                const int coff = (((ks * 4 + fq) ^ fx16) << 4);
#pragma unroll
                for (int t = h * HT; t < (h + 1) * HT; ++t) af[ks][t] = *(const bf16x8*)(As + t * 2048 + coff);
            };
            auto mm = [&](int ks, int h) {
#pragma unroll
                for (int tm = h * HT; tm < (h + 1) * HT; ++tm)
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
            auto frags = [&](int ks) {
                const int coff = (((ks * 4 + fq) ^ fx16) << 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) wf[ks][t] = *(const bf16x8*)(Ws + t * 2048 + coff);
#pragma unroll
                for (int t = 0; t < TM16; ++t) af[ks][t] = *(const bf16x8*)(As + t * 2048 + coff);
            };
            frags(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks == 0) frags(1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tm = 0; tm < TM16; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn)
                        acc16[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][tn], af[ks][tm], acc16[tn][tm], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#endif
#else
        f32x16 acc[2][TM];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        for (int kt = 0; kt < nk; ++kt, ++g) {
            gp_barrier();                    // (A) the loaders have landed stage g % 3
            const char* As = smem + (g % GP_NS) * GP_STAGE + (wm * (BM / 2) + fr) * 128;
            const char* Ws = smem + (g % GP_NS) * GP_STAGE + BM * 128 + (wn * 64 + fr) * 128;
            bf16x8 af[2][TM], wf[2][2];      // fragments of k-step ks in [ks & 1]: the next ones load under the current MFMAs
            auto frags = [&](int ks) {
                const int coff = (((ks * 2 + fh) ^ fx) << 4);
#pragma unroll
                for (int t = 0; t < TM; ++t) af[ks & 1][t] = *(const bf16x8*)(As + t * 4096 + coff);
#pragma unroll
                for (int t = 0; t < 2; ++t) wf[ks & 1][t] = *(const bf16x8*)(Ws + t * 4096 + coff);
            };
            frags(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) frags(ks + 1);
                // keep the six LDS reads of the next k-step ahead of this k-step's eight MFMAs (left alone, the scheduler sinks
                // them to just before their use and a lone wave then waits out the LDS latency twice per k-step)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
                        acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks & 1][tn], af[ks & 1][tm], acc[tn][tm], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#endif
        // ---------------- epilogue: eight 32-row slabs through the stage that was just consumed (17 barriers) ----------------
        float* Cs = (float*)(smem + ((g - 1) % GP_NS) * GP_STAGE);
        const int n = cur.col0 + ec;
        float bias4[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr && n < p.N && tile < ntiles_mn) {      // split-K: only the first K range adds the bias
#pragma unroll
            for (int j = 0; j < 4; ++j) bias4[j] = bf2f(p.bias[n + j]);
        }
        // second operand of the epilogue (residual / saved pre-activation): fetched one slab ahead of its use
        constexpr bool HAS_AUX = EPI == EPI_GATED_RES || EPI == EPI_DGELU;
        u32x2 aux[4], auxn[4];
        auto aux_fetch = [&](int slab, u32x2* dst) {
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int m = cur.row0 + slab * 32 + pass * 8 + er;
                dst[pass] = (u32x2){0u, 0u};
                if (HAS_AUX && m < p.M && n < p.N) dst[pass] = gemm_epilogue_aux_load<EPI>(p, m, n);
            }
        };
        aux_fetch(0, aux);
        GateCtx gctx;
        if (EPI == EPI_GATED_RES) gctx = gate_ctx_load(p, cur.row0, BM, n);
        gp_barrier();                        // every multiplier is done reading the stage (the loaders mirror this one too)
#pragma unroll
        for (int slab = 0; slab < Cfg::SLABS; ++slab) {
            if (wm == slab / TM) {
#if GP_MFMA16
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn)
                        *(f32x4*)(Cs + (t2 * 16 + fr16) * GP_CS_LD + wn * 64 + tn * 16 + 4 * fq) = acc16[tn][(slab % TM) * 2 + t2];
#else
                const int tm = slab % TM;
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int nl = wn * 64 + tn * 32 + 8 * q + 4 * fh;
                        *(f32x4*)(Cs + fr * GP_CS_LD + nl) =
                            (f32x4){acc[tn][tm][4 * q], acc[tn][tm][4 * q + 1], acc[tn][tm][4 * q + 2], acc[tn][tm][4 * q + 3]};
                    }
#endif
            }
            gp_barrier();
            if (slab < Cfg::SLABS - 1) aux_fetch(slab + 1, auxn);
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int ml = pass * 8 + er;
                const int m = cur.row0 + slab * 32 + ml;
                if (m < p.M && n < p.N) {
                    const f32x4 v = *(const f32x4*)(Cs + ml * GP_CS_LD + ec);
                    gemm_epilogue_store_aux<EPI, OUT_F32>(p, m, n, v, bias4, aux[pass], EPI == EPI_GATED_RES ? &gctx : nullptr);
                }
            }
            gp_barrier();
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) aux[pass] = auxn[pass];
        }
    }
}

// BM = 128 (four-stage ring, three K-tiles in flight) for GEMMs with few rows: twice the workgroups on the narrow outputs
static bool VT_CAT(pc_small_m, VT_SUFFIX)(const GemmParams& p) { return p.M <= 1024; }

template <int EPI, bool F32>
static int VT_CAT(launch_pc, VT_SUFFIX)(const GemmParams& p, hipStream_t st) {
    const bool small = VT_CAT(pc_small_m, VT_SUFFIX)(p);
    const int bm = small ? 128 : GP_BM;
    const int nbm = (p.M + bm - 1) / bm, nbn = (p.N + GP_BN - 1) / GP_BN;
    static int slots = 0;                 // persistent grid: one workgroup per CU, a multiple of 8
    if (slots == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        slots = cus >= 8 ? cus / 8 * 8 : 8;
    }
    const int ntiles = nbm * nbn * (p.splits > 1 ? p.splits : 1);
    const int grid = ntiles < slots ? (ntiles + 7) / 8 * 8 : slots;
    if (small) hipLaunchKernelGGL((GEMM_PC_KERNEL<EPI, F32, 128>), dim3(grid), dim3(512), 0, st, p);
    else hipLaunchKernelGGL((GEMM_PC_KERNEL<EPI, F32, GP_BM>), dim3(grid), dim3(512), 0, st, p);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}

// same validation as vt_gemm_bf16 (gemm_bf16.hip), which calls this when the producer / consumer tiling is selected
int VT_CAT(vt_gemm_pc_dispatch, VT_SUFFIX)(const GemmParams& p, int epilogue, int out_fp32, hipStream_t st) {
    switch (epilogue) {
        case EPI_BIAS:
            return out_fp32 ? VT_CAT(launch_pc, VT_SUFFIX)<EPI_BIAS, true>(p, st) : VT_CAT(launch_pc, VT_SUFFIX)<EPI_BIAS, false>(p, st);
        case EPI_BIAS_GELU: return VT_CAT(launch_pc, VT_SUFFIX)<EPI_BIAS_GELU, false>(p, st);
        case EPI_GATED_RES: return VT_CAT(launch_pc, VT_SUFFIX)<EPI_GATED_RES, false>(p, st);
        case EPI_DGELU: return VT_CAT(launch_pc, VT_SUFFIX)<EPI_DGELU, false>(p, st);
        default: return VT_ERR_UNSUPPORTED;
    }
}

// Split-K GEMM for a few hundred rows against a large weight (M <= ~1024: the frozen T5 encoder, 2 x 226 tokens): the output has
// too few 256 x 128 tiles to occupy the chip and the weights stream from HBM, so the K range is cut into `splits` pieces that
// run as separate work items of the persistent producer / consumer kernel and add their partial tiles into the fp32 C with
// atomics.  C (fp32 [M, N], row stride ldc) is zeroed here; vt_residual_cast_bf16 turns it into bf16 (+ residual).
// splits must divide K / 64; splits <= 0 picks the largest divisor <= 8 that keeps >= 8 K-tiles per piece and does not
// exceed one work item per CU.
extern "C" int VT_CAT(vt_gemm_splitk_f32, VT_SUFFIX)(const void* A, int lda, const void* W, int ldw, float* C, int ldc,
                                                      int M, int N, int K, int splits, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (K % GP_BK) != 0 || (N % 4) != 0) return VT_ERR_BAD_SHAPE;
    if ((lda % 8) || (ldw % 8) || (ldc % 4) || lda < K || ldw < K || ldc < N) return VT_ERR_BAD_SHAPE;
    if ((((uintptr_t)A) | ((uintptr_t)W) | ((uintptr_t)C)) & 15) return VT_ERR_BAD_ALIGN;
    const int nk = K / GP_BK;
    if (splits <= 0) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        const int tiles = ((M + GP_BM - 1) / GP_BM) * ((N + GP_BN - 1) / GP_BN);
        splits = 1;
        for (int s = 2; s <= 8; ++s)
            if (nk % s == 0 && nk / s >= 8 && tiles * s <= cus) splits = s;
    }
    if (splits > nk || (nk % splits) != 0) return VT_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    GemmParams p;
    p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.C = C; p.bias = nullptr; p.R = nullptr; p.gate_txt = nullptr; p.gate_vid = nullptr;
    p.C2 = nullptr; p.U = nullptr;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = 0; p.ldc2 = 0; p.ldu = 0;
    p.S = 1; p.St = 0; p.gate_bstride = 0; p.r_mod = 0; p.splits = splits;
    if (splits > 1 && hipMemset2DAsync(C, (size_t)ldc * 4, 0, (size_t)N * 4, (size_t)M, st) != hipSuccess) return VT_ERR_LAUNCH;
    return VT_CAT(launch_pc, VT_SUFFIX)<EPI_BIAS, true>(p, st);
}
