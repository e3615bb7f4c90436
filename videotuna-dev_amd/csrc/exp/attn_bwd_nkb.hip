// Flash-attention backward (dQ, dK, dV), head_dim 64, bf16 in / fp32 accumulate, non-causal, gfx950.
//
// Autograd of the F.scaled_dot_product_attention call that the reference reaches through
// videotuna/models/cogvideo_hf/cogvideo_pl.py:865-871 (loss.backward() under PL).
// P is recomputed from Q, K and the forward's log2-domain LSE; the S x S matrices never touch HBM.
//
// Structure: one workgroup = 4 waves (one per SIMD, up to 512 registers each) = 128*NKB keys of one (batch, head);
// wave w owns 32*NKB keys (NKB sub-tiles of 32) and keeps dK^T / dV^T for them in 64*NKB accumulator registers while
// the workgroup sweeps all queries in steps of 64 rows.
//   * S = Q K^T and dP = dO V^T are computed with the KEY on the MFMA lane, so the fp32 tiles P and dS are, after bf16
//     packing, directly the B operands of dV^T += dO^T P and dK^T += Q^T dS (A operands = transposed LDS reads of the
//     dO / Q tiles, ds_read_b64_tr_b16).  V fragments stay in registers, K fragments are row reads of the K image.
//   * the row constants -lse2 and -delta (delta = rowsum(dO*O)) are the INITIAL accumulators of S and dP, so
//     P = exp2(S'') and dS = P * dP' need no subtraction; keys / queries past the end are masked by -1e30 constants.
//   * with one wave per SIMD nothing else hides MFMA latency: the VALU work of tile j is interleaved in program order
//     with the dV/dK MFMAs of tile j-1 and the S/dP MFMAs of tile j+1 (software pipeline over the 2*NKB tiles of a step).
//   * only dS crosses LDS: every wave writes its part of a [keys][64 q] image; after a barrier each wave computes one
//     32x32 tile of dQ over ALL keys of the block and adds it to a fp32 dQ buffer with buffer_atomic_add_f32 (one
//     register of a 32x32 accumulator = two full 128-B row segments, the full-rate atomic shape; rows past the end are
//     dropped by the buffer bounds check).  The atomics are the structural floor of this design (memory-side rate
//     ~1.3 TB/s): bytes = |dQ| * (S / keys-per-block), hence the large key block (NKB = 3 -> 384 keys).
//   * Q / dO tiles of the next step arrive by LDS-DMA (global_load_lds, swizzle applied on the source address) while
//     the dQ phase runs; the 16 atomics of a step are issued after the closing barrier so no wait ever drains them.
// All LDS images (K block, dS, Q tile, dO tile) have 128-B rows and share one XOR swizzle that is conflict-free for
// both the row reads (ds_read_b128) and the transposed reads.
#include "../common.h"

// Experiment hooks: the same source can be compiled again under another symbol suffix / variant flags
// (tools/build_variants.sh) so kernel variants are A/B-timed in ONE process.  The shipped build defines none of these.
#ifndef VT_SUFFIX
#define VT_SUFFIX
#endif
#ifndef VT_NKB
#define VT_NKB 3         // 32-key sub-tiles per wave: 2 -> 256-key blocks, 3 -> 384, 4 -> 512
#endif
#ifndef VT_ABL
#define VT_ABL 0         // timing-only ablations (results are WRONG): 1 = no dQ phase, 2 = no dQ atomics
#endif
#define VT_CAT_(a, b) a##b
#define VT_CAT(a, b) VT_CAT_(a, b)
#define BWD_KERNEL VT_CAT(attn_bwd_hd64_kernel, VT_SUFFIX)
#define BWD_BODY VT_CAT(attn_bwd_body, VT_SUFFIX)
#define DELTA_KERNEL VT_CAT(attn_bwd_delta_kernel, VT_SUFFIX)
#define BWD_ENTRY VT_CAT(vt_attn_bwd_hd64, VT_SUFFIX)

struct AttnBwdParams {
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* v;
    const bf16_t* dout;
    const float* lse2;    // [B,H,S]
    const float* delta;   // [B,H,S]
    float* dq;            // fp32 accumulation buffer, pre-zeroed
    bf16_t* dk;
    bf16_t* dv;
    int S, H, B;
    long long q_rs, k_rs, v_rs, do_rs, dq_rs, dk_rs, dv_rs;
    long long q_bs, k_bs, v_bs, do_bs, dq_bs, dk_bs, dv_bs;
    float scale, scale_log2;
};

#define NKB VT_NKB
#define KEYS_PER_WG (128 * NKB)
#define KIMG 0
#define DSIMG (KEYS_PER_WG * 128)
#define QTILE (2 * KEYS_PER_WG * 128)
#define LSEOFF (QTILE + 16384)
#define BWD_LDS (LSEOFF + 512)

typedef __attribute__((ext_vector_type(8))) short short8v;

__device__ __forceinline__ int swz_f(int row) {
    return (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 2) & 1);
}
__device__ __forceinline__ int swz_off(int row, int chunk) { return row * 128 + ((chunk ^ swz_f(row)) << 4); }

// two transposed 8-byte reads -> one 8 x bf16 MFMA operand
__device__ __forceinline__ bf16x8 tr_pair(const char* p0, const char* p1) {
    short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(p0));
    short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(p1));
    short8v v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v8);
}

template <bool PRESCALED>
__device__ __forceinline__ void BWD_BODY(const AttnBwdParams& p, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5, g = lane >> 4, ql = (lane & 15) >> 2, pl = lane & 3;
    const int nkblk = (p.S + KEYS_PER_WG - 1) / KEYS_PER_WG;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int kblk = id % nkblk, bh = id / nkblk;
    const int head = bh % p.H, b = bh / p.H;
    const int key0 = kblk * KEYS_PER_WG;

    const bf16_t* qb = p.q + (size_t)b * p.q_bs + head * 64;
    const bf16_t* kb_ = p.k + (size_t)b * p.k_bs + head * 64;
    const bf16_t* vb = p.v + (size_t)b * p.v_bs + head * 64;
    const bf16_t* dob = p.dout + (size_t)b * p.do_bs + head * 64;
    __amdgpu_buffer_rsrc_t rk = make_rsrc(kb_, (unsigned)((long long)(p.S - 1) * p.k_rs * 2 + 128));
    __amdgpu_buffer_rsrc_t rv = make_rsrc(vb, (unsigned)((long long)(p.S - 1) * p.v_rs * 2 + 128));
    // dQ accumulation buffer of this (batch, head): rows >= S are out of range -> the atomics are dropped by hardware
    __amdgpu_buffer_rsrc_t rdq = make_rsrc(p.dq + (size_t)b * p.dq_bs + head * 64, (unsigned)((long long)(p.S - 1) * p.dq_rs * 4 + 256));
    const float* lse_b = p.lse2 + (size_t)bh * p.S;
    const float* dl_b = p.delta + (size_t)bh * p.S;

    // ---- K block image: row reads give the B operand of S, transposed reads the B operand of dQ ----
#pragma unroll
    for (int j = 0; j < 4 * NKB; ++j) {
        int i = tid + 256 * j;
        int key = i >> 3, c = i & 7;
        u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, (int)((key0 + key) * p.k_rs * 2) + c * 16, 0, 0));
        *(u32x4*)(smem + KIMG + swz_off(key, c)) = v;
    }
    // ---- V fragments of this wave's keys (B operand of dP), resident for the whole kernel ----
    bf16x8 vf[NKB][4];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const int key = key0 + 32 * NKB * w + 32 * kb + r;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            vf[kb][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rv, (int)(key * p.v_rs * 2) + (16 * s + 8 * h) * 2, 0, 0));
    }
    // keys past the end of the sequence: a -1e30 column constant in the S accumulator makes P exactly 0
    // (only sub-tiles that straddle / pass the end take the masked path: the test is wave-uniform)
    float kmask[NKB];
    bool ktail[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        kmask[kb] = (key0 + 32 * NKB * w + 32 * kb + r) < p.S ? 0.f : -1.0e30f;
        ktail[kb] = (key0 + 32 * NKB * w + 32 * kb + 32) > p.S;
    }

    // ---- per-lane LDS offsets ----
    int rowrd[4];                         // row read of a 128-B-row image (Q / dO tile, K image), k-step s
#pragma unroll
    for (int s = 0; s < 4; ++s) rowrd[s] = r * 128 + (((2 * s + h) ^ swz_f(r)) << 4);
    int trA[2][2];                        // transposed read of the Q / dO tile, [dt][sec]
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int sec = 0; sec < 2; ++sec) {
            int fx = ((ql >> 1) << 2) | (sec << 1) | h;
            trA[dt][sec] = (4 * h + ql + 8 * sec) * 128 + (((4 * dt + 2 * (g & 1) + (pl >> 1)) ^ fx) << 4) + (pl & 1) * 8;
        }
    const int qs_w = w & 1, dt_w = w >> 1;
    int trQA[2], trQB[2];                 // dQ phase: dS image (A) and K image (B), [sec]
#pragma unroll
    for (int sec = 0; sec < 2; ++sec) {
        int fx = ((ql >> 1) << 2) | (h << 1) | sec;
        int row = (8 * h + ql + 4 * sec) * 128;
        trQA[sec] = row + (((4 * qs_w + 2 * (g & 1) + (pl >> 1)) ^ fx) << 4) + (pl & 1) * 8;
        trQB[sec] = row + (((4 * dt_w + 2 * (g & 1) + (pl >> 1)) ^ fx) << 4) + (pl & 1) * 8;
    }
    const int fr = swz_f(r);
    const int dq_voff = (int)((32 * qs_w + 4 * h) * p.dq_rs * 4) + (32 * dt_w + r) * 4;
    const int dq_rowb = (int)(p.dq_rs * 4);

    // ---- LDS-DMA staging of the Q / dO tiles: 16 blocks of 1 KiB (8 rows x 8 chunks); wave w moves blocks w, w+4, w+8, w+12.
    // The DMA writes lane l at block + 16 l, i.e. (row l>>3, physical chunk l&7); the swizzle goes on the SOURCE chunk.
    const int dma_rl = lane >> 3, dma_cp = lane & 7;
    auto dma_tiles = [&](int t) {
        const int q0 = t * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int bi = w + 4 * j;                       // wave-uniform
            const int row = 8 * (bi & 7) + dma_rl;          // row inside the 64-row tile
            int qrow = q0 + row;
            qrow = qrow < p.S ? qrow : p.S - 1;             // clamp (queries past the end are masked through lneg)
            const int chunk = dma_cp ^ swz_f(row);
            const bf16_t* src = ((bi < 8) ? qb + (size_t)qrow * p.q_rs : dob + (size_t)qrow * p.do_rs) + chunk * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + QTILE + bi * 1024), 16, 0, 0);
        }
    };
    // row constants, staged in the form the MFMA accumulators want:  S'' = Q K^T - lse2/c  and  dP' = dO V^T - delta.
    // threads t and t+128 stage the same value (keeps the loop body free of divergent branches).
    const int stat_i = tid & 63;
    const bool stat_is_lse = (tid & 64) == 0;
    const float* stat_src = stat_is_lse ? lse_b : dl_b;
    const float stat_mul = stat_is_lse ? (PRESCALED ? -1.0f : -1.0f / p.scale_log2) : -1.0f;
    const float stat_oob = stat_is_lse ? -1.0e30f : 0.f;
    const int stat_lds = LSEOFF + (tid & 127) * 4;
    auto load_stat = [&](int t) {
        int qi = t * 64 + stat_i;
        const bool ok = qi < p.S;
        qi = ok ? qi : p.S - 1;
        const float v = stat_src[qi] * stat_mul;
        return ok ? v : stat_oob;
    };

    f32x16 dk_acc[NKB][2], dv_acc[NKB][2];
#pragma unroll
    for (int a = 0; a < NKB; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) { dk_acc[a][c][i] = 0.f; dv_acc[a][c][i] = 0.f; }

    const float sc = p.scale_log2;
    const int nsteps = (p.S + 63) / 64;
    const char* qimg = smem + QTILE;
    const char* doimg = qimg + 8192;
    const float* lsel = (const float*)(smem + LSEOFF);
    char* dsimg = smem + DSIMG;
    const char* kimg_w = smem + KIMG + (32 * NKB * w) * 128;       // this wave's rows of the K image

    dma_tiles(0);
    *(float*)(smem + stat_lds) = load_stat(0);
    __syncthreads();
    float dq_prev[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) dq_prev[i] = 0.f;

    for (int t = 0; t < nsteps; ++t) {
        // ---- atomics of the previous step's dQ tile: issued here, a whole tile phase away from the next wait ----
#if VT_ABL != 1 && VT_ABL != 2
        if (t > 0) {
            const int soff = (int)((long long)(t - 1) * 64 * p.dq_rs * 4);
#pragma unroll
            for (int i = 0; i < 16; ++i)
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(dq_prev[i], rdq, dq_voff + ((i & 3) + 8 * (i >> 2)) * dq_rowb, soff, 0);
        }
#endif
        const float stat_next = load_stat(t + 1);

        // ---- software-pipelined tile schedule: tiles j = 0 .. 2*NKB-1 = (qs = j / NKB, kb = j % NKB) ----
        auto rowfrags = [&](int qs, bf16x8 (&qa)[4], bf16x8 (&doa)[4]) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                qa[s] = *(const bf16x8*)(qimg + qs * 4096 + rowrd[s]);
                doa[s] = *(const bf16x8*)(doimg + qs * 4096 + rowrd[s]);
            }
        };
        auto trfrags = [&](int qs, bf16x8 (&qT)[2][2], bf16x8 (&doT)[2][2]) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int ro = (32 * qs + 16 * s2) * 128;
                    qT[s2][dt] = tr_pair(qimg + ro + trA[dt][0], qimg + ro + trA[dt][1]);
                    doT[s2][dt] = tr_pair(doimg + ro + trA[dt][0], doimg + ro + trA[dt][1]);
                }
        };
        // accumulator init = row constants straight from LDS; accumulator register 4g'+e <-> q = 32qs + 8g' + 4h + e
        auto qk_init = [&](int qs, int kb, f32x16& sacc, f32x16& pacc) {
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                f32x4 a = *(const f32x4*)(lsel + 32 * qs + 8 * gg + 4 * h);
                f32x4 c = *(const f32x4*)(lsel + 64 + 32 * qs + 8 * gg + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) { sacc[4 * gg + e] = a[e]; pacc[4 * gg + e] = c[e]; }
            }
            if (ktail[kb]) {
#pragma unroll
                for (int i = 0; i < 16; ++i) sacc[i] += kmask[kb];
            }
        };
        // MFMA number m (0..7) of QK: k-step m>>1, S for even m (K fragment = row read of the K image), dP for odd m
        auto qk_mfma = [&](int m, int kb, const bf16x8 (&qa)[4], const bf16x8 (&doa)[4], f32x16& sacc, f32x16& pacc) {
            if ((m & 1) == 0) {
                const bf16x8 kfr = *(const bf16x8*)(kimg_w + kb * 4096 + rowrd[m >> 1]);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[m >> 1], kfr, sacc, 0, 0, 0);
            } else {
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(doa[m >> 1], vf[kb][m >> 1], pacc, 0, 0, 0);
            }
        };
        // MFMA number m (0..7) of PV: s2 = m>>2, dt = (m>>1)&1, dV for even m, dK for odd m
        auto pv_mfma = [&](int m, int kb, const bf16x8 (&qT)[2][2], const bf16x8 (&doT)[2][2], const unsigned (&pw)[8],
                           const unsigned (&dw)[8]) {
            const int s2 = m >> 2, dt = (m >> 1) & 1;
            if ((m & 1) == 0) {
                const u32x4 pb4 = {pw[4 * s2], pw[4 * s2 + 1], pw[4 * s2 + 2], pw[4 * s2 + 3]};
                dv_acc[kb][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(doT[s2][dt], __builtin_bit_cast(bf16x8, pb4), dv_acc[kb][dt], 0, 0, 0);
            } else {
                const u32x4 db4 = {dw[4 * s2], dw[4 * s2 + 1], dw[4 * s2 + 2], dw[4 * s2 + 3]};
                dk_acc[kb][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qT[s2][dt], __builtin_bit_cast(bf16x8, db4), dk_acc[kb][dt], 0, 0, 0);
            }
        };
        auto sm_pair = [&](int i, const f32x16& sacc, const f32x16& pacc, unsigned (&pw)[8], unsigned (&dw)[8]) {
            const float p0 = __builtin_amdgcn_exp2f(PRESCALED ? sacc[2 * i] : sacc[2 * i] * sc);
            const float p1 = __builtin_amdgcn_exp2f(PRESCALED ? sacc[2 * i + 1] : sacc[2 * i + 1] * sc);
            pw[i] = pack2(p0, p1);
            dw[i] = pack2(p0 * pacc[2 * i], p1 * pacc[2 * i + 1]);
        };
        // dS image: row = key (32 NKB w + 32 kb + r), 8 bytes = q 32qs + 8g' + 4h + (0..3)
        auto wr_ds = [&](int qs, int kb, const unsigned (&dw)[8]) {
            char* drow = dsimg + (32 * NKB * w + 32 * kb + r) * 128 + 8 * h;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                u32x2 two = {dw[2 * gg], dw[2 * gg + 1]};
                *(u32x2*)(drow + (((4 * qs + gg) ^ fr) << 4)) = two;
            }
        };
        {
            constexpr int NT = 2 * NKB;
            f32x16 sacc[2], pacc[2];
            unsigned pw[2][8], dw[2][8];
            bf16x8 qa[4], doa[4], qT[2][2], doT[2][2];
            rowfrags(0, qa, doa);
            qk_init(0, 0, sacc[0], pacc[0]);
#pragma unroll
            for (int m = 0; m < 8; ++m) qk_mfma(m, 0, qa, doa, sacc[0], pacc[0]);        // QK(0): exposed
            trfrags(0, qT, doT);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int cur = j & 1, nxt = cur ^ 1;
                const int jn = j + 1, jp = j - 1;
                if (jn == NKB) rowfrags(1, qa, doa);             // next tile starts query sub-slice 1
                if (jp == NKB) trfrags(1, qT, doT);              // previous tile was the first of sub-slice 1
                if (jn < NT) qk_init(jn / NKB, jn % NKB, sacc[nxt], pacc[nxt]);
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    if (j > 0) pv_mfma(m, jp % NKB, qT, doT, pw[nxt], dw[nxt]);
                    if (jn < NT) qk_mfma(m, jn % NKB, qa, doa, sacc[nxt], pacc[nxt]);
                    sm_pair(m, sacc[cur], pacc[cur], pw[cur], dw[cur]);
                }
                wr_ds(j / NKB, j % NKB, dw[cur]);
                __builtin_amdgcn_sched_barrier(0);           // keep each slot's loads and live ranges inside the slot
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) pv_mfma(m, (NT - 1) % NKB, qT, doT, pw[(NT - 1) & 1], dw[(NT - 1) & 1]);   // PV(last): exposed
        }
        __syncthreads();          // dS image complete; everybody is done with this step's Q / dO tiles and row constants

        // ---- next step's tiles by LDS-DMA (lands under the dQ phase) + its row constants ----
        dma_tiles(t + 1);
        *(float*)(smem + stat_lds) = stat_next;

#if VT_ABL != 1
        // ---- dQ tile (32 q x 32 d) of this wave over all keys of the block; operands fetched two k-steps ahead ----
        f32x16 dq_acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) dq_acc[i] = 0.f;
        {
            constexpr int NS = 8 * NKB;
            bf16x8 fa[3], fb[3];
#pragma unroll
            for (int s3 = 0; s3 < 2; ++s3) {
                fa[s3] = tr_pair(dsimg + s3 * 2048 + trQA[0], dsimg + s3 * 2048 + trQA[1]);
                fb[s3] = tr_pair(smem + KIMG + s3 * 2048 + trQB[0], smem + KIMG + s3 * 2048 + trQB[1]);
            }
#pragma unroll
            for (int s3 = 0; s3 < NS; ++s3) {
                if (s3 + 2 < NS) {
                    fa[(s3 + 2) % 3] = tr_pair(dsimg + (s3 + 2) * 2048 + trQA[0], dsimg + (s3 + 2) * 2048 + trQA[1]);
                    fb[(s3 + 2) % 3] = tr_pair(smem + KIMG + (s3 + 2) * 2048 + trQB[0], smem + KIMG + (s3 + 2) * 2048 + trQB[1]);
                }
                dq_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s3 % 3], fb[s3 % 3], dq_acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) dq_prev[i] = dq_acc[i] * p.scale;
#if VT_ABL == 2
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(dq_prev[i]));
#endif
#endif
        __syncthreads();          // DMA landed (the barrier's vmcnt(0) finds no atomics in flight), dS image free again
    }
#if VT_ABL != 1 && VT_ABL != 2
    {
        const int soff = (int)((long long)(nsteps - 1) * 64 * p.dq_rs * 4);
#pragma unroll
        for (int i = 0; i < 16; ++i)
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(dq_prev[i], rdq, dq_voff + ((i & 3) + 8 * (i >> 2)) * dq_rowb, soff, 0);
    }
#endif

    // ---- epilogue: dK^T, dV^T accumulators -> dk[key][d], dv[key][d] ----
    const float dk_mul = PRESCALED ? 0.6931471805599453f : p.scale;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const int key = key0 + 32 * NKB * w + 32 * kb + r;
        if (key < p.S) {
            bf16_t* dkp = p.dk + (size_t)b * p.dk_bs + (size_t)key * p.dk_rs + head * 64;
            bf16_t* dvp = p.dv + (size_t)b * p.dv_bs + (size_t)key * p.dv_rs + head * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    u32x2 a, c;
                    a[0] = pack2(dk_acc[kb][dt][4 * gg + 0] * dk_mul, dk_acc[kb][dt][4 * gg + 1] * dk_mul);
                    a[1] = pack2(dk_acc[kb][dt][4 * gg + 2] * dk_mul, dk_acc[kb][dt][4 * gg + 3] * dk_mul);
                    c[0] = pack2(dv_acc[kb][dt][4 * gg + 0], dv_acc[kb][dt][4 * gg + 1]);
                    c[1] = pack2(dv_acc[kb][dt][4 * gg + 2], dv_acc[kb][dt][4 * gg + 3]);
                    *(u32x2*)(dkp + 32 * dt + 8 * gg + 4 * h) = a;
                    *(u32x2*)(dvp + 32 * dt + 8 * gg + 4 * h) = c;
                }
        }
    }
}

template <bool PRESCALED>
__global__ __launch_bounds__(256, 1) void BWD_KERNEL(AttnBwdParams p) {
    __shared__ __attribute__((aligned(1024))) char smem[BWD_LDS];
    BWD_BODY<PRESCALED>(p, smem);
}

// delta[b,h,s] = sum_d dO[b,s,h,d] * O[b,s,h,d]   (8 lanes per (s,h) row of 64 elements)
__global__ __launch_bounds__(256) void DELTA_KERNEL(const bf16_t* o, const bf16_t* dout, float* delta,
                                                            int B, int H, int S, long long o_rs, long long do_rs,
                                                            long long o_bs, long long do_bs) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long row = gid >> 3;          // (b, s, h) flattened with h fastest
    const int sub = (int)(gid & 7);
    const long long total = (long long)B * S * H;
    float acc = 0.f;
    long long bs = 0; int hh = 0;
    if (row < total) {
        hh = (int)(row % H);
        bs = row / H;
        const int s = (int)(bs % S);
        const int b = (int)(bs / S);
        u32x4 a = *(const u32x4*)(o + (size_t)b * o_bs + (size_t)s * o_rs + hh * 64 + sub * 8);
        u32x4 c = *(const u32x4*)(dout + (size_t)b * do_bs + (size_t)s * do_rs + hh * 64 + sub * 8);
        float fa[8], fc[8];
        unpack8(a, fa);
        unpack8(c, fc);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += fa[i] * fc[i];
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (row < total && sub == 0) {
        const int s = (int)(bs % S);
        const int b = (int)(bs / S);
        delta[((size_t)b * H + hh) * S + s] = acc;
    }
}

extern "C" int BWD_ENTRY(const void* q, const void* k, const void* v, const void* o, const void* dout,
                                const float* lse2, float* delta_ws, float* dq_f32, void* dk, void* dv,
                                int B, int H, int S,
                                long long q_rs, long long k_rs, long long v_rs, long long o_rs, long long do_rs,
                                long long dq_rs, long long dk_rs, long long dv_rs,
                                long long q_bs, long long k_bs, long long v_bs, long long o_bs, long long do_bs,
                                long long dq_bs, long long dk_bs, long long dv_bs,
                                float softmax_scale, int q_prescaled, void* stream) {
    if (B <= 0 || H <= 0 || S <= 0) return VT_ERR_BAD_SHAPE;
    if ((long long)S * dq_rs * 4 >= 0x7fffffffLL) return VT_ERR_BAD_SHAPE;
    if ((q_rs % 8) || (k_rs % 8) || (v_rs % 8) || (o_rs % 8) || (do_rs % 8) || (dk_rs % 4) || (dv_rs % 4)) return VT_ERR_BAD_SHAPE;
    if ((q_bs % 8) || (k_bs % 8) || (v_bs % 8) || (o_bs % 8) || (do_bs % 8) || (dk_bs % 4) || (dv_bs % 4)) return VT_ERR_BAD_SHAPE;
    if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)o) | ((uintptr_t)dout)) & 15) return VT_ERR_BAD_ALIGN;
    if ((((uintptr_t)dk) | ((uintptr_t)dv)) & 7) return VT_ERR_BAD_ALIGN;
    const long long lim = 0x7fffffffLL;
    if ((long long)S * q_rs * 2 >= lim || (long long)S * k_rs * 2 >= lim || (long long)S * v_rs * 2 >= lim ||
        (long long)S * do_rs * 2 >= lim) return VT_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    {
        const long long total = (long long)B * S * H * 8;
        const int blocks = (int)((total + 255) / 256);
        hipLaunchKernelGGL(DELTA_KERNEL, dim3(blocks), dim3(256), 0, st, (const bf16_t*)o, (const bf16_t*)dout,
                           delta_ws, B, H, S, o_rs, do_rs, o_bs, do_bs);
    }
    AttnBwdParams p;
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.dout = (const bf16_t*)dout;
    p.lse2 = lse2; p.delta = delta_ws; p.dq = dq_f32; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv;
    p.S = S; p.H = H; p.B = B;
    p.q_rs = q_rs; p.k_rs = k_rs; p.v_rs = v_rs; p.do_rs = do_rs; p.dq_rs = dq_rs; p.dk_rs = dk_rs; p.dv_rs = dv_rs;
    p.q_bs = q_bs; p.k_bs = k_bs; p.v_bs = v_bs; p.do_bs = do_bs; p.dq_bs = dq_bs; p.dk_bs = dk_bs; p.dv_bs = dv_bs;
    p.scale = softmax_scale; p.scale_log2 = softmax_scale * 1.4426950408889634f;
    const int nkb = (S + KEYS_PER_WG - 1) / KEYS_PER_WG;
    if (q_prescaled) hipLaunchKernelGGL(BWD_KERNEL<true>, dim3(nkb * H * B), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(BWD_KERNEL<false>, dim3(nkb * H * B), dim3(256), 0, st, p);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}
