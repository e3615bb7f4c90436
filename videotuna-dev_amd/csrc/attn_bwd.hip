// Flash-attention backward (dQ, dK, dV), head_dim 64, bf16 in / fp32 accumulate, non-causal, gfx950.
//
// Autograd of the F.scaled_dot_product_attention call that the reference reaches through
// videotuna/models/cogvideo_hf/cogvideo_pl.py:865-871 (loss.backward() under PL).
// P is recomputed from Q, K and the forward's log2-domain LSE; the S x S matrices never touch HBM.
//
// Structure: one workgroup = 4 waves (one per SIMD, up to 512 registers each) = 256 keys of one
// (batch, head); wave w owns keys [64w, 64w+64) and keeps dK^T and dV^T for them in 128 accumulator
// registers while the workgroup sweeps all queries in steps of 64 rows.
//   * S = Q K^T and dP = dO V^T are computed with the KEY on the MFMA lane (K / V fragments live in
//     registers for the whole kernel), so the fp32 tiles P and dS are, after bf16 packing, directly
//     the B operands of dV^T += dO^T P and dK^T += Q^T dS (A operands = transposed LDS reads of the
//     dO / Q tiles, ds_read_b64_tr_b16).
//   * -delta (delta = rowsum(dO*O)) is loaded as the initial accumulator of dP.
//   * only dS crosses LDS: every wave writes its [64 keys][64 q] part of a [256][64] image; after
//     one barrier each wave computes one 32x32 tile of dQ (its (q-half, d-half)) over all 256 keys
//     and adds it to a fp32 dQ buffer with global_atomic_add_f32 (one register of a 32x32
//     accumulator = two full 128-B row segments, the full-rate atomic shape).
// All four LDS images (K block, dS, Q tile, dO tile) have 128-B rows and share one XOR swizzle that
// is conflict-free for both the row reads (ds_read_b128) and the transposed reads.
#include "common.h"

// Experiment hooks: the same source can be compiled a second time under another symbol suffix / variant flags
// (tools/build_variants.sh) so kernel variants are A/B-timed in ONE process.  The shipped build defines neither.
#ifndef VT_SUFFIX
#define VT_SUFFIX
#endif
#ifndef VT_PIPE
#define VT_PIPE 1
#endif
#ifndef VT_DRAIN
#define VT_DRAIN 0    // 1 = spread each step's 16 dQ atomics over the next step's MFMA slots (measured SLOWER: 10.5 vs 8.9 ms)
#endif
#ifndef VT_DQSHIFT
#define VT_DQSHIFT 0   // 1 = the dQ MFMAs of step t-1 run inside step t VALU-bound slots (measured SLOWER: 11.5 vs 8.9 ms, the atomics then leave every wave in one burst right after the barrier)
#endif
#ifndef VT_ABL
#define VT_ABL 0      // timing-only ablations (results are WRONG): 1 = no dQ phase, 2 = no dQ atomics, 3 = no exp2, 4 = no dS image write
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

#define KIMG 0
#define DSIMG 32768
#define QTILE 98304
#define LSEOFF 131072
#define BWD_LDS 132096

typedef __attribute__((ext_vector_type(8))) short short8v;

__device__ __forceinline__ int swz_f(int row) {
    return (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 2) & 1);
}
__device__ __forceinline__ int swz_off(int row, int chunk) { return row * 128 + ((chunk ^ swz_f(row)) << 4); }

// two transposed 8-byte reads (rows +0..3 and rows +sec_stride) -> one 8 x bf16 MFMA operand
__device__ __forceinline__ bf16x8 tr_pair(const char* p0, const char* p1) {
    short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(p0));
    short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(p1));
    short8v v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v8);
}

template <bool RAGGED, bool PRESCALED>
__device__ __forceinline__ void BWD_BODY(const AttnBwdParams& p, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5, g = lane >> 4, ql = (lane & 15) >> 2, pl = lane & 3;
    const int nkb = (p.S + 255) / 256;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int kblk = id % nkb, bh = id / nkb;
    const int head = bh % p.H, b = bh / p.H;
    const int key0 = kblk * 256;

    const bf16_t* qb = p.q + (size_t)b * p.q_bs + head * 64;
    const bf16_t* kb_ = p.k + (size_t)b * p.k_bs + head * 64;
    const bf16_t* vb = p.v + (size_t)b * p.v_bs + head * 64;
    const bf16_t* dob = p.dout + (size_t)b * p.do_bs + head * 64;
    __amdgpu_buffer_rsrc_t rq = make_rsrc(qb, (unsigned)((long long)(p.S - 1) * p.q_rs * 2 + 128));
    __amdgpu_buffer_rsrc_t rk = make_rsrc(kb_, (unsigned)((long long)(p.S - 1) * p.k_rs * 2 + 128));
    __amdgpu_buffer_rsrc_t rv = make_rsrc(vb, (unsigned)((long long)(p.S - 1) * p.v_rs * 2 + 128));
    __amdgpu_buffer_rsrc_t rdo = make_rsrc(dob, (unsigned)((long long)(p.S - 1) * p.do_rs * 2 + 128));
    // dQ accumulation buffer of this (batch, head): rows >= S are out of range -> the atomics are dropped by hardware
    __amdgpu_buffer_rsrc_t rdq = make_rsrc(p.dq + (size_t)b * p.dq_bs + head * 64, (unsigned)((long long)(p.S - 1) * p.dq_rs * 4 + 256));
    const float* lse_b = p.lse2 + (size_t)bh * p.S;
    const float* dl_b = p.delta + (size_t)bh * p.S;

    // ---- K block image (B operand of dQ) : 256 keys x 8 chunks ----
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int i = tid + 256 * j;
        int key = i >> 3, c = i & 7;
        u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, (int)((key0 + key) * p.k_rs * 2) + c * 16, 0, 0));
        *(u32x4*)(smem + KIMG + swz_off(key, c)) = v;
    }
    // ---- K / V fragments of this wave's 64 keys (B operands of S and dP), resident for the whole kernel ----
    bf16x8 kf[2][4], vf[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int key = key0 + 64 * w + 32 * kb + r;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[kb][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rk, (int)(key * p.k_rs * 2) + (16 * s + 8 * h) * 2, 0, 0));
            vf[kb][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rv, (int)(key * p.v_rs * 2) + (16 * s + 8 * h) * 2, 0, 0));
        }
    }
    // keys past the end of the sequence: a -1e30 column constant in the S accumulator makes P exactly 0
    float kmask[2] = {0.f, 0.f};
    if (RAGGED) {
        kmask[0] = (key0 + 64 * w + r) < p.S ? 0.f : -1.0e30f;
        kmask[1] = (key0 + 64 * w + 32 + r) < p.S ? 0.f : -1.0e30f;
    }

    // ---- per-lane LDS offsets ----
    int rowrd[4];                         // row read of the Q / dO tile (A operand of S / dP), k-step s
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
    // dQ atomics: byte offset of (row 32qs_w + 4h, column 32dt_w + r) inside a 64-row step; register i adds
    // ((i&3) + 8(i>>2)) rows
    const int dq_voff = (int)((32 * qs_w + 4 * h) * p.dq_rs * 4) + (32 * dt_w + r) * 4;
    const int dq_rowb = (int)(p.dq_rs * 4);

    // ---- staging of the Q / dO tiles (64 rows x 8 chunks each): 2 + 2 chunks per thread (branch free) ----
    int st_voq[2], st_vodo[2], st_lds[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int i = tid + 256 * j;
        st_voq[j] = (int)((i >> 3) * p.q_rs * 2) + (i & 7) * 16;
        st_vodo[j] = (int)((i >> 3) * p.do_rs * 2) + (i & 7) * 16;
        st_lds[j] = swz_off(i >> 3, i & 7);
    }
    // row constants, staged in the form the MFMA accumulators want:  S'' = Q K^T - lse2/c  and  dP' = dO V^T - delta.
    // threads t and t+128 stage the same value (keeps the loop body free of divergent branches).
    const int stat_i = tid & 63;
    const bool stat_is_lse = (tid & 64) == 0;
    const float* stat_src = stat_is_lse ? lse_b : dl_b;
    const float stat_mul = stat_is_lse ? (PRESCALED ? -1.0f : -1.0f / p.scale_log2) : -1.0f;
    const int stat_lds = LSEOFF + (tid & 127) * 4;
    u32x4 gq[2], gdo[2];
    float gstat = 0.f;
    auto gload = [&](int t) {
        const int q0 = t * 64;
        const int sq = (int)((long long)q0 * p.q_rs * 2), sdo = (int)((long long)q0 * p.do_rs * 2);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            gq[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rq, st_voq[j], sq, 0));
            gdo[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdo, st_vodo[j], sdo, 0));
        }
        int qi = q0 + stat_i;
        const bool ok = qi < p.S;
        qi = ok ? qi : p.S - 1;
        const float v = stat_src[qi] * stat_mul;
        gstat = ok ? v : 0.f;
    };
    auto lstore = [&](int buf) {
        char* base = smem + QTILE + buf * 16384;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            *(u32x4*)(base + st_lds[j]) = gq[j];
            *(u32x4*)(base + 8192 + st_lds[j]) = gdo[j];
        }
        *(float*)(smem + stat_lds + buf * 512) = gstat;
    };

    f32x16 dk_acc[2][2], dv_acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) { dk_acc[a][c][i] = 0.f; dv_acc[a][c][i] = 0.f; }

    const float sc = p.scale_log2;
    const int nsteps = (p.S + 63) / 64;
    // dQ tile of the PREVIOUS step: its 16 atomics are drained one at a time between the MFMAs of the current step
    // (a 16-deep burst per wave, from every CU at once, saturates the memory-side atomic units)
    float dq_prev[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) dq_prev[i] = 0.f;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int t = 0; t < nsteps; ++t) {
        const int buf = t & 1;
        gload(t + 1);                                  // past the end: bounds-checked loads return zeros
        const char* qimg = smem + QTILE + buf * 16384;
        const char* doimg = qimg + 8192;
        const float* lsel = (const float*)(smem + LSEOFF + buf * 512);
        char* dsimg = smem + DSIMG + buf * 32768;

#if VT_PIPE
        // ---- software-pipelined tile schedule -------------------------------------------------------------
        // Tiles j = 0..3 = (qs, kb) = (0,0) (0,1) (1,0) (1,1).  With one wave per SIMD nothing else hides an MFMA's
        // latency, so the VALU work of tile j (exp2, dS, bf16 packing: SM) is interleaved in program order with 16
        // MFMAs that do not depend on it: the dV/dK products of tile j-1 (PV) and the S/dP products of tile j+1 (QK).
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
        // accumulator init = row constants straight from LDS:  S'' starts at -lse2/c (+ key mask), dP' at -delta;
        // accumulator register 4g'+e <-> q = 32qs + 8g' + 4h + e
        auto qk_init = [&](int qs, int kb, f32x16& sacc, f32x16& pacc) {
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                f32x4 a = *(const f32x4*)(lsel + 32 * qs + 8 * gg + 4 * h);
                f32x4 c = *(const f32x4*)(lsel + 64 + 32 * qs + 8 * gg + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) { sacc[4 * gg + e] = RAGGED ? a[e] + kmask[kb] : a[e]; pacc[4 * gg + e] = c[e]; }
            }
        };
        // MFMA number m (0..7) of QK: k-step m>>1, S for even m, dP for odd m
        auto qk_mfma = [&](int m, int kb, const bf16x8 (&qa)[4], const bf16x8 (&doa)[4], f32x16& sacc, f32x16& pacc) {
            if ((m & 1) == 0) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[m >> 1], kf[kb][m >> 1], sacc, 0, 0, 0);
            else pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(doa[m >> 1], vf[kb][m >> 1], pacc, 0, 0, 0);
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
#if VT_ABL == 3
            const float p0 = sacc[2 * i], p1 = sacc[2 * i + 1];
#else
            const float p0 = __builtin_amdgcn_exp2f(PRESCALED ? sacc[2 * i] : sacc[2 * i] * sc);
            const float p1 = __builtin_amdgcn_exp2f(PRESCALED ? sacc[2 * i + 1] : sacc[2 * i + 1] * sc);
#endif
            pw[i] = pack2(p0, p1);
            dw[i] = pack2(p0 * pacc[2 * i], p1 * pacc[2 * i + 1]);
        };
        auto wr_ds = [&](int qs, int kb, const unsigned (&dw)[8]) {
#if VT_ABL == 4
            return;
#endif
            char* drow = dsimg + (64 * w + 32 * kb + r) * 128 + 8 * h;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                u32x2 two = {dw[2 * gg], dw[2 * gg + 1]};
                *(u32x2*)(drow + (((4 * qs + gg) ^ fr) << 4)) = two;
            }
        };
        const int soff_prev = (int)((long long)(t - 1) * 64 * p.dq_rs * 4);     // t == 0: negative row offset -> dropped? no: guarded
        auto drain = [&](int i) {
#if VT_DRAIN && VT_ABL != 1 && VT_ABL != 2
            if (t > 0)
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(dq_prev[i], rdq, dq_voff + ((i & 3) + 8 * (i >> 2)) * dq_rowb,
                                                                soff_prev, 0);
#endif
        };
        f32x16 sA, pA, sB, pB;
        bf16x8 qa[4], doa[4], qT[2][2], doT[2][2];
        unsigned pwA[8], dwA[8], pwB[8], dwB[8];
#if VT_DQSHIFT
        // dQ tile of the PREVIOUS step (its dS image sits in the other buffer): 16 MFMAs that depend on nothing in this
        // step, issued inside the two slots that have only 8 MFMAs for 8 softmax pairs (operands one k-step ahead)
        const char* dsprev = smem + DSIMG + (buf ^ 1) * 32768;
        f32x16 dqp;
#pragma unroll
        for (int i = 0; i < 16; ++i) dqp[i] = 0.f;
        bf16x8 ra[2], rb[2];
        auto dq_load = [&](int s3, int slot) {
            ra[slot] = tr_pair(dsprev + s3 * 2048 + trQA[0], dsprev + s3 * 2048 + trQA[1]);
            rb[slot] = tr_pair(smem + KIMG + s3 * 2048 + trQB[0], smem + KIMG + s3 * 2048 + trQB[1]);
        };
        dq_load(0, 0);
#endif
        rowfrags(0, qa, doa);
        qk_init(0, 0, sA, pA);
#pragma unroll
        for (int m = 0; m < 8; ++m) qk_mfma(m, 0, qa, doa, sA, pA);            // QK(0): exposed
        trfrags(0, qT, doT);
        // slot 0: QK(1) || SM(0)
        qk_init(0, 1, sB, pB);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            qk_mfma(m, 1, qa, doa, sB, pB);
#if VT_DQSHIFT
            dq_load(m + 1, (m + 1) & 1);                  // k-step 8 is loaded here for slot 3
            dqp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra[m & 1], rb[m & 1], dqp, 0, 0, 0);
#endif
            sm_pair(m, sA, pA, pwA, dwA);
            if (m & 1) drain(m >> 1);
        }
        rowfrags(1, qa, doa);
        wr_ds(0, 0, dwA);
        // slot 1: PV(0) + QK(2) || SM(1)
        qk_init(1, 0, sA, pA);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            pv_mfma(m, 0, qT, doT, pwA, dwA);
            qk_mfma(m, 0, qa, doa, sA, pA);
            sm_pair(m, sB, pB, pwB, dwB);
            if (m & 1) drain(4 + (m >> 1));
        }
        wr_ds(0, 1, dwB);
        // slot 2: PV(1) + QK(3) || SM(2)
        qk_init(1, 1, sB, pB);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            pv_mfma(m, 1, qT, doT, pwB, dwB);
            qk_mfma(m, 1, qa, doa, sB, pB);
            sm_pair(m, sA, pA, pwA, dwA);
            if (m & 1) drain(8 + (m >> 1));
        }
        wr_ds(1, 0, dwA);
        trfrags(1, qT, doT);
        // slot 3: PV(2) || SM(3)
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            pv_mfma(m, 0, qT, doT, pwA, dwA);
#if VT_DQSHIFT
            if (m + 1 < 8) dq_load(8 + m + 1, (m + 1) & 1);
            dqp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra[m & 1], rb[m & 1], dqp, 0, 0, 0);
#endif
            sm_pair(m, sB, pB, pwB, dwB);
            if (m & 1) drain(12 + (m >> 1));
        }
        wr_ds(1, 1, dwB);
        // PV(3): exposed
#pragma unroll
        for (int m = 0; m < 8; ++m) pv_mfma(m, 1, qT, doT, pwB, dwB);
#else
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            // row constants of this 32-query sub-slice: accumulator register 4g'+e <-> q = 32qs + 8g' + 4h + e
            f32x16 lneg, dneg;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                f32x4 a = *(const f32x4*)(lsel + 32 * qs + 8 * gg + 4 * h);
                f32x4 c = *(const f32x4*)(lsel + 64 + 32 * qs + 8 * gg + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) { lneg[4 * gg + e] = a[e]; dneg[4 * gg + e] = c[e]; }
            }
            bf16x8 qa[4], doa[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                qa[s] = *(const bf16x8*)(qimg + qs * 4096 + rowrd[s]);
                doa[s] = *(const bf16x8*)(doimg + qs * 4096 + rowrd[s]);
            }
            bf16x8 qT[2][2], doT[2][2];   // [s'][dt]
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int ro = (32 * qs + 16 * s2) * 128;
                    qT[s2][dt] = tr_pair(qimg + ro + trA[dt][0], qimg + ro + trA[dt][1]);
                    doT[s2][dt] = tr_pair(doimg + ro + trA[dt][0], doimg + ro + trA[dt][1]);
                }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f32x16 sacc = lneg, pacc = dneg;
                if (RAGGED) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) sacc[i] += kmask[kb];
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[s], kf[kb][s], sacc, 0, 0, 0);
                    pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(doa[s], vf[kb][s], pacc, 0, 0, 0);
                }
                // P = exp2(c * S''), dS = P * dP'; both packed to bf16 pairs (B operands + dS image)
                unsigned int pw[8], dw[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float p0 = __builtin_amdgcn_exp2f(PRESCALED ? sacc[2 * i] : sacc[2 * i] * sc);
                    const float p1 = __builtin_amdgcn_exp2f(PRESCALED ? sacc[2 * i + 1] : sacc[2 * i + 1] * sc);
                    pw[i] = pack2(p0, p1);
                    dw[i] = pack2(p0 * pacc[2 * i], p1 * pacc[2 * i + 1]);
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const u32x4 pb4 = {pw[4 * s2], pw[4 * s2 + 1], pw[4 * s2 + 2], pw[4 * s2 + 3]};
                    const u32x4 db4 = {dw[4 * s2], dw[4 * s2 + 1], dw[4 * s2 + 2], dw[4 * s2 + 3]};
                    const bf16x8 pb = __builtin_bit_cast(bf16x8, pb4), dsb = __builtin_bit_cast(bf16x8, db4);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        dv_acc[kb][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(doT[s2][dt], pb, dv_acc[kb][dt], 0, 0, 0);
                        dk_acc[kb][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qT[s2][dt], dsb, dk_acc[kb][dt], 0, 0, 0);
                    }
                }
                // dS image: row = key (64w + 32kb + r), 8 bytes = q 32qs + 8g' + 4h + (0..3)
                char* drow = dsimg + (64 * w + 32 * kb + r) * 128 + 8 * h;
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    u32x2 two = {dw[2 * gg], dw[2 * gg + 1]};
                    *(u32x2*)(drow + (((4 * qs + gg) ^ fr) << 4)) = two;
                }
            }
        }
#endif
        lstore(buf ^ 1);
        __syncthreads();

#if VT_ABL != 1
#if VT_PIPE && VT_DQSHIFT
        // ---- atomics of the dQ tile that was accumulated during this step (it belongs to step t-1) ----
        if (t > 0)
        {
            const int soff = (int)((long long)(t - 1) * 64 * p.dq_rs * 4);
#if VT_ABL == 2
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(dqp[i]));
#else
#pragma unroll
            for (int i = 0; i < 16; ++i)
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(dqp[i] * p.scale, rdq,
                                                                dq_voff + ((i & 3) + 8 * (i >> 2)) * dq_rowb, soff, 0);
#endif
        }
#else
        // ---- dQ tile (32 q x 32 d) of this wave over all 256 keys ----
        f32x16 dq_acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) dq_acc[i] = 0.f;
        {
            // operands are fetched two k-steps ahead of the MFMA that consumes them (one wave per SIMD: nothing else
            // hides the LDS latency)
            bf16x8 fa[3], fb[3];
#pragma unroll
            for (int s3 = 0; s3 < 2; ++s3) {
                fa[s3] = tr_pair(dsimg + s3 * 2048 + trQA[0], dsimg + s3 * 2048 + trQA[1]);
                fb[s3] = tr_pair(smem + KIMG + s3 * 2048 + trQB[0], smem + KIMG + s3 * 2048 + trQB[1]);
            }
#pragma unroll
            for (int s3 = 0; s3 < 16; ++s3) {
                if (s3 + 2 < 16) {
                    fa[(s3 + 2) % 3] = tr_pair(dsimg + (s3 + 2) * 2048 + trQA[0], dsimg + (s3 + 2) * 2048 + trQA[1]);
                    fb[(s3 + 2) % 3] = tr_pair(smem + KIMG + (s3 + 2) * 2048 + trQB[0], smem + KIMG + (s3 + 2) * 2048 + trQB[1]);
                }
                dq_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s3 % 3], fb[s3 % 3], dq_acc, 0, 0, 0);
            }
        }
        {
            const int soff = (int)((long long)(t) * 64 * p.dq_rs * 4);
#if VT_ABL == 2
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(dq_acc[i]));
#else
#pragma unroll
            for (int i = 0; i < 16; ++i)
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(dq_acc[i] * p.scale, rdq,
                                                                dq_voff + ((i & 3) + 8 * (i >> 2)) * dq_rowb, soff, 0);
#endif
        }
#endif
#endif
    }

#if VT_PIPE && VT_DQSHIFT && VT_ABL != 1
    {   // the last step's dQ tile: nobody computes it "one step later"
        const char* dslast = smem + DSIMG + ((nsteps - 1) & 1) * 32768;
        f32x16 dq_acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) dq_acc[i] = 0.f;
        {
            // operands are fetched two k-steps ahead of the MFMA that consumes them (one wave per SIMD: nothing else
            // hides the LDS latency)
            bf16x8 fa[3], fb[3];
#pragma unroll
            for (int s3 = 0; s3 < 2; ++s3) {
                fa[s3] = tr_pair(dslast + s3 * 2048 + trQA[0], dslast + s3 * 2048 + trQA[1]);
                fb[s3] = tr_pair(smem + KIMG + s3 * 2048 + trQB[0], smem + KIMG + s3 * 2048 + trQB[1]);
            }
#pragma unroll
            for (int s3 = 0; s3 < 16; ++s3) {
                if (s3 + 2 < 16) {
                    fa[(s3 + 2) % 3] = tr_pair(dslast + (s3 + 2) * 2048 + trQA[0], dslast + (s3 + 2) * 2048 + trQA[1]);
                    fb[(s3 + 2) % 3] = tr_pair(smem + KIMG + (s3 + 2) * 2048 + trQB[0], smem + KIMG + (s3 + 2) * 2048 + trQB[1]);
                }
                dq_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s3 % 3], fb[s3 % 3], dq_acc, 0, 0, 0);
            }
        }
        {
            const int soff = (int)((long long)(nsteps - 1) * 64 * p.dq_rs * 4);
#if VT_ABL == 2
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(dq_acc[i]));
#else
#pragma unroll
            for (int i = 0; i < 16; ++i)
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(dq_acc[i] * p.scale, rdq,
                                                                dq_voff + ((i & 3) + 8 * (i >> 2)) * dq_rowb, soff, 0);
#endif
        }
    }
#endif
    // ---- epilogue: dK^T, dV^T accumulators -> dk[key][d], dv[key][d] ----
    const float dk_mul = PRESCALED ? 0.6931471805599453f : p.scale;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int key = key0 + 64 * w + 32 * kb + r;
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
    __shared__ __attribute__((aligned(16))) char smem[BWD_LDS];
    const int nkb = (p.S + 255) / 256;
    const int kblk = xcd_remap(blockIdx.x, gridDim.x) % nkb;
    if ((kblk + 1) * 256 > p.S) BWD_BODY<true, PRESCALED>(p, smem);   // block-uniform: only the last key block is ragged
    else BWD_BODY<false, PRESCALED>(p, smem);
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
    const int nkb = (S + 255) / 256;
    if (q_prescaled) hipLaunchKernelGGL(BWD_KERNEL<true>, dim3(nkb * H * B), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(BWD_KERNEL<false>, dim3(nkb * H * B), dim3(256), 0, st, p);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}
