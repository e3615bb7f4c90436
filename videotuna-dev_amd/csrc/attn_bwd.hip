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

__global__ __launch_bounds__(256, 1) void attn_bwd_hd64_kernel(AttnBwdParams p) {
    __shared__ __attribute__((aligned(16))) char smem[BWD_LDS];
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
    // ---- K / V fragments of this wave's 64 keys (B operands of S and dP) ----
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
    const bool kvalid0 = (key0 + 64 * w + r) < p.S;
    const bool kvalid1 = (key0 + 64 * w + 32 + r) < p.S;
    const bool ragged = (key0 + 256) > p.S;

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

    // ---- staging of the Q / dO tiles (64 rows x 8 chunks each): 2 + 2 chunks per thread ----
    int st_row[2], st_c[2], st_lds[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int i = tid + 256 * j;
        st_row[j] = i >> 3;
        st_c[j] = i & 7;
        st_lds[j] = swz_off(st_row[j], st_c[j]);
    }
    u32x4 gq[2], gdo[2];
    float gstat = 0.f;
    auto gload = [&](int t) {
        const int q0 = t * 64;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            gq[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rq, (int)((q0 + st_row[j]) * p.q_rs * 2) + st_c[j] * 16, 0, 0));
            gdo[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdo, (int)((q0 + st_row[j]) * p.do_rs * 2) + st_c[j] * 16, 0, 0));
        }
        if (tid < 128) {
            const int qi = q0 + (tid & 63);
            const float* src = (tid < 64) ? lse_b : dl_b;
            gstat = (qi < p.S) ? src[qi] : 0.f;
        }
    };
    auto lstore = [&](int buf) {
        char* base = smem + QTILE + buf * 16384;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            *(u32x4*)(base + st_lds[j]) = gq[j];
            *(u32x4*)(base + 8192 + st_lds[j]) = gdo[j];
        }
        if (tid < 128) *(float*)(smem + LSEOFF + buf * 512 + tid * 4) = gstat;
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
    gload(0);
    lstore(0);
    __syncthreads();
    for (int t = 0; t < nsteps; ++t) {
        const int buf = t & 1;
        if (t + 1 < nsteps) gload(t + 1);
        const char* qimg = smem + QTILE + buf * 16384;
        const char* doimg = qimg + 8192;
        const float* lsel = (const float*)(smem + LSEOFF + buf * 512);
        char* dsimg = smem + DSIMG + buf * 32768;

#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            // row constants for this 32-query sub-slice: reg 4g'+e <-> q = 32qs + 8g' + 4h + e
            f32x4 l4[4], d4[4];
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                l4[gg] = *(const f32x4*)(lsel + 32 * qs + 8 * gg + 4 * h);
                d4[gg] = *(const f32x4*)(lsel + 64 + 32 * qs + 8 * gg + 4 * h);
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
                f32x16 sacc, pacc;
#pragma unroll
                for (int i = 0; i < 16; ++i) { sacc[i] = 0.f; pacc[i] = -d4[i >> 2][i & 3]; }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[s], kf[kb][s], sacc, 0, 0, 0);
                    pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(doa[s], vf[kb][s], pacc, 0, 0, 0);
                }
                const bool kv = kb == 0 ? kvalid0 : kvalid1;
                float pr[16], dsr[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float pv = __builtin_amdgcn_exp2f(sacc[i] * sc - l4[i >> 2][i & 3]);
                    if (ragged) pv = kv ? pv : 0.f;
                    pr[i] = pv;
                    dsr[i] = pv * pacc[i];
                }
                bf16x8 pb[2], dsb[2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        pb[s2][j] = (bf16_t)pr[8 * s2 + j];
                        dsb[s2][j] = (bf16_t)dsr[8 * s2 + j];
                    }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        dv_acc[kb][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(doT[s2][dt], pb[s2], dv_acc[kb][dt], 0, 0, 0);
                        dk_acc[kb][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qT[s2][dt], dsb[s2], dk_acc[kb][dt], 0, 0, 0);
                    }
                // dS image: row = key (64w + 32kb + r), 8 bytes = q 32qs + 8g' + 4h + (0..3)
                char* drow = dsimg + (64 * w + 32 * kb + r) * 128 + 8 * h;
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    u32x2 two;
                    two[0] = pack2(dsr[4 * gg + 0], dsr[4 * gg + 1]);
                    two[1] = pack2(dsr[4 * gg + 2], dsr[4 * gg + 3]);
                    *(u32x2*)(drow + (((4 * qs + gg) ^ fr) << 4)) = two;
                }
            }
        }
        if (t + 1 < nsteps) lstore(buf ^ 1);
        __syncthreads();

        // ---- dQ tile (32 q x 32 d) of this wave over all 256 keys ----
        f32x16 dq_acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) dq_acc[i] = 0.f;
#pragma unroll
        for (int s3 = 0; s3 < 16; ++s3) {
            bf16x8 a = tr_pair(dsimg + s3 * 2048 + trQA[0], dsimg + s3 * 2048 + trQA[1]);
            bf16x8 bb = tr_pair(smem + KIMG + s3 * 2048 + trQB[0], smem + KIMG + s3 * 2048 + trQB[1]);
            dq_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, dq_acc, 0, 0, 0);
        }
        {
            float* dqp = p.dq + (size_t)b * p.dq_bs + head * 64 + 32 * dt_w + r;
            const int qbase = t * 64 + 32 * qs_w + 4 * h;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int qrow = qbase + (i & 3) + 8 * (i >> 2);
                if (qrow < p.S) atomicAdd(dqp + (size_t)qrow * p.dq_rs, dq_acc[i] * p.scale);
            }
        }
    }

    // ---- epilogue: dK^T, dV^T accumulators -> dk[key][d], dv[key][d] ----
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
                    a[0] = pack2(dk_acc[kb][dt][4 * gg + 0] * p.scale, dk_acc[kb][dt][4 * gg + 1] * p.scale);
                    a[1] = pack2(dk_acc[kb][dt][4 * gg + 2] * p.scale, dk_acc[kb][dt][4 * gg + 3] * p.scale);
                    c[0] = pack2(dv_acc[kb][dt][4 * gg + 0], dv_acc[kb][dt][4 * gg + 1]);
                    c[1] = pack2(dv_acc[kb][dt][4 * gg + 2], dv_acc[kb][dt][4 * gg + 3]);
                    *(u32x2*)(dkp + 32 * dt + 8 * gg + 4 * h) = a;
                    *(u32x2*)(dvp + 32 * dt + 8 * gg + 4 * h) = c;
                }
        }
    }
}

// delta[b,h,s] = sum_d dO[b,s,h,d] * O[b,s,h,d]   (8 lanes per (s,h) row of 64 elements)
__global__ __launch_bounds__(256) void attn_bwd_delta_kernel(const bf16_t* o, const bf16_t* dout, float* delta,
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

extern "C" int vt_attn_bwd_hd64(const void* q, const void* k, const void* v, const void* o, const void* dout,
                                const float* lse2, float* delta_ws, float* dq_f32, void* dk, void* dv,
                                int B, int H, int S,
                                long long q_rs, long long k_rs, long long v_rs, long long o_rs, long long do_rs,
                                long long dq_rs, long long dk_rs, long long dv_rs,
                                long long q_bs, long long k_bs, long long v_bs, long long o_bs, long long do_bs,
                                long long dq_bs, long long dk_bs, long long dv_bs,
                                float softmax_scale, void* stream) {
    if (B <= 0 || H <= 0 || S <= 0) return VT_ERR_BAD_SHAPE;
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
        hipLaunchKernelGGL(attn_bwd_delta_kernel, dim3(blocks), dim3(256), 0, st, (const bf16_t*)o, (const bf16_t*)dout,
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
    hipLaunchKernelGGL(attn_bwd_hd64_kernel, dim3(nkb * H * B), dim3(256), 0, st, p);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}
