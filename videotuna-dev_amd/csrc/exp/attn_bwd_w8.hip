// EXPERIMENT: flash-attention backward with EIGHT waves per workgroup (two per SIMD, <= 256 registers each).
//
// Same math, LDS images, persistent schedule and output contract as csrc/attn_bwd.hip (autograd of the SDPA call reached
// through videotuna/models/cogvideo_hf/cogvideo_pl.py:865-871); pure atomics, no hand-off chains.  Why: PMC shows the
// shipped 4-wave kernel (one wave per SIMD, all 512 registers) with MFMA busy 36 %, VALU busy 38 % and both at once only
// 13 % of the time -- nothing else is resident on a SIMD to run under an MFMA or an exp2.  Here a workgroup still owns 256
// keys, but wave w owns only keys [32w, 32w+32): half the accumulators, half the softmax work per wave, and a second wave
// on every SIMD to overlap with.  The dQ phase (dS image x K image, 4 tiles of 32x32 over 256 keys) is done by waves 0..3
// exactly as in the shipped kernel while waves 4..7 already start the next step (double-buffered dS image).
#include "../common.h"

#ifndef VT_SUFFIX
#define VT_SUFFIX _w8
#endif
#define VT_CAT_(a, b) a##b
#define VT_CAT(a, b) VT_CAT_(a, b)
#define W8_KERNEL VT_CAT(attn_bwd_w8_kernel, VT_SUFFIX)
#define W8_DELTA VT_CAT(attn_bwd_w8_delta_kernel, VT_SUFFIX)
#define W8_BODY VT_CAT(w8_body, VT_SUFFIX)

struct AttnBwdParamsW8 {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* dout;
    const float* lse2; const float* delta; float* dq; bf16_t* dk; bf16_t* dv;
    int S, H, B;
    long long q_rs, k_rs, v_rs, do_rs, dq_rs, dk_rs, dv_rs;
    long long q_bs, k_bs, v_bs, do_bs, dq_bs, dk_bs, dv_bs;
    float scale, scale_log2;
};

#define KIMG 0
#define DSIMG 32768
#define QTILE 98304
#define LSEOFF 131072
#define W8_LDS 132096

typedef __attribute__((ext_vector_type(8))) short short8w;

static __device__ __forceinline__ int w8_swz_f(int row) { return (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 2) & 1); }
static __device__ __forceinline__ int w8_swz_off(int row, int chunk) { return row * 128 + ((chunk ^ w8_swz_f(row)) << 4); }
static __device__ __forceinline__ bf16x8 w8_tr_pair(const char* p0, const char* p1) {
    short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(p0));
    short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(p1));
    short8w v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v8);
}

template <bool RAGGED, bool PRESCALED>
__device__ __forceinline__ void W8_BODY(const AttnBwdParamsW8& p, char* smem, const int id) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);          // 0..7
    const int r = lane & 31, h = lane >> 5, g = lane >> 4, ql = (lane & 15) >> 2, pl = lane & 3;
    const int nkb = (p.S + 255) / 256;
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
    __amdgpu_buffer_rsrc_t rdq = make_rsrc(p.dq + (size_t)b * p.dq_bs + head * 64, (unsigned)((long long)(p.S - 1) * p.dq_rs * 4 + 256));
    const float* lse_b = p.lse2 + (size_t)bh * p.S;
    const float* dl_b = p.delta + (size_t)bh * p.S;

    // ---- K block image (B operand of dQ): 256 keys x 8 chunks, 4 per thread ----
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = tid + 512 * j;
        const int key = i >> 3, c = i & 7;
        u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, (int)((key0 + key) * p.k_rs * 2) + c * 16, 0, 0));
        *(u32x4*)(smem + KIMG + w8_swz_off(key, c)) = v;
    }
    // ---- K / V fragments of this wave's 32 keys, resident for the whole key block ----
    bf16x8 kf[4], vf[4];
    {
        const int key = key0 + 32 * w + r;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rk, (int)(key * p.k_rs * 2) + (16 * s + 8 * h) * 2, 0, 0));
            vf[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rv, (int)(key * p.v_rs * 2) + (16 * s + 8 * h) * 2, 0, 0));
        }
    }
    const float kmask = (RAGGED && (key0 + 32 * w + r) >= p.S) ? -1.0e30f : 0.f;

    // ---- per-lane LDS offsets (identical to csrc/attn_bwd.hip) ----
    int rowrd[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) rowrd[s] = r * 128 + (((2 * s + h) ^ w8_swz_f(r)) << 4);
    int trA[2][2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int sec = 0; sec < 2; ++sec) {
            const int fx = ((ql >> 1) << 2) | (sec << 1) | h;
            trA[dt][sec] = (4 * h + ql + 8 * sec) * 128 + (((4 * dt + 2 * (g & 1) + (pl >> 1)) ^ fx) << 4) + (pl & 1) * 8;
        }
    const int qs_w = w & 1, dt_w = (w >> 1) & 1;          // dQ phase (waves 0..3): (q-half, d-half) of the 64x64 tile
    int trQA[2], trQB[2];
#pragma unroll
    for (int sec = 0; sec < 2; ++sec) {
        const int fx = ((ql >> 1) << 2) | (h << 1) | sec;
        const int row = (8 * h + ql + 4 * sec) * 128;
        trQA[sec] = row + (((4 * qs_w + 2 * (g & 1) + (pl >> 1)) ^ fx) << 4) + (pl & 1) * 8;
        trQB[sec] = row + (((4 * dt_w + 2 * (g & 1) + (pl >> 1)) ^ fx) << 4) + (pl & 1) * 8;
    }
    const int fr = w8_swz_f(r);
    const int dq_voff = (int)((32 * qs_w + 4 * h) * p.dq_rs * 4) + (32 * dt_w + r) * 4;
    const int dq_rowb = (int)(p.dq_rs * 4);

    // ---- staging of the Q / dO tiles (64 rows x 8 chunks each): one chunk of each per thread ----
    const int st_voq = (int)((tid >> 3) * p.q_rs * 2) + (tid & 7) * 16;
    const int st_vodo = (int)((tid >> 3) * p.do_rs * 2) + (tid & 7) * 16;
    const int st_lds = w8_swz_off(tid >> 3, tid & 7);
    const int stat_i = tid & 63;
    const bool stat_is_lse = (tid & 64) == 0;
    const float* stat_src = stat_is_lse ? lse_b : dl_b;
    const float stat_mul = stat_is_lse ? (PRESCALED ? -1.0f : -1.0f / p.scale_log2) : -1.0f;
    const int stat_lds = LSEOFF + (tid & 127) * 4;
    u32x4 gq, gdo;
    float gstat = 0.f;
    auto gload = [&](int t) {
        const int q0 = t * 64;
        const int sq = (int)((long long)q0 * p.q_rs * 2), sdo = (int)((long long)q0 * p.do_rs * 2);
        gq = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rq, st_voq, sq, 0));
        gdo = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdo, st_vodo, sdo, 0));
        int qi = q0 + stat_i;
        const bool ok = qi < p.S;
        qi = ok ? qi : p.S - 1;
        const float v = stat_src[qi] * stat_mul;
        gstat = ok ? v : 0.f;
    };
    auto lstore = [&](int buf) {
        char* base = smem + QTILE + buf * 16384;
        *(u32x4*)(base + st_lds) = gq;
        *(u32x4*)(base + 8192 + st_lds) = gdo;
        *(float*)(smem + stat_lds + buf * 512) = gstat;     // threads t, t+128, t+256, t+384 write the same value
    };

    f32x16 dk_acc[2], dv_acc[2];          // [dt]: dK^T / dV^T of this wave's 32 keys
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dk_acc[c][i] = 0.f; dv_acc[c][i] = 0.f; }

    const float sc = p.scale_log2;
    const int nsteps = (p.S + 63) / 64;
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

#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            // S'' and dP' accumulators start from the row constants (-lse2/c [+ key mask], -delta)
            f32x16 sacc, pacc;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const f32x4 a = *(const f32x4*)(lsel + 32 * qs + 8 * gg + 4 * h);
                const f32x4 c = *(const f32x4*)(lsel + 64 + 32 * qs + 8 * gg + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) { sacc[4 * gg + e] = RAGGED ? a[e] + kmask : a[e]; pacc[4 * gg + e] = c[e]; }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 qa = *(const bf16x8*)(qimg + qs * 4096 + rowrd[s]);
                const bf16x8 doa = *(const bf16x8*)(doimg + qs * 4096 + rowrd[s]);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[s], sacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(doa, vf[s], pacc, 0, 0, 0);
            }
            // P = exp2(c * S''), dS = P * dP'; both packed to bf16 pairs (B operands + dS image)
            unsigned pw[8], dw[8];
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
                const int ro = (32 * qs + 16 * s2) * 128;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const bf16x8 doT = w8_tr_pair(doimg + ro + trA[dt][0], doimg + ro + trA[dt][1]);
                    const bf16x8 qT = w8_tr_pair(qimg + ro + trA[dt][0], qimg + ro + trA[dt][1]);
                    dv_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(doT, pb, dv_acc[dt], 0, 0, 0);
                    dk_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qT, dsb, dk_acc[dt], 0, 0, 0);
                }
            }
            // dS image: row = key (32w + r), 8 bytes = q 32qs + 8g' + 4h + (0..3)
            char* drow = dsimg + (32 * w + r) * 128 + 8 * h;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const u32x2 two = {dw[2 * gg], dw[2 * gg + 1]};
                *(u32x2*)(drow + (((4 * qs + gg) ^ fr) << 4)) = two;
            }
        }
        lstore(buf ^ 1);
        __syncthreads();

        // ---- dQ tile (32 q x 32 d) over all 256 keys: waves 0..3 only; waves 4..7 go on with the next step ----
        if (w < 4) {
            f32x16 dq_acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) dq_acc[i] = 0.f;
#pragma unroll
            for (int s3 = 0; s3 < 16; ++s3) {
                const bf16x8 fa = w8_tr_pair(dsimg + s3 * 2048 + trQA[0], dsimg + s3 * 2048 + trQA[1]);
                const bf16x8 fb = w8_tr_pair(smem + KIMG + s3 * 2048 + trQB[0], smem + KIMG + s3 * 2048 + trQB[1]);
                dq_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, dq_acc, 0, 0, 0);
            }
            const int soff = (int)((long long)t * 64 * p.dq_rs * 4);
#ifdef W8_NOATOMICS       // timing-only ablation (results wrong)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(dq_acc[i]));
            (void)soff;
#else
#pragma unroll
            for (int i = 0; i < 16; ++i)
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(dq_acc[i] * p.scale, rdq, dq_voff,
                                                                soff + ((i & 3) + 8 * (i >> 2)) * dq_rowb, 0);
#endif
        }
    }

    // ---- epilogue: dK^T, dV^T accumulators -> dk[key][d], dv[key][d] ----
    const float dk_mul = PRESCALED ? 0.6931471805599453f : p.scale;
    {
        const int key = key0 + 32 * w + r;
        if (key < p.S) {
            bf16_t* dkp = p.dk + (size_t)b * p.dk_bs + (size_t)key * p.dk_rs + head * 64;
            bf16_t* dvp = p.dv + (size_t)b * p.dv_bs + (size_t)key * p.dv_rs + head * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    u32x2 a, c;
                    a[0] = pack2(dk_acc[dt][4 * gg + 0] * dk_mul, dk_acc[dt][4 * gg + 1] * dk_mul);
                    a[1] = pack2(dk_acc[dt][4 * gg + 2] * dk_mul, dk_acc[dt][4 * gg + 3] * dk_mul);
                    c[0] = pack2(dv_acc[dt][4 * gg + 0], dv_acc[dt][4 * gg + 1]);
                    c[1] = pack2(dv_acc[dt][4 * gg + 2], dv_acc[dt][4 * gg + 3]);
                    *(u32x2*)(dkp + 32 * dt + 8 * gg + 4 * h) = a;
                    *(u32x2*)(dvp + 32 * dt + 8 * gg + 4 * h) = c;
                }
        }
    }
}

template <bool PRESCALED>
__global__ __launch_bounds__(512, 1) void W8_KERNEL(AttnBwdParamsW8 p) {
    __shared__ __attribute__((aligned(16))) char smem[W8_LDS];
    const int nkb = (p.S + 255) / 256;
    const int nitems = nkb * p.H * p.B;
    const int spx = gridDim.x >> 3;
    const int slot = (int)(blockIdx.x & 7) * spx + (int)(blockIdx.x >> 3);
    for (int item = slot; item < nitems; item += gridDim.x) {
        const int kblk = item % nkb;
        if ((kblk + 1) * 256 > p.S) W8_BODY<true, PRESCALED>(p, smem, item);
        else W8_BODY<false, PRESCALED>(p, smem, item);
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void W8_DELTA(const bf16_t* o, const bf16_t* dout, float* delta, int B, int H, int S,
                                                                long long o_rs, long long do_rs, long long o_bs, long long do_bs) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long row = gid >> 3;
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

// same signature as vt_attn_bwd_hd64 (the chain workspace is accepted and ignored) so tools/kbench_variants.py can A/B it
extern "C" int VT_CAT(vt_attn_bwd_hd64, VT_SUFFIX)(const void* q, const void* k, const void* v, const void* o, const void* dout,
                                const float* lse2, float* delta_ws, float* dq_f32, void* dk, void* dv,
                                int B, int H, int S,
                                long long q_rs, long long k_rs, long long v_rs, long long o_rs, long long do_rs,
                                long long dq_rs, long long dk_rs, long long dv_rs,
                                long long q_bs, long long k_bs, long long v_bs, long long o_bs, long long do_bs,
                                long long dq_bs, long long dk_bs, long long dv_bs,
                                float softmax_scale, int q_prescaled, void* chain_ws, long long chain_ws_bytes, void* stream) {
    if (B <= 0 || H <= 0 || S <= 0) return VT_ERR_BAD_SHAPE;
    if ((long long)S * dq_rs * 4 >= 0x7fffffffLL) return VT_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    {
        const long long total = (long long)B * S * H * 8;
        const int blocks = (int)((total + 255) / 256);
        hipLaunchKernelGGL(W8_DELTA, dim3(blocks), dim3(256), 0, st, (const bf16_t*)o, (const bf16_t*)dout,
                           delta_ws, B, H, S, o_rs, do_rs, o_bs, do_bs);
    }
    AttnBwdParamsW8 p;
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.dout = (const bf16_t*)dout;
    p.lse2 = lse2; p.delta = delta_ws; p.dq = dq_f32; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv;
    p.S = S; p.H = H; p.B = B;
    p.q_rs = q_rs; p.k_rs = k_rs; p.v_rs = v_rs; p.do_rs = do_rs; p.dq_rs = dq_rs; p.dk_rs = dk_rs; p.dv_rs = dv_rs;
    p.q_bs = q_bs; p.k_bs = k_bs; p.v_bs = v_bs; p.do_bs = do_bs; p.dq_bs = dq_bs; p.dk_bs = dk_bs; p.dv_bs = dv_bs;
    p.scale = softmax_scale; p.scale_log2 = softmax_scale * 1.4426950408889634f;
    const long long nwg = (long long)((S + 255) / 256) * H * B;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
    const int slots = cus >= 8 ? cus / 8 * 8 : 8;
    const long long grid = nwg < slots ? (nwg + 7) / 8 * 8 : slots;
    if (q_prescaled) hipLaunchKernelGGL(W8_KERNEL<true>, dim3((unsigned)grid), dim3(512), 0, st, p);
    else hipLaunchKernelGGL(W8_KERNEL<false>, dim3((unsigned)grid), dim3(512), 0, st, p);
    (void)chain_ws; (void)chain_ws_bytes;
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}
extern "C" int VT_CAT(vt_attn_bwd_set_chain, VT_SUFFIX)(int, int) { return VT_OK; }
extern "C" long long VT_CAT(vt_attn_bwd_chain_ws_bytes, VT_SUFFIX)(int, int, int) { return 0; }
