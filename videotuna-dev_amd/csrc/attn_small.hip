// Attention against a SHORT key set that fits a workgroup's LDS (<= 128 keys), head_dim 64, forward and backward, gfx950.
// Two users in the VideoCrafter2 UNet (BASELINE configs[3]):
//   * text cross-attention ....... BasicTransformerBlock.attn2 of SpatialTransformer (lvdm/modules/attention.py:101-181, 299-310):
//                                  q = all T*H*W positions of a sample (the 77 text keys are the same for every frame -- the reference
//                                  repeat_interleaves the context over frames, openaimodel3d.py:664), k/v = context[:, :77]
//   * temporal self-attention .... both attentions of TemporalTransformer (attention.py:395-519, only_self_att): one sequence of
//                                  T <= 32 frames per pixel -- B*H*W = 10 240 sequences of 16 tokens at the first level.  mask_block = T:
//                                  consecutive sequences are packed into 32-row tiles and a tile attends to itself under a
//                                  block-diagonal mask, so "tiny-sequence attention" runs on full 32x32 MFMA tiles (the waste is
//                                  FLOPs nobody counts: the kernel is bound by reading q, k, v and writing o).
// The reference uses the einsum -> softmax -> einsum path (attention.py:126-149) or xformers; scores never leave the chip here.
//
// All keys of a (batch item, head) are resident, so the softmax is single-pass (no online rescaling).  Forward computes
// S^T = K Q^T (lane = query, registers = keys: row statistics are lane-local + one cross-half shuffle) and O = P V with P taken
// from the S^T accumulators as the A operand (the MFMA C layout of S^T IS the A layout of P up to a fixed permutation of the
// keys, applied to the V^T image in LDS).  Backward computes S = Q K^T and dP = dO V^T with the KEY on the lane, so that P^T and
// dS^T feed dV = P^T dO and dK = dS^T Q straight from registers; only dS crosses LDS once (per wave) for dQ = dS K.
#include "common.h"

#define AS_KROW 144          // bytes per row of the row-major K / V images (64 bf16 + 16 B pad)
#define AS_MAXK 128
#define AS_TROW(skp) (((skp) + 8) * 2)     // bytes per row of a transposed image [64][skp + 8]

struct AttnSmallParams {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* o; const bf16_t* dout;
    bf16_t* out; float* lse2;
    bf16_t* dq; bf16_t* dk; bf16_t* dv; float* dk32; float* dv32;
    long long q_rs, q_bs, k_rs, k_bs, v_rs, v_bs, o_rs, o_bs, do_rs, do_bs, dq_rs, dq_bs, dk_rs, dk_bs, dv_rs, dv_bs;
    int NB, H, Sq, Sk, mask_block, chunk;
    float scale2;            // softmax_scale * log2(e)
    float scale;
};

__device__ __forceinline__ int as_crow(int r, int hh) { return ((r >> 2) << 3) + (hh << 2) + (r & 3); }     // row of C register r (32x32 MFMA)

__device__ __forceinline__ bf16x8 as_pack8(const f32x16& a, int s2) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)a[8 * s2 + e];
    return o;
}
// B / A operand whose contraction index follows the C-register order: slots e <-> index (2 s2 + e/4) * 8 + 4 hh + e%4 of a row of a
// transposed image: two 8-byte reads
__device__ __forceinline__ bf16x8 as_tfrag(const char* row, int s2, int hh) {
    const u32x2 lo = *(const u32x2*)(row + ((2 * s2) * 8 + 4 * hh) * 2);
    const u32x2 hi = *(const u32x2*)(row + ((2 * s2 + 1) * 8 + 4 * hh) * 2);
    u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
}

// stage K (row-major) and, optionally, V row-major / K^T / V^T of one (item, head) into LDS; keys >= nk are zeros
template <bool WANT_VROW, bool WANT_KT, bool WANT_VT>
__device__ __forceinline__ void as_stage_kv(const AttnSmallParams& p, const bf16_t* kb, const bf16_t* vb, int nk, int skp, char* Ks, char* Vs, char* Kt,
                                            char* Vt) {
    const int tid = threadIdx.x;
    for (int i = tid; i < skp * 8; i += 256) {
        const int key = i >> 3, ch = i & 7;
        u32x4 kk = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
        if (key < nk) {
            kk = *(const u32x4*)(kb + (long long)key * p.k_rs + ch * 8);
            vv = *(const u32x4*)(vb + (long long)key * p.v_rs + ch * 8);
        }
        *(u32x4*)(Ks + key * AS_KROW + ch * 16) = kk;
        if (WANT_VROW) *(u32x4*)(Vs + key * AS_KROW + ch * 16) = vv;
        if (WANT_KT) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                *(unsigned short*)(Kt + (ch * 8 + e) * AS_TROW(skp) + key * 2) = (unsigned short)((kk[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
        }
        if (WANT_VT) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                *(unsigned short*)(Vt + (ch * 8 + e) * AS_TROW(skp) + key * 2) = (unsigned short)((vv[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
        }
    }
}

// ============================================================================================================ forward
template <int NKT, bool MASKED>
__global__ __launch_bounds__(256, 2) void attn_small_fwd_kernel(AttnSmallParams p) {
    constexpr int SKP = NKT * 32;
    constexpr int NACC = MASKED ? 1 : NKT;         // masked: a wave meets ONE key tile, its own
    __shared__ __attribute__((aligned(16))) char Ks[SKP * AS_KROW];
    __shared__ __attribute__((aligned(16))) char Vt[64 * AS_TROW(SKP)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y, item = blockIdx.z;
    constexpr bool masked = MASKED;
    // masked (temporal) mode: item = 128 consecutive rows of ONE [R, .] row space (q == k row space), keys = those rows
    const long long qbase = masked ? (long long)item * 128 * p.q_rs : (long long)item * p.q_bs + (long long)blockIdx.x * 128 * p.q_rs;
    const int q0 = masked ? 0 : blockIdx.x * 128;                    // first query of this workgroup inside the item
    const int nq = masked ? min(128, p.Sq - item * 128) : min(128, p.Sq - q0);
    const int nk = masked ? nq : p.Sk;
    const bf16_t* kb = p.k + (masked ? (long long)item * 128 * p.k_rs : (long long)item * p.k_bs) + h * 64;
    const bf16_t* vb = p.v + (masked ? (long long)item * 128 * p.v_rs : (long long)item * p.v_bs) + h * 64;
    as_stage_kv<false, false, true>(p, kb, vb, nk, SKP, Ks, nullptr, nullptr, Vt);
    __syncthreads();
    const int ql = lane & 31, hh = lane >> 5;
    const int qi = wave * 32 + ql;                                   // query row inside the workgroup's 128
    const bool qok = qi < nq;
    const bf16_t* qp = p.q + qbase + (long long)qi * p.q_rs + h * 64;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        u32x4 t = {0u, 0u, 0u, 0u};
        if (qok) t = *(const u32x4*)(qp + ks * 16 + hh * 8);
        qf[ks] = __builtin_bit_cast(bf16x8, t);
    }
    f32x16 s[NACC];
    float mx = -1e30f;
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
        const int kt = MASKED ? wave : a;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[a][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 kf = *(const bf16x8*)(Ks + (kt * 32 + ql) * AS_KROW + ks * 32 + hh * 16);
            s[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[a], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kl = as_crow(r, hh);
            bool ok = kt * 32 + kl < nk;
            if (MASKED) ok = ok && (kl / p.mask_block == ql / p.mask_block);
            s[a][r] = ok ? s[a][r] * p.scale2 : -1e30f;
            mx = fmaxf(mx, s[a][r]);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = s[a][r] > -1e29f ? __builtin_amdgcn_exp2f(s[a][r] - mx) : 0.f;
            s[a][r] = e;
            sum += e;
        }
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    if (hh == 0 && qok && p.lse2 != nullptr) {
        const long long row = masked ? (long long)item * 128 + qi : (long long)q0 + qi;
        p.lse2[((long long)(masked ? 0 : item) * p.H + h) * p.Sq + row] = mx + __builtin_amdgcn_logf(sum);       // v_log_f32 = log2
    }
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
        const int kt = MASKED ? wave : a;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[a][e] *= inv;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pf = as_pack8(s[a], s2);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const bf16x8 vf = as_tfrag(Vt + (dt * 32 + ql) * AS_TROW(SKP) + kt * 64, s2, hh);
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, vf, o[dt], 0, 0, 0);
            }
        }
    }
    bf16_t* ob = p.out + (masked ? (long long)item * 128 * p.o_rs : (long long)item * p.o_bs + (long long)q0 * p.o_rs) + h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qr = wave * 32 + as_crow(r, hh);
            if (qr < nq) ob[(long long)qr * p.o_rs + dt * 32 + ql] = (bf16_t)o[dt][r];
        }
}

// ============================================================================================================ backward
template <int NKT, bool MASKED>
__global__ __launch_bounds__(256, 1) void attn_small_bwd_kernel(AttnSmallParams p) {
    constexpr int SKP = NKT * 32;
    constexpr int NACC = MASKED ? 1 : NKT;         // masked: a wave meets ONE key tile, its own (one query tile per wave)
    constexpr int TR = AS_TROW(SKP);
    __shared__ __attribute__((aligned(16))) char Ks[SKP * AS_KROW];
    __shared__ __attribute__((aligned(16))) char Vs[SKP * AS_KROW];
    __shared__ __attribute__((aligned(16))) char Kt[64 * TR];
    __shared__ __attribute__((aligned(16))) char Wv[4][64 * 80 * 2 + 32 * TR + 256];      // per wave: Q^T | dO^T | dS | lse, delta
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y, item = blockIdx.z;
    constexpr bool masked = MASKED;
    const int c0 = masked ? 0 : blockIdx.x * p.chunk;                 // first query of this workgroup's chunk inside the item
    const int nq = masked ? min(128, p.Sq - item * 128) : min(p.chunk, p.Sq - c0);
    const int nk = masked ? nq : p.Sk;
    const long long ibq = masked ? (long long)item * 128 : 0;         // row offset of a masked item inside the shared row space
    const bf16_t* kb = p.k + (masked ? ibq * p.k_rs : (long long)item * p.k_bs) + h * 64;
    const bf16_t* vb = p.v + (masked ? ibq * p.v_rs : (long long)item * p.v_bs) + h * 64;
    as_stage_kv<true, true, false>(p, kb, vb, nk, SKP, Ks, Vs, Kt, nullptr);
    __syncthreads();
    const int ql = lane & 31, hh = lane >> 5;
    char* Qt = Wv[wave];
    char* dOt = Qt + 64 * 80;
    char* dSs = dOt + 64 * 80;
    float* stat = (float*)(dSs + 32 * TR);            // [0,32): lse2, [32,64): delta
    const long long qoff = masked ? ibq * p.q_rs : (long long)item * p.q_bs + (long long)c0 * p.q_rs;
    const long long ooff = masked ? ibq * p.o_rs : (long long)item * p.o_bs + (long long)c0 * p.o_rs;
    const long long dooff = masked ? ibq * p.do_rs : (long long)item * p.do_bs + (long long)c0 * p.do_rs;
    const long long dqoff = masked ? ibq * p.dq_rs : (long long)item * p.dq_bs + (long long)c0 * p.dq_rs;
    const long long lse_off = ((long long)(masked ? 0 : item) * p.H + h) * p.Sq + (masked ? ibq : c0);

    f32x16 dK[NACC][2], dV[NACC][2];
#pragma unroll
    for (int kt = 0; kt < NACC; ++kt)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { dK[kt][dt][e] = 0.f; dV[kt][dt][e] = 0.f; }

    const int ntile = (nq + 31) / 32;
    for (int qt = wave; qt < ntile; qt += 4) {
        const int qi = qt * 32 + ql;
        const bool qok = qi < nq;
        bf16x8 qf[4], dof[4];
        float dl = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            u32x4 tq = {0u, 0u, 0u, 0u}, td = {0u, 0u, 0u, 0u}, to = {0u, 0u, 0u, 0u};
            if (qok) {
                tq = *(const u32x4*)(p.q + qoff + (long long)qi * p.q_rs + h * 64 + ks * 16 + hh * 8);
                td = *(const u32x4*)(p.dout + dooff + (long long)qi * p.do_rs + h * 64 + ks * 16 + hh * 8);
                to = *(const u32x4*)(p.o + ooff + (long long)qi * p.o_rs + h * 64 + ks * 16 + hh * 8);
            }
            qf[ks] = __builtin_bit_cast(bf16x8, tq);
            dof[ks] = __builtin_bit_cast(bf16x8, td);
            float a[8], b[8];
            unpack8(td, a); unpack8(to, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                dl += a[e] * b[e];
                const int d = ks * 16 + hh * 8 + e;
                *(unsigned short*)(Qt + d * 80 + ql * 2) = (unsigned short)((tq[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
                *(unsigned short*)(dOt + d * 80 + ql * 2) = (unsigned short)((td[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
            }
        }
        dl += __shfl_xor(dl, 32, 64);
        if (hh == 0) {
            stat[ql] = qok ? p.lse2[lse_off + qi] : 0.f;
            stat[32 + ql] = dl;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            const int kt = MASKED ? (qt & 3) : a;
            f32x16 S, dP;
#pragma unroll
            for (int e = 0; e < 16; ++e) { S[e] = 0.f; dP[e] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(Ks + (kt * 32 + ql) * AS_KROW + ks * 32 + hh * 16);
                const bf16x8 vf = *(const bf16x8*)(Vs + (kt * 32 + ql) * AS_KROW + ks * 32 + hh * 16);
                S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[ks], kf, S, 0, 0, 0);          // [query rows][key cols]
                dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof[ks], vf, dP, 0, 0, 0);
            }
            const int key = kt * 32 + ql;
            const bool kok = key < nk;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qr = as_crow(r, hh);
                bool ok = kok && (qt * 32 + qr) < nq;
                if (MASKED) ok = ok && (ql / p.mask_block == qr / p.mask_block);
                const float pr = ok ? __builtin_amdgcn_exp2f(S[r] * p.scale2 - stat[qr]) : 0.f;
                S[r] = pr;
                dP[r] = pr * (dP[r] - stat[32 + qr]) * p.scale;
                *(unsigned short*)(dSs + qr * TR + key * 2) = __builtin_bit_cast(unsigned short, (bf16_t)dP[r]);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = as_pack8(S, s2), dsf = as_pack8(dP, s2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const bf16x8 dob = as_tfrag(dOt + (dt * 32 + ql) * 80, s2, hh);
                    const bf16x8 qb = as_tfrag(Qt + (dt * 32 + ql) * 80, s2, hh);
                    dV[a][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, dob, dV[a][dt], 0, 0, 0);    // [key rows][d cols]
                    dK[a][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsf, qb, dK[a][dt], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // dQ[32 q][64 d] = dS K
        f32x16 dQ[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) dQ[dt][e] = 0.f;
        // masked: the query tile met only its own key tile -> two 16-key steps starting there
        const int s_lo = MASKED ? (qt & 3) * 2 : 0;
#pragma unroll
        for (int si = 0; si < (MASKED ? 2 : SKP / 16); ++si) {
            const int s = s_lo + si;
            const bf16x8 af = *(const bf16x8*)(dSs + ql * TR + (s * 16 + hh * 8) * 2);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const bf16x8 bfk = *(const bf16x8*)(Kt + (dt * 32 + ql) * TR + (s * 16 + hh * 8) * 2);
                dQ[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfk, dQ[dt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qr = qt * 32 + as_crow(r, hh);
                if (qr < nq) p.dq[dqoff + (long long)qr * p.dq_rs + h * 64 + dt * 32 + ql] = (bf16_t)dQ[dt][r];
            }
        __builtin_amdgcn_wave_barrier();
    }
    // ---- dK / dV: direct bf16 stores when the keys belong to this workgroup alone (masked mode), fp32 atomics otherwise ----
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
        const int kt = MASKED ? wave : a;           // masked: wave w ran query tile w against key tile w
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + as_crow(r, hh);
                if (key >= nk) continue;
                const int col = h * 64 + dt * 32 + ql;
                if (MASKED) {
                    p.dk[(ibq + key) * p.dk_rs + col] = (bf16_t)dK[a][dt][r];
                    p.dv[(ibq + key) * p.dv_rs + col] = (bf16_t)dV[a][dt][r];
                } else {
                    atomicAdd(p.dk32 + ((long long)item * p.Sk + key) * p.dk_rs + col, dK[a][dt][r]);
                    atomicAdd(p.dv32 + ((long long)item * p.Sk + key) * p.dv_rs + col, dV[a][dt][r]);
                }
            }
    }
}

static int as_check(const void* q, const void* k, const void* v, int NB, int H, int Sq, int Sk, int mask_block, const long long* strides, int nstr) {
    if (NB <= 0 || H <= 0 || Sq <= 0 || Sk <= 0 || Sk > AS_MAXK) return VT_ERR_BAD_SHAPE;
    if (mask_block < 0 || (mask_block > 0 && (32 % mask_block))) return VT_ERR_BAD_SHAPE;
    for (int i = 0; i < nstr; ++i)
        if (strides[i] % 8) return VT_ERR_BAD_SHAPE;
    if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v)) & 15) return VT_ERR_BAD_ALIGN;
    return VT_OK;
}

// Element (item b, row s, head h, d) of q / o lives at base + b*bs + s*rs + h*64 + d (same for k / v with their strides).
// mask_block == 0 ("cross"): NB items, Sq queries and Sk <= 128 keys each.  mask_block = T > 0 ("packed self-attention", 32 % T == 0):
// q, k, v, o are [Sq, .] row spaces holding Sq / T consecutive sequences of T rows (the batch strides are ignored, NB must be 1, Sk = Sq);
// a row attends to the rows of its own sequence.  lse2 (fp32 [NB, H, Sq], log2 domain of the scaled scores) is kept for the backward.
extern "C" int vt_attn_small_fwd(const void* q, const void* k, const void* v, void* o, float* lse2, int NB, int H, int Sq, int Sk,
                                 long long q_rs, long long q_bs, long long k_rs, long long k_bs, long long v_rs, long long v_bs,
                                 long long o_rs, long long o_bs, float softmax_scale, int mask_block, void* stream) {
    const long long st[8] = {q_rs, q_bs, k_rs, k_bs, v_rs, v_bs, o_rs, o_bs};
    int rc = as_check(q, k, v, NB, H, Sq, mask_block ? 1 : Sk, mask_block, st, 8);
    if (rc != VT_OK) return rc;
    if (mask_block > 0 && (NB != 1 || Sk != Sq)) return VT_ERR_BAD_SHAPE;
    AttnSmallParams p = {};
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.out = (bf16_t*)o; p.lse2 = lse2;
    p.q_rs = q_rs; p.q_bs = q_bs; p.k_rs = k_rs; p.k_bs = k_bs; p.v_rs = v_rs; p.v_bs = v_bs; p.o_rs = o_rs; p.o_bs = o_bs;
    p.NB = NB; p.H = H; p.Sq = Sq; p.Sk = Sk; p.mask_block = mask_block;
    p.scale = softmax_scale; p.scale2 = softmax_scale * 1.4426950408889634f;
    hipStream_t s = (hipStream_t)stream;
    if (mask_block > 0) {
        hipLaunchKernelGGL((attn_small_fwd_kernel<4, true>), dim3(1, H, (Sq + 127) / 128), dim3(256), 0, s, p);
    } else {
        const dim3 grid((Sq + 127) / 128, H, NB);
        const int nkt = (Sk + 31) / 32;
        if (nkt == 1) hipLaunchKernelGGL((attn_small_fwd_kernel<1, false>), grid, dim3(256), 0, s, p);
        else if (nkt == 2) hipLaunchKernelGGL((attn_small_fwd_kernel<2, false>), grid, dim3(256), 0, s, p);
        else if (nkt == 3) hipLaunchKernelGGL((attn_small_fwd_kernel<3, false>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((attn_small_fwd_kernel<4, false>), grid, dim3(256), 0, s, p);
    }
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}

// dq: bf16, layout like q.  mask_block > 0: dk, dv bf16 in the k / v row space (dk32 / dv32 unused).  mask_block == 0: dk32, dv32 fp32
// [NB, Sk, dk_rs] accumulators that the CALLER ZEROES (query chunks add into them with atomics; dk / dv unused).
extern "C" int vt_attn_small_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse2,
                                 void* dq, void* dk, void* dv, float* dk32, float* dv32, int NB, int H, int Sq, int Sk,
                                 long long q_rs, long long q_bs, long long k_rs, long long k_bs, long long v_rs, long long v_bs,
                                 long long o_rs, long long o_bs, long long do_rs, long long do_bs, long long dq_rs, long long dq_bs,
                                 long long dk_rs, long long dv_rs, float softmax_scale, int mask_block, void* stream) {
    const long long st[12] = {q_rs, q_bs, k_rs, k_bs, v_rs, v_bs, o_rs, o_bs, do_rs, do_bs, dq_rs, dq_bs};
    int rc = as_check(q, k, v, NB, H, Sq, mask_block ? 1 : Sk, mask_block, st, 12);
    if (rc != VT_OK) return rc;
    if (mask_block > 0 && (NB != 1 || Sk != Sq || dk == nullptr || dv == nullptr)) return VT_ERR_BAD_SHAPE;
    if (mask_block == 0 && (dk32 == nullptr || dv32 == nullptr)) return VT_ERR_BAD_SHAPE;
    if ((((uintptr_t)o) | ((uintptr_t)dout)) & 15) return VT_ERR_BAD_ALIGN;
    AttnSmallParams p = {};
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (const bf16_t*)o; p.dout = (const bf16_t*)dout;
    p.lse2 = const_cast<float*>(lse2);
    p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv; p.dk32 = dk32; p.dv32 = dv32;
    p.q_rs = q_rs; p.q_bs = q_bs; p.k_rs = k_rs; p.k_bs = k_bs; p.v_rs = v_rs; p.v_bs = v_bs; p.o_rs = o_rs; p.o_bs = o_bs;
    p.do_rs = do_rs; p.do_bs = do_bs; p.dq_rs = dq_rs; p.dq_bs = dq_bs; p.dk_rs = dk_rs; p.dv_rs = dv_rs;
    p.NB = NB; p.H = H; p.Sq = Sq; p.Sk = Sk; p.mask_block = mask_block;
    p.scale = softmax_scale; p.scale2 = softmax_scale * 1.4426950408889634f;
    hipStream_t s = (hipStream_t)stream;
    if (mask_block > 0) {
        p.chunk = 128;
        hipLaunchKernelGGL((attn_small_bwd_kernel<4, true>), dim3(1, H, (Sq + 127) / 128), dim3(256), 0, s, p);
    } else {
        p.chunk = 1024;
        const dim3 grid((Sq + p.chunk - 1) / p.chunk, H, NB);
        const int nkt = (Sk + 31) / 32;
        if (nkt == 1) hipLaunchKernelGGL((attn_small_bwd_kernel<1, false>), grid, dim3(256), 0, s, p);
        else if (nkt == 2) hipLaunchKernelGGL((attn_small_bwd_kernel<2, false>), grid, dim3(256), 0, s, p);
        else if (nkt == 3) hipLaunchKernelGGL((attn_small_bwd_kernel<3, false>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((attn_small_bwd_kernel<4, false>), grid, dim3(256), 0, s, p);
    }
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}
