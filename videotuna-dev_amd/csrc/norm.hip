// Fused normalisation kernels (HBM-bound; one read + one write per element), gfx950.
//
//  * ln_modulate fwd/bwd : LayerNorm(affine) followed by adaLN modulation y = LN(x)*(1+scale)+shift with
//    separate (shift, scale) for the text rows and the video rows of each sample -- diffusers'
//    CogVideoXLayerNormZero / AdaLayerNorm / norm_final, reached from cogvideo_pl.py:865-871
//    (in-tree twin videotuna/models/cogvideo_sat/dit_video_concat.py:430-431,577-645).
//  * qk_layernorm fwd/bwd: LayerNorm(64, eps 1e-6, affine) on every head of q and k (SURVEY a4;
//    twin dit_video_concat.py:554-575,686-690).
// One wave per token row for the model-dim LayerNorm (row kept in registers, two-pass statistics in
// fp32, 16-byte vector loads/stores); 8 lanes per (token, head) for the head-dim LayerNorm.
#include "common.h"

#define LN_MAXCH 8      // up to 8 chunks of 8 elements per lane -> D <= 4096

struct LnParams {
    const bf16_t* x; int ldx;
    bf16_t* y; int ldy;
    const bf16_t* gamma; const bf16_t* beta;
    const float* shift_txt; const float* scale_txt; const float* shift_vid; const float* scale_vid;
    int mod_bstride;
    float* mean; float* rstd;
    int M, D, S, St;
    float eps;
};

template <int NCH>
__global__ __launch_bounds__(256) void ln_modulate_fwd_kernel(LnParams p) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= p.M) return;
    const int nchunks = p.D >> 3;
    const bf16_t* xr = p.x + (size_t)m * p.ldx;
    float v[NCH][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunks) {
            u32x4 raw = *(const u32x4*)(xr + c * 8);
            unpack8(raw, v[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[i][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
    }
    const float mean = wave_sum(sum) / (float)p.D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { float d = v[i][j] - mean; sq += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)p.D + p.eps);
    if (lane == 0 && p.mean != nullptr) { p.mean[m] = mean; p.rstd[m] = rstd; }
    const float* sh = nullptr; const float* scp = nullptr;
    if (p.shift_vid != nullptr) {
        const int b = m / p.S;
        const int s = m - b * p.S;
        sh = (s < p.St ? p.shift_txt : p.shift_vid) + (size_t)b * p.mod_bstride;
        scp = (s < p.St ? p.scale_txt : p.scale_vid) + (size_t)b * p.mod_bstride;
    }
    bf16_t* yr = p.y + (size_t)m * p.ldy;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunks) {
            float o[8];
            float ga[8], be[8];
            if (p.gamma != nullptr) {
                unpack8(*(const u32x4*)(p.gamma + c * 8), ga);
                unpack8(*(const u32x4*)(p.beta + c * 8), be);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = (v[i][j] - mean) * rstd;
                if (p.gamma != nullptr) t = t * ga[j] + be[j];
                o[j] = t;
            }
            if (sh != nullptr) {
                f32x4 s0 = *(const f32x4*)(scp + c * 8), s1 = *(const f32x4*)(scp + c * 8 + 4);
                f32x4 h0 = *(const f32x4*)(sh + c * 8), h1 = *(const f32x4*)(sh + c * 8 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = o[j] * (1.f + s0[j]) + h0[j];
                    o[j + 4] = o[j + 4] * (1.f + s1[j]) + h1[j];
                }
            }
            *(u32x4*)(yr + c * 8) = pack8(o);
        }
    }
}

struct LnBwdParams {
    const bf16_t* dy; int lddy;       // grad wrt the modulated output
    const bf16_t* x; int ldx;         // LN input
    const float* mean; const float* rstd;
    const bf16_t* gamma;              // or null
    const float* scale_txt; const float* scale_vid; int mod_bstride;   // or null
    const bf16_t* dres; int lddres;   // residual-stream gradient to add (or null)
    bf16_t* dx; int lddx;
    int M, D, S, St;
};

template <int NCH>
__global__ __launch_bounds__(256) void ln_modulate_bwd_kernel(LnBwdParams p) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= p.M) return;
    const int nchunks = p.D >> 3;
    const float mean = p.mean[m], rstd = p.rstd[m];
    const float* scp = nullptr;
    if (p.scale_vid != nullptr) {
        const int b = m / p.S;
        const int s = m - b * p.S;
        scp = (s < p.St ? p.scale_txt : p.scale_vid) + (size_t)b * p.mod_bstride;
    }
    float gv[NCH][8], xh[NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunks) {
            float dyv[8], xv[8];
            unpack8(*(const u32x4*)(p.dy + (size_t)m * p.lddy + c * 8), dyv);
            unpack8(*(const u32x4*)(p.x + (size_t)m * p.ldx + c * 8), xv);
            float ga[8];
            if (p.gamma != nullptr) unpack8(*(const u32x4*)(p.gamma + c * 8), ga);
            float sc8[8];
            if (scp != nullptr) {
                f32x4 a = *(const f32x4*)(scp + c * 8), bq = *(const f32x4*)(scp + c * 8 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { sc8[j] = 1.f + a[j]; sc8[j + 4] = 1.f + bq[j]; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float gg = dyv[j];
                if (scp != nullptr) gg *= sc8[j];
                if (p.gamma != nullptr) gg *= ga[j];
                float xhat = (xv[j] - mean) * rstd;
                gv[i][j] = gg;
                xh[i][j] = xhat;
                s1 += gg;
                s2 += gg * xhat;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { gv[i][j] = 0.f; xh[i][j] = 0.f; }
        }
    }
    const float invD = 1.f / (float)p.D;
    const float m1 = wave_sum(s1) * invD, m2 = wave_sum(s2) * invD;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunks) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = rstd * (gv[i][j] - m1 - xh[i][j] * m2);
            if (p.dres != nullptr) {
                float dr[8];
                unpack8(*(const u32x4*)(p.dres + (size_t)m * p.lddres + c * 8), dr);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += dr[j];
            }
            *(u32x4*)(p.dx + (size_t)m * p.lddx + c * 8) = pack8(o);
        }
    }
}

static int nch_for(int D) { return ((D >> 3) + 63) / 64; }

extern "C" int vt_ln_modulate_fwd(const void* x, int ldx, void* y, int ldy, const void* gamma, const void* beta,
                                  const float* shift_txt, const float* scale_txt, const float* shift_vid,
                                  const float* scale_vid, int mod_bstride, float* mean, float* rstd,
                                  int M, int D, int S, int St, float eps, void* stream) {
    if (M <= 0 || D <= 0 || (D % 8) || D > 4096 || (ldx % 8) || (ldy % 8)) return VT_ERR_BAD_SHAPE;
    if ((gamma == nullptr) != (beta == nullptr)) return VT_ERR_BAD_SHAPE;
    if (shift_vid != nullptr && (shift_txt == nullptr || scale_txt == nullptr || scale_vid == nullptr || (mod_bstride % 4)))
        return VT_ERR_BAD_SHAPE;
    if ((((uintptr_t)x) | ((uintptr_t)y)) & 15) return VT_ERR_BAD_ALIGN;
    LnParams p{(const bf16_t*)x, ldx, (bf16_t*)y, ldy, (const bf16_t*)gamma, (const bf16_t*)beta,
               shift_txt, scale_txt, shift_vid, scale_vid, mod_bstride, mean, rstd, M, D, S > 0 ? S : 1, St, eps};
    dim3 grid((M + 3) / 4), block(256);
    hipStream_t st = (hipStream_t)stream;
    const int nch = nch_for(D);
    if (nch <= 1) hipLaunchKernelGGL(ln_modulate_fwd_kernel<1>, grid, block, 0, st, p);
    else if (nch <= 4) hipLaunchKernelGGL(ln_modulate_fwd_kernel<4>, grid, block, 0, st, p);
    else hipLaunchKernelGGL(ln_modulate_fwd_kernel<8>, grid, block, 0, st, p);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}

extern "C" int vt_ln_modulate_bwd(const void* dy, int lddy, const void* x, int ldx, const float* mean, const float* rstd,
                                  const void* gamma, const float* scale_txt, const float* scale_vid, int mod_bstride,
                                  const void* dres, int lddres, void* dx, int lddx,
                                  int M, int D, int S, int St, void* stream) {
    if (M <= 0 || D <= 0 || (D % 8) || D > 4096 || (ldx % 8) || (lddy % 8) || (lddx % 8)) return VT_ERR_BAD_SHAPE;
    if (dres != nullptr && (lddres % 8)) return VT_ERR_BAD_SHAPE;
    if (scale_vid != nullptr && (scale_txt == nullptr || (mod_bstride % 4))) return VT_ERR_BAD_SHAPE;
    if ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx) | ((uintptr_t)dres)) & 15) return VT_ERR_BAD_ALIGN;
    LnBwdParams p{(const bf16_t*)dy, lddy, (const bf16_t*)x, ldx, mean, rstd, (const bf16_t*)gamma,
                  scale_txt, scale_vid, mod_bstride, (const bf16_t*)dres, lddres, (bf16_t*)dx, lddx,
                  M, D, S > 0 ? S : 1, St};
    dim3 grid((M + 3) / 4), block(256);
    hipStream_t st = (hipStream_t)stream;
    const int nch = nch_for(D);
    if (nch <= 1) hipLaunchKernelGGL(ln_modulate_bwd_kernel<1>, grid, block, 0, st, p);
    else if (nch <= 4) hipLaunchKernelGGL(ln_modulate_bwd_kernel<4>, grid, block, 0, st, p);
    else hipLaunchKernelGGL(ln_modulate_bwd_kernel<8>, grid, block, 0, st, p);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------------
// per-head LayerNorm(64) of q and k.  8 lanes per (row, which, head) group of 64 elements.
// Optional rotary position embedding (CogVideoX-5B, cogvideo_pl.py:442-473 builds the tables, diffusers'
// CogVideoXAttnProcessor2_0 applies them to the video rows of q and k after the LayerNorm): tables are fp32 [S-St, 64].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void qk_layernorm_fwd_kernel(const bf16_t* qkv, int ld, bf16_t* out, int ldo,
                                                              const bf16_t* gq, const bf16_t* bq, const bf16_t* gk,
                                                              const bf16_t* bk, float* mean, float* rstd,
                                                              long long M, int H, float eps, float q_scale,
                                                              const float* rope_cos, const float* rope_sin, int S, int St) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long grp = gid >> 3;
    const int sub = (int)(gid & 7);
    const long long total = M * 2 * H;
    const bool ok = grp < total;
    const long long m = ok ? grp / (2 * H) : 0;
    const int wh = ok ? (int)(grp % (2 * H)) : 0;       // which*H + head ; which 0 = q, 1 = k
    float v[8];
    if (ok) unpack8(*(const u32x4*)(qkv + (size_t)m * ld + wh * 64 + sub * 8), v);
    else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    const float mu = s * (1.f / 64.f);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { float d = v[j] - mu; q += d * d; }
    q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
    const float rs = rsqrtf(q * (1.f / 64.f) + eps);
    if (!ok) return;
    const bool isk = wh >= H;
    float ga[8], be[8];
    unpack8(*(const u32x4*)((isk ? gk : gq) + sub * 8), ga);
    unpack8(*(const u32x4*)((isk ? bk : bq) + sub * 8), be);
    const float osc = isk ? 1.0f : q_scale;     // q_hat may be pre-multiplied by softmax_scale*log2(e) for the attention kernels
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (v[j] - mu) * rs * ga[j] + be[j];
    // rotary embedding of the video rows (CogVideoX-5B): out = x*cos + rotate_pairs(x)*sin, pairs (2i, 2i+1) are lane-local
    const int spos = rope_cos != nullptr ? (int)(m % S) - St : -1;
    if (spos >= 0) {
        float cs[8], sn[8];
        rope_load8(rope_cos + (size_t)spos * 64 + sub * 8, cs);
        rope_load8(rope_sin + (size_t)spos * 64 + sub * 8, sn);
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const float a = o[j], b = o[j + 1];
            o[j] = a * cs[j] - b * sn[j];
            o[j + 1] = b * cs[j + 1] + a * sn[j + 1];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] *= osc;
    *(u32x4*)(out + (size_t)m * ldo + wh * 64 + sub * 8) = pack8(o);
    if (sub == 0) { mean[m * 2 * H + wh] = mu; rstd[m * 2 * H + wh] = rs; }
}

// dq_hat is fp32 (the attention backward's atomic accumulation buffer), dk_hat is bf16.
__global__ __launch_bounds__(256) void qk_layernorm_bwd_kernel(const float* dqh, int lddq, const bf16_t* dkh, int lddk,
                                                              const bf16_t* qkv, int ld, const float* mean,
                                                              const float* rstd, const bf16_t* gq, const bf16_t* gk,
                                                              bf16_t* dqkv, int ldd, long long M, int H,
                                                              const float* rope_cos, const float* rope_sin, int S, int St) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long grp = gid >> 3;
    const int sub = (int)(gid & 7);
    const long long total = M * 2 * H;
    const bool ok = grp < total;
    const long long m = ok ? grp / (2 * H) : 0;
    const int wh = ok ? (int)(grp % (2 * H)) : 0;
    const bool isk = wh >= H;
    float g[8], xh[8];
    float mu = 0.f, rs = 0.f;
    if (ok) {
        mu = mean[m * 2 * H + wh];
        rs = rstd[m * 2 * H + wh];
        float dy[8], xv[8], ga[8];
        if (isk) {
            unpack8(*(const u32x4*)(dkh + (size_t)m * lddk + (wh - H) * 64 + sub * 8), dy);
        } else {
            const float* src = dqh + (size_t)m * lddq + wh * 64 + sub * 8;
            f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { dy[j] = a[j]; dy[j + 4] = b[j]; }
        }
        const int spos = rope_cos != nullptr ? (int)(m % S) - St : -1;
        if (spos >= 0) rope_bwd8(rope_cos, rope_sin, spos, sub, dy);
        unpack8(*(const u32x4*)(qkv + (size_t)m * ld + wh * 64 + sub * 8), xv);
        unpack8(*(const u32x4*)((isk ? gk : gq) + sub * 8), ga);
#pragma unroll
        for (int j = 0; j < 8; ++j) { g[j] = dy[j] * ga[j]; xh[j] = (xv[j] - mu) * rs; }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { g[j] = 0.f; xh[j] = 0.f; }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1 += g[j]; s2 += g[j] * xh[j]; }
    s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64); s1 += __shfl_xor(s1, 4, 64);
    s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64); s2 += __shfl_xor(s2, 4, 64);
    if (!ok) return;
    const float m1 = s1 * (1.f / 64.f), m2 = s2 * (1.f / 64.f);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = rs * (g[j] - m1 - xh[j] * m2);
    *(u32x4*)(dqkv + (size_t)m * ldd + wh * 64 + sub * 8) = pack8(o);
}

extern "C" int vt_qk_layernorm_fwd(const void* qkv, int ld, void* out, int ldo, const void* gq, const void* bq,
                                   const void* gk, const void* bk, float* mean, float* rstd,
                                   long long M, int H, float eps, float q_scale,
                                   const float* rope_cos, const float* rope_sin, int S, int St, void* stream) {
    if (M <= 0 || H <= 0 || (ld % 8) || (ldo % 8) || ld < 2 * H * 64 || ldo < 2 * H * 64) return VT_ERR_BAD_SHAPE;
    if ((rope_cos == nullptr) != (rope_sin == nullptr)) return VT_ERR_BAD_SHAPE;
    if (rope_cos != nullptr && (S <= 0 || St < 0 || St > S || (M % S))) return VT_ERR_BAD_SHAPE;
    if ((((uintptr_t)rope_cos) | ((uintptr_t)rope_sin)) & 15) return VT_ERR_BAD_ALIGN;
    if ((((uintptr_t)qkv) | ((uintptr_t)out) | ((uintptr_t)gq) | ((uintptr_t)bq) | ((uintptr_t)gk) | ((uintptr_t)bk)) & 15)
        return VT_ERR_BAD_ALIGN;
    const long long threads = M * 2 * H * 8;
    const long long blocks = (threads + 255) / 256;
    if (blocks > 0x7fffffffLL) return VT_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(qk_layernorm_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)qkv, ld, (bf16_t*)out, ldo, (const bf16_t*)gq, (const bf16_t*)bq,
                       (const bf16_t*)gk, (const bf16_t*)bk, mean, rstd, M, H, eps, q_scale, rope_cos, rope_sin,
                       S > 0 ? S : 1, St);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}

extern "C" int vt_qk_layernorm_bwd(const float* dq_hat, int lddq, const void* dk_hat, int lddk, const void* qkv, int ld,
                                   const float* mean, const float* rstd, const void* gq, const void* gk,
                                   void* dqkv, int ldd, long long M, int H,
                                   const float* rope_cos, const float* rope_sin, int S, int St, void* stream) {
    if (M <= 0 || H <= 0 || (ld % 8) || (ldd % 8) || (lddq % 4) || (lddk % 8)) return VT_ERR_BAD_SHAPE;
    if ((rope_cos == nullptr) != (rope_sin == nullptr)) return VT_ERR_BAD_SHAPE;
    if (rope_cos != nullptr && (S <= 0 || St < 0 || St > S || (M % S))) return VT_ERR_BAD_SHAPE;
    if ((((uintptr_t)rope_cos) | ((uintptr_t)rope_sin)) & 15) return VT_ERR_BAD_ALIGN;
    if ((((uintptr_t)qkv) | ((uintptr_t)dqkv) | ((uintptr_t)dq_hat) | ((uintptr_t)dk_hat) | ((uintptr_t)gq) | ((uintptr_t)gk)) & 15)
        return VT_ERR_BAD_ALIGN;
    const long long threads = M * 2 * H * 8;
    const long long blocks = (threads + 255) / 256;
    if (blocks > 0x7fffffffLL) return VT_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(qk_layernorm_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       dq_hat, lddq, (const bf16_t*)dk_hat, lddk, (const bf16_t*)qkv, ld, mean, rstd,
                       (const bf16_t*)gq, (const bf16_t*)gk, (bf16_t*)dqkv, ldd, M, H, rope_cos, rope_sin, S > 0 ? S : 1, St);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}
