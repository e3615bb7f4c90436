// Per-head RMS q/k normalisation + rotary embedding of the image tokens for head_dim 128, forward and backward, and the scatter of
// q^ | k^ | v into the joint [image; text] sequence the attention kernel reads -- HunyuanVideo's MMDoubleStreamBlock / MMSingleStreamBlock
// (videotuna/models/hunyuan/hyvideo_t2v/modules/models.py:166-196, 351-361; RMSNorm modules/norm_layers.py:5-58, eps 1e-6, fp32
// statistics; apply_rotary_emb modules/posemb_layers.py:133-188: pairs (2i, 2i+1), x cos + rotate_half(x) sin).  SURVEY 8(a) a16.
// The reference runs rearrange -> RMSNorm(q), RMSNorm(k) -> rotary on a slice -> torch.cat((img, txt)) as separate passes; here one
// kernel reads the fused qkv projection once and writes the attention operand once.  HBM-bound.
//   in : qkv bf16 [M, ld], thirds q | k | v of H*128 each; row m = (sample m / L, position m % L)
//   out: bf16 [., ldo], thirds q^ | k^ | v; row = sample * Lout + row_off + position   (Lout >= row_off + L: the joint sequence)
//   positions < S_rope rotate with cos / sin fp32 [S_rope, 128];  rstd fp32 [M, 2H] is kept for the backward.
#include "common.h"

__device__ __forceinline__ void qn_load8f(const float* p, float* d) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { d[j] = a[j]; d[j + 4] = b[j]; }
}

__global__ __launch_bounds__(256) void qk_rmsnorm_rope_fwd_kernel(const bf16_t* qkv, long long ld, bf16_t* out, long long ldo, const bf16_t* gq,
                                                                 const bf16_t* gk, float* rstd, const float* cs, const float* sn, long long M, int H,
                                                                 int L, int Lout, int row_off, int S_rope, float eps) {
    const int grp = threadIdx.x >> 4, sub = threadIdx.x & 15;              // 16 lanes x 8 elements = one 128-wide head vector
    const long long unit = (long long)blockIdx.x * 16 + grp;              // (row, which in {q,k,v}, head)
    const long long units = M * 3 * H;
    if (unit >= units) return;
    const int hh = (int)(unit % H);
    const int which = (int)((unit / H) % 3);
    const long long m = unit / (3LL * H);
    const int pos = (int)(m % L);
    const long long orow = (m / L) * Lout + row_off + pos;
    const long long col = (long long)which * H * 128 + hh * 128 + sub * 8;
    u32x4 raw = *(const u32x4*)(qkv + m * ld + col);
    if (which == 2) { *(u32x4*)(out + orow * ldo + col) = raw; return; }
    float x[8], g[8];
    unpack8(raw, x);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += x[j] * x[j];
    ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64); ss += __shfl_xor(ss, 8, 64);
    const float r = rsqrtf(ss * (1.0f / 128.0f) + eps);
    if (sub == 0) rstd[m * 2 * H + which * H + hh] = r;
    unpack8(*(const u32x4*)((which == 0 ? gq : gk) + sub * 8), g);
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = bf2f(f2bf(x[j] * r)) * g[j];          // norm output is cast back to the activation dtype before the weight
    if (cs != nullptr && pos < S_rope) {
        float c[8], s[8];
        qn_load8f(cs + (long long)pos * 128 + sub * 8, c); qn_load8f(sn + (long long)pos * 128 + sub * 8, s);
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const float a = y[j], b = y[j + 1];
            y[j] = a * c[j] - b * s[j];
            y[j + 1] = b * c[j + 1] + a * s[j + 1];
        }
    }
    *(u32x4*)(out + orow * ldo + col) = pack8(y);
}

// dout: gradients of q^ | k^ | v in the OUT layout; dqkv: gradient of the fused projection in the IN layout; dgq / dgk fp32 [128] accumulated
__global__ __launch_bounds__(256) void qk_rmsnorm_rope_bwd_kernel(const bf16_t* dout, long long lddo, const bf16_t* qkv, long long ld, bf16_t* dqkv,
                                                                 long long ldd, const bf16_t* gq, const bf16_t* gk, const float* rstd, const float* cs,
                                                                 const float* sn, float* dgq, float* dgk, long long M, int H, int L, int Lout,
                                                                 int row_off, int S_rope) {
    // dgq / dgk: every 16-lane group leaves its unit's 128 products in its own LDS row (plain stores -- the first version added them with LDS
    // atomics into one shared row and spent 6x the kernel's memory time on the conflicts), the block sums the 16 rows and adds once per column;
    // dgq == nullptr (frozen norm weights, LoRA mode): skipped altogether
    __shared__ float red[16][256];
    const bool want = dgq != nullptr;
    const int grp = threadIdx.x >> 4, sub = threadIdx.x & 15;
    if (want) {
#pragma unroll
        for (int j = 0; j < 16; ++j) red[grp][sub * 16 + j] = 0.f;
    }
    const long long unit = (long long)blockIdx.x * 16 + grp;
    const long long units = M * 3 * H;
    if (unit < units) {
        const int hh = (int)(unit % H);
        const int which = (int)((unit / H) % 3);
        const long long m = unit / (3LL * H);
        const int pos = (int)(m % L);
        const long long orow = (m / L) * Lout + row_off + pos;
        const long long col = (long long)which * H * 128 + hh * 128 + sub * 8;
        const u32x4 draw = *(const u32x4*)(dout + orow * lddo + col);
        if (which == 2) {
            *(u32x4*)(dqkv + m * ldd + col) = draw;
        } else {
            float dy[8], x[8], g[8];
            unpack8(draw, dy);
            unpack8(*(const u32x4*)(qkv + m * ld + col), x);
            unpack8(*(const u32x4*)((which == 0 ? gq : gk) + sub * 8), g);
            if (cs != nullptr && pos < S_rope) {              // transposed rotation
                float c[8], s[8];
                qn_load8f(cs + (long long)pos * 128 + sub * 8, c); qn_load8f(sn + (long long)pos * 128 + sub * 8, s);
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const float a = dy[j], b = dy[j + 1];
                    dy[j] = a * c[j] + b * s[j + 1];
                    dy[j + 1] = b * c[j + 1] - a * s[j];
                }
            }
            const float r = rstd[m * 2 * H + which * H + hh];
            float dot = 0.f, dxn[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                dxn[j] = dy[j] * g[j]; dot += dxn[j] * x[j];
                if (want) red[grp][which * 128 + sub * 8 + j] = dy[j] * x[j] * r;
            }
            dot += __shfl_xor(dot, 1, 64); dot += __shfl_xor(dot, 2, 64); dot += __shfl_xor(dot, 4, 64); dot += __shfl_xor(dot, 8, 64);
            const float k3 = r * r * r * dot * (1.0f / 128.0f);
            float dx[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) dx[j] = r * dxn[j] - x[j] * k3;
            *(u32x4*)(dqkv + m * ldd + col) = pack8(dx);
        }
    }
    if (want) {
        __syncthreads();
        float sum = 0.f;
#pragma unroll
        for (int gi = 0; gi < 16; ++gi) sum += red[gi][threadIdx.x];
        if (sum != 0.f) {
            if (threadIdx.x < 128) atomicAdd(dgq + threadIdx.x, sum);
            else atomicAdd(dgk + threadIdx.x - 128, sum);
        }
    }
}

extern "C" int vt_qk_rmsnorm_rope128_fwd(const void* qkv, long long ld, void* out, long long ldo, const void* gq, const void* gk, float* rstd,
                                         const float* rope_cos, const float* rope_sin, long long M, int H, int L, int Lout, int row_off,
                                         int S_rope, float eps, void* stream) {
    if (M <= 0 || H <= 0 || L <= 0 || (M % L) || Lout < row_off + L || row_off < 0 || (ld % 8) || (ldo % 8) || ld < 3LL * H * 128 || ldo < 3LL * H * 128)
        return VT_ERR_BAD_SHAPE;
    if ((rope_cos == nullptr) != (rope_sin == nullptr) || S_rope < 0 || S_rope > L) return VT_ERR_BAD_SHAPE;
    if ((((uintptr_t)qkv) | ((uintptr_t)out) | ((uintptr_t)gq) | ((uintptr_t)gk)) & 15) return VT_ERR_BAD_ALIGN;
    const long long units = M * 3 * H;
    hipLaunchKernelGGL(qk_rmsnorm_rope_fwd_kernel, dim3((unsigned)((units + 15) / 16)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, ld,
                       (bf16_t*)out, ldo, (const bf16_t*)gq, (const bf16_t*)gk, rstd, rope_cos, rope_sin, M, H, L, Lout, row_off, S_rope, eps);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}
extern "C" int vt_qk_rmsnorm_rope128_bwd(const void* dout, long long lddo, const void* qkv, long long ld, void* dqkv, long long ldd, const void* gq,
                                         const void* gk, const float* rstd, const float* rope_cos, const float* rope_sin, float* dgq, float* dgk,
                                         long long M, int H, int L, int Lout, int row_off, int S_rope, void* stream) {
    if (M <= 0 || H <= 0 || L <= 0 || (M % L) || Lout < row_off + L || (ld % 8) || (lddo % 8) || (ldd % 8) || ((dgq == nullptr) != (dgk == nullptr)))
        return VT_ERR_BAD_SHAPE;
    if ((((uintptr_t)qkv) | ((uintptr_t)dout) | ((uintptr_t)dqkv)) & 15) return VT_ERR_BAD_ALIGN;
    const long long units = M * 3 * H;
    hipLaunchKernelGGL(qk_rmsnorm_rope_bwd_kernel, dim3((unsigned)((units + 15) / 16)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dout, lddo,
                       (const bf16_t*)qkv, ld, (bf16_t*)dqkv, ldd, (const bf16_t*)gq, (const bf16_t*)gk, rstd, rope_cos, rope_sin, dgq, dgk, M, H, L,
                       Lout, row_off, S_rope);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}
