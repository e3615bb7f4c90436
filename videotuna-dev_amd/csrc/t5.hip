// HBM-bound kernels of the frozen T5 text encoder (the step before the DiT,
// videotuna/models/cogvideo_hf/cogvideo_pl.py:254-286 -> transformers' T5EncoderModel, T5 v1.1 XXL: d_model 4096,
// 64 heads x 64, d_ff 10240, gated-GELU, 24 layers).  The linears run on vt_gemm_bf16, the attention on
// vt_attn_fwd_bias_hd64; what is left are the two row-wise pieces below.
//   T5LayerNorm ....... y = x * rsqrt(mean(x^2) + eps) * w   (no mean subtraction, no bias; fp32 statistics)
//   T5DenseGatedActDense  h = gelu_new(x Wi0^T) * (x Wi1^T)  -- the two products come from one fused GEMM [M, 2F]
#include "common.h"

__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* x, long long ldx, const bf16_t* w, bf16_t* y, long long ldy,
                                                      int D, float eps) {
    __shared__ float red[4];
    const long long m = blockIdx.x;
    const bf16_t* xr = x + m * ldx;
    const int nch = D >> 3;
    float ss = 0.f;
    for (int c = threadIdx.x; c < nch; c += 256) {
        float v[8];
        unpack8(*(const u32x4*)(xr + c * 8), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += v[j] * v[j];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float rs = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)D + eps);
    bf16_t* yr = y + m * ldy;
    for (int c = threadIdx.x; c < nch; c += 256) {
        float v[8], g[8];
        unpack8(*(const u32x4*)(xr + c * 8), v);
        unpack8(*(const u32x4*)(w + c * 8), g);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * rs * g[j];
        *(u32x4*)(yr + c * 8) = pack8(v);
    }
}

extern "C" int vt_rmsnorm_bf16(const void* x, long long ldx, const void* w, void* y, long long ldy, long long M, int D, float eps,
                               void* stream) {
    if (M <= 0 || M > 0x7fffffffLL || D <= 0 || (D % 8) || (ldx % 8) || (ldy % 8) || ldx < D || ldy < D) return VT_ERR_BAD_SHAPE;
    if ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)w)) & 15) return VT_ERR_BAD_ALIGN;
    hipLaunchKernelGGL(rmsnorm_kernel, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (const bf16_t*)w,
                       (bf16_t*)y, ldy, D, eps);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}

// y[m, f] = gelu_tanh(u[m, f]) * u[m, F + f]
__global__ __launch_bounds__(256) void gated_gelu_kernel(const bf16_t* u, long long ldu, bf16_t* y, long long ldy, long long M, int F) {
    const int nch = F >> 3;
    const long long total = M * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long m = i / nch;
        const int c = (int)(i - m * nch);
        float a[8], b[8];
        unpack8(*(const u32x4*)(u + m * ldu + c * 8), a);
        unpack8(*(const u32x4*)(u + m * ldu + F + c * 8), b);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = gelu_tanh_f(a[j]) * b[j];
        *(u32x4*)(y + m * ldy + c * 8) = pack8(a);
    }
}

extern "C" int vt_gated_gelu_bf16(const void* u, long long ldu, void* y, long long ldy, long long M, int F, void* stream) {
    if (M <= 0 || F <= 0 || (F % 8) || (ldu % 8) || (ldy % 8) || ldu < 2 * (long long)F || ldy < F) return VT_ERR_BAD_SHAPE;
    if ((((uintptr_t)u) | ((uintptr_t)y)) & 15) return VT_ERR_BAD_ALIGN;
    const long long total = M * (F >> 3);
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(gated_gelu_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)u, ldu, (bf16_t*)y, ldy, M, F);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}

// out[m, n] = bf16(acc[m, n] (+ R[m, n])): the finishing pass of vt_gemm_splitk_f32 (fp32 partial sums -> bf16, residual add)
__global__ __launch_bounds__(256) void residual_cast_kernel(const float* acc, long long lda, const bf16_t* R, long long ldr, bf16_t* out,
                                                            long long ldo, long long M, int N) {
    const int nch = N >> 3;
    const long long total = M * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long m = i / nch;
        const int c = (int)(i - m * nch);
        const f32x4 a0 = *(const f32x4*)(acc + m * lda + c * 8), a1 = *(const f32x4*)(acc + m * lda + c * 8 + 4);
        float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        if (R != nullptr) {
            float r[8];
            unpack8(*(const u32x4*)(R + m * ldr + c * 8), r);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += r[j];
        }
        *(u32x4*)(out + m * ldo + c * 8) = pack8(v);
    }
}

extern "C" int vt_residual_cast_bf16(const float* acc, long long lda, const void* R, long long ldr, void* out, long long ldo, long long M,
                                     int N, void* stream) {
    if (M <= 0 || N <= 0 || (N % 8) || (lda % 4) || (ldo % 8) || lda < N || ldo < N || (R != nullptr && ((ldr % 8) || ldr < N)))
        return VT_ERR_BAD_SHAPE;
    if ((((uintptr_t)acc) | ((uintptr_t)out) | ((uintptr_t)R)) & 15) return VT_ERR_BAD_ALIGN;
    const long long total = M * (N >> 3);
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(residual_cast_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, (hipStream_t)stream, acc, lda,
                       (const bf16_t*)R, ldr, (bf16_t*)out, ldo, M, N);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}
