// Small HBM-bound kernels around the DiT: time embedding, patchify, diffusion forward process + loss,
// gate multiply, SiLU, fused AdamW.  Reference semantics:
//   add_noise / get_velocity / 1/(1-abar) weighted MSE ... videotuna/models/cogvideo_hf/cogvideo_pl.py:864-886
//   AdamW ................................................ cogvideo_pl.py:774-779 (torch.optim.AdamW defaults)
//   sinusoid (cos first) ................................. videotuna/utils/diffusion_utils.py:9-33
//   patchify (c p q) / token order (t h w) ............... videotuna/models/cogvideo_sat/dit_video_concat.py:20-56,434-454
#include "common.h"
#include <math.h>

// ---------------- y = x * gate[b(m), seg(m)] (rows of D bf16) ----------------
__global__ __launch_bounds__(256) void gate_mul_kernel(const bf16_t* x, int ldx, bf16_t* y, int ldy, const float* g_txt,
                                                      const float* g_vid, int bstride, long long M, int D, int S, int St) {
    const int nch = D >> 3;
    const long long total = M * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long m = i / nch;
        const int c = (int)(i - m * nch);
        const int b = (int)(m / S);
        const int s = (int)(m - (long long)b * S);
        const float* g = (s < St ? g_txt : g_vid) + (size_t)b * bstride + c * 8;
        float v[8];
        unpack8(*(const u32x4*)(x + (size_t)m * ldx + c * 8), v);
        f32x4 a = *(const f32x4*)g, bq = *(const f32x4*)(g + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] *= a[j]; v[j + 4] *= bq[j]; }
        *(u32x4*)(y + (size_t)m * ldy + c * 8) = pack8(v);
    }
}

extern "C" int vt_gate_mul(const void* x, int ldx, void* y, int ldy, const float* g_txt, const float* g_vid, int bstride,
                           long long M, int D, int S, int St, void* stream) {
    if (M <= 0 || D <= 0 || (D % 8) || (ldx % 8) || (ldy % 8) || (bstride % 4) || S <= 0) return VT_ERR_BAD_SHAPE;
    if ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)g_txt) | ((uintptr_t)g_vid)) & 15) return VT_ERR_BAD_ALIGN;
    long long total = M * (D >> 3);
    int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(gate_mul_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (bf16_t*)y, ldy,
                       g_txt, g_vid, bstride, M, D, S, St);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}

// ---------------- SiLU (bf16 -> bf16) ----------------
__global__ void silu_kernel(const bf16_t* x, bf16_t* y, long long n) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = f2bf(silu_f(bf2f(x[i])));
}
extern "C" int vt_silu_bf16(const void* x, void* y, long long n, void* stream) {
    if (n <= 0) return VT_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(silu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, n);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}

// ---------------- fp32 -> bf16 cast ----------------
__global__ void cast_f32_bf16_kernel(const float* x, bf16_t* y, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = f2bf(x[i]);
}
extern "C" int vt_cast_f32_bf16(const float* x, void* y, long long n, void* stream) {
    if (n <= 0) return VT_ERR_BAD_SHAPE;
    long long b = (n + 255) / 256;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)(b > 8192 ? 8192 : b)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, n);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}

// ---------------- sinusoidal timestep embedding: [cos | sin] when flip_sin_to_cos ----------------
__global__ void timestep_embedding_kernel(const long long* t, bf16_t* out, int B, int D, int flip, float freq_shift,
                                          float max_period) {
    const int half = D / 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, k = i - b * half;
    const float freq = expf(-logf(max_period) * (float)k / ((float)half - freq_shift));
    const float arg = (float)t[b] * freq;
    const float sv = sinf(arg), cv = cosf(arg);
    bf16_t* o = out + (size_t)b * D;
    if (flip) { o[k] = f2bf(cv); o[half + k] = f2bf(sv); }
    else { o[k] = f2bf(sv); o[half + k] = f2bf(cv); }
}
extern "C" int vt_timestep_embedding(const long long* t, void* out, int B, int D, int flip_sin_to_cos, float freq_shift,
                                     void* stream) {
    if (B <= 0 || D <= 0 || (D & 1)) return VT_ERR_BAD_SHAPE;
    const int n = B * (D / 2);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, (bf16_t*)out, B, D,
                       flip_sin_to_cos, freq_shift, 10000.0f);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}

// ---------------- patchify / unpatchify: [B,F,C,H,W] <-> [B*F*h*w, C*p*p] (c p q), tokens (t h w) ----------------
template <bool TO_TOKENS>
__global__ void patch_kernel(bf16_t* img, bf16_t* tok, int B, int F, int C, int H, int W, int P, int ldt) {
    const long long n = (long long)B * F * C * H * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        long long r = i;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H); r /= H;
        const int c = (int)(r % C); r /= C;
        const int f = (int)(r % F); r /= F;
        const int b = (int)r;
        const int hh = H / P, ww = W / P;
        const long long token = (((long long)b * F + f) * hh + y / P) * ww + x / P;
        const int col = (c * P + (y % P)) * P + (x % P);
        if (TO_TOKENS) tok[token * ldt + col] = img[i];
        else img[i] = tok[token * ldt + col];
    }
}
extern "C" int vt_patchify(const void* img, void* tok, int B, int F, int C, int H, int W, int P, int ldt, void* stream) {
    if (B <= 0 || F <= 0 || C <= 0 || H <= 0 || W <= 0 || P <= 0 || (H % P) || (W % P) || ldt < C * P * P) return VT_ERR_BAD_SHAPE;
    long long n = (long long)B * F * C * H * W, b = (n + 255) / 256;
    hipLaunchKernelGGL(patch_kernel<true>, dim3((unsigned)(b > 8192 ? 8192 : b)), dim3(256), 0, (hipStream_t)stream,
                       (bf16_t*)const_cast<void*>(img), (bf16_t*)tok, B, F, C, H, W, P, ldt);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}
extern "C" int vt_unpatchify(const void* tok, void* img, int B, int F, int C, int H, int W, int P, int ldt, void* stream) {
    if (B <= 0 || F <= 0 || C <= 0 || H <= 0 || W <= 0 || P <= 0 || (H % P) || (W % P) || ldt < C * P * P) return VT_ERR_BAD_SHAPE;
    long long n = (long long)B * F * C * H * W, b = (n + 255) / 256;
    hipLaunchKernelGGL(patch_kernel<false>, dim3((unsigned)(b > 8192 ? 8192 : b)), dim3(256), 0, (hipStream_t)stream,
                       (bf16_t*)img, (bf16_t*)const_cast<void*>(tok), B, F, C, H, W, P, ldt);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}

// ---------------- forward diffusion: noisy = sqrt(abar) x0 + sqrt(1-abar) eps (fp32 in, bf16 out) ----------------
__global__ void add_noise_kernel(const float* x0, const float* noise, const float* sa, const float* sb, bf16_t* out,
                                 long long per, int B) {
    const long long n = per * B;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / per);
        out[i] = f2bf(sa[b] * x0[i] + sb[b] * noise[i]);
    }
}
extern "C" int vt_add_noise(const float* x0, const float* noise, const float* sqrt_ab, const float* sqrt_1mab, void* noisy,
                            long long per_sample, int B, void* stream) {
    if (per_sample <= 0 || B <= 0) return VT_ERR_BAD_SHAPE;
    long long b = (per_sample * B + 255) / 256;
    hipLaunchKernelGGL(add_noise_kernel, dim3((unsigned)(b > 4096 ? 4096 : b)), dim3(256), 0, (hipStream_t)stream, x0, noise,
                       sqrt_ab, sqrt_1mab, (bf16_t*)noisy, per_sample, B);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}

// ---------------- loss: x0_hat = sa*noisy - sb*v ; L = mean_b mean_i w_b (x0_hat - x0)^2 ; also dL/dv ----------------
#define LOSS_BLOCKS 512
__global__ __launch_bounds__(256) void loss_kernel(const bf16_t* vpred, const bf16_t* noisy, const float* x0, const float* sa,
                                                   const float* sb, const float* w, float* partials, bf16_t* dv,
                                                   long long per, int B, float gscale) {
    __shared__ float red[4];
    const long long n = per * B;
    const float inv = 1.0f / ((float)per * (float)B);
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / per);
        const float pred = sa[b] * bf2f(noisy[i]) - sb[b] * bf2f(vpred[i]);
        const float diff = pred - x0[i];
        acc += w[b] * diff * diff;
        if (dv != nullptr) dv[i] = f2bf(gscale * inv * 2.0f * w[b] * diff * (-sb[b]));
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1] + red[2] + red[3]) * inv;
}
__global__ void loss_final_kernel(const float* partials, float* loss, int n) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += partials[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = red[0] + red[1] + red[2] + red[3];
}
extern "C" int vt_diffusion_loss(const void* vpred, const void* noisy, const float* x0, const float* sqrt_ab,
                                 const float* sqrt_1mab, const float* weights, float* loss, float* partials_ws /*>=512*/,
                                 void* dvpred, long long per_sample, int B, float grad_scale, void* stream) {
    if (per_sample <= 0 || B <= 0) return VT_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(loss_kernel, dim3(LOSS_BLOCKS), dim3(256), 0, st, (const bf16_t*)vpred, (const bf16_t*)noisy, x0, sqrt_ab,
                       sqrt_1mab, weights, partials_ws, (bf16_t*)dvpred, per_sample, B, grad_scale);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, partials_ws, loss, LOSS_BLOCKS);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}

// backward of the loss wrt the v-prediction; the upstream gradient is read from DEVICE memory (no host sync)
__global__ __launch_bounds__(256) void loss_bwd_kernel(const bf16_t* vpred, const bf16_t* noisy, const float* x0, const float* sa,
                                                       const float* sb, const float* w, const float* gout, bf16_t* dv,
                                                       long long per, int B) {
    const long long n = per * B;
    const float g = gout[0] * 2.0f / ((float)per * (float)B);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / per);
        const float pred = sa[b] * bf2f(noisy[i]) - sb[b] * bf2f(vpred[i]);
        dv[i] = f2bf(g * w[b] * (pred - x0[i]) * (-sb[b]));
    }
}
extern "C" int vt_diffusion_loss_bwd(const void* vpred, const void* noisy, const float* x0, const float* sqrt_ab,
                                     const float* sqrt_1mab, const float* weights, const float* grad_out, void* dvpred,
                                     long long per_sample, int B, void* stream) {
    if (per_sample <= 0 || B <= 0) return VT_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(LOSS_BLOCKS), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)vpred,
                       (const bf16_t*)noisy, x0, sqrt_ab, sqrt_1mab, weights, grad_out, (bf16_t*)dvpred, per_sample, B);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}

// ---------------- fused AdamW over one flat fp32 buffer (+ bf16 compute copy) ----------------
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, bf16_t* pb, long long n,
                                                    float lr, float b1, float b2, float eps, float wd, float bc1, float bc2,
                                                    float gscale, const int* guard) {
    if (guard != nullptr && *guard != 0) return;      // a kernel of this step flagged its results invalid: refuse the update
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gr = g[i] * gscale;
        float pv = p[i] * (1.0f - lr * wd);
        const float mn = b1 * m[i] + (1.0f - b1) * gr;
        const float vn = b2 * v[i] + (1.0f - b2) * gr * gr;
        const float denom = sqrtf(vn) / sqrtf(bc2) + eps;
        pv -= (lr / bc1) * mn / denom;
        p[i] = pv; m[i] = mn; v[i] = vn;
        if (pb != nullptr) pb[i] = f2bf(pv);
    }
}
extern "C" int vt_adamw(float* p, const float* g, float* m, float* v, void* p_bf16, long long n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int step, float grad_scale, const int* guard,
                        void* stream) {
    if (n <= 0 || step <= 0) return VT_ERR_BAD_SHAPE;
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step)), bc2 = (float)(1.0 - pow((double)beta2, (double)step));
    long long b = (n + 255) / 256;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)(b > 16384 ? 16384 : b)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                       (bf16_t*)p_bf16, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, guard);
    return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}
