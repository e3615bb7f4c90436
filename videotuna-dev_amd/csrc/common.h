// Shared device helpers for the vt355 HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define VT_OK 0
#define VT_ERR_BAD_SHAPE (-1)
#define VT_ERR_BAD_ALIGN (-2)
#define VT_ERR_LAUNCH (-3)
#define VT_ERR_UNSUPPORTED (-4)

#define WAVE 64

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __uint_as_float(((unsigned int)b) << 16);
}
__device__ __forceinline__ float bf2f(bf16_t x) { return (float)x; }
__device__ __forceinline__ bf16_t f2bf(float x) { return (bf16_t)x; }

// unpack 8 bf16 (one 16-byte chunk held as 4 dwords) into 8 floats
__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(v[i] << 16);
        f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ unsigned int pack2(float lo, float hi) {
    bf16x2 p;
    p[0] = (bf16_t)lo;
    p[1] = (bf16_t)hi;
    return __builtin_bit_cast(unsigned int, p);
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack2(f[2 * i], f[2 * i + 1]);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// tanh-GELU as x * sigmoid(2u), u = k0 (x + k1 x^3): sigmoid(2u) = 1 - 1 / (2^(2 log2e u) + 1) with the hardware exp2 / rcp
// (no IEEE division: these run in GEMM epilogues while the matrix pipe waits).  Saturates correctly: 2^w = inf -> s = 1,
// 2^w = 0 -> s = 0.
__device__ __forceinline__ float gelu_sigmoid_f(float x, float x2) {
    const float c0 = 2.0f * 1.4426950408889634f * 0.7978845608028654f, c1 = c0 * 0.044715f;
    const float w = x * __builtin_fmaf(c1, x2, c0);
    return 1.0f - __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(w) + 1.0f);
}
__device__ __forceinline__ float gelu_tanh_f(float x) { return x * gelu_sigmoid_f(x, x * x); }
// d/dx [x s(x)] = s + x s (1 - s) * 2 du/dx,  du/dx = k0 (1 + 3 k1 x^2)
__device__ __forceinline__ float gelu_tanh_grad_f(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float x2 = x * x;
    const float sg = gelu_sigmoid_f(x, x2);
    const float du2 = __builtin_fmaf(6.0f * k0 * k1, x2, 2.0f * k0);
    return __builtin_fmaf(x * du2, sg * (1.0f - sg), sg);
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// buffer resource for bounds-checked (OOB reads return 0, OOB stores dropped) raw buffer ops
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// Bijective XCD-aware remap of a linear workgroup id (8 XCDs, round-robin dispatch): consecutive
// logical ids land on the same XCD so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int nx = 8;
    int q = nwg / nx, r = nwg % nx;
    int xcd = orig % nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + orig / nx;
}
// rotary position embedding helpers (fp32 tables [S - St, 64]; pairs (2i, 2i+1) live in one lane's 8 elements)
__device__ __forceinline__ void rope_load8(const float* src, float* d) {
    f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { d[j] = a[j]; d[j + 4] = b[j]; }
}
// transpose of the rotation: applied to an incoming gradient
__device__ __forceinline__ void rope_bwd8(const float* rope_cos, const float* rope_sin, int spos, int sub, float* dy) {
    float cs[8], sn[8];
    rope_load8(rope_cos + (size_t)spos * 64 + sub * 8, cs);
    rope_load8(rope_sin + (size_t)spos * 64 + sub * 8, sn);
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const float a = dy[j], b = dy[j + 1];
        dy[j] = a * cs[j] + b * sn[j + 1];
        dy[j + 1] = b * cs[j + 1] - a * sn[j];
    }
}

