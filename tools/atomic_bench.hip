// Micro-benchmark: where do fp32 global atomics execute on gfx950 and at what rate?
// Decides whether attn_bwd's dQ accumulation (9.45 GB of atomic payload per sample) can be made cheaper by locality.
//   pattern 0: "dq"      -- 4200 WGs, WG (slice = id/70) walks its 4.55 MB slice once, 64 rows x 256 B per step (the real pattern)
//   pattern 1: "private" -- each WG re-adds into its own 16 KB (no sharing, L2 resident)
//   pattern 2: "xcd"     -- all WGs of an XCD share one 1 MB window, re-added (shared inside one L2 only)
//   pattern 3: "chip"    -- all WGs of the chip share one 1 MB window
// aux: 0 (sc1=0) or 16 (sc1=1).   usage: atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((__vector_size__(4 * sizeof(int)))) int i32x4_t;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int nx = 8; int q = nwg / nx, r = nwg % nx; int xcd = orig % nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + orig / nx;
}

template <int AUX>
__global__ __launch_bounds__(256) void k(float* buf, int pattern, int steps, long long slice_floats, int per_slice) {
    const int tid = threadIdx.x;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    float* base; long long win;      // window in floats
    if (pattern == 0) { base = buf + (long long)(id / per_slice) * slice_floats; win = slice_floats; }
    else if (pattern == 1) { base = buf + (long long)id * 4096; win = 4096; }
    else if (pattern == 2) { base = buf + (long long)(blockIdx.x % 8) * 262144; win = 262144; }
    else { base = buf; win = 262144; }
    __amdgpu_buffer_rsrc_t r = mk(base, (unsigned)(win * 4));
    // one step = 64 rows x 64 floats (16 KB): 4 waves x 16 atomics x 64 lanes x 4 B
    const int lane = tid & 63, w = tid >> 6;
    for (int t = 0; t < steps; ++t) {
        const int soff = (int)(((long long)t * 4096) % win) * 4;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(1.0f, r, ((w * 16 + i) * 64 + lane) * 4, soff, AUX);
    }
}

int main() {
    const long long slice = 17776LL * 64;          // floats per (b, head) dQ slice
    const int nsl = 60, per = 70;
    const size_t bytes = (size_t)slice * nsl * 4 + (1 << 26);
    float* d; hipMalloc(&d, bytes); hipMemset(d, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[4] = {"dq-walk", "private16K", "xcd-1MB", "chip-1MB"};
    for (int aux = 0; aux <= 16; aux += 16)
        for (int pat = 0; pat < 4; ++pat) {
            const int grid = nsl * per;
            const int steps = pat == 0 ? (int)(slice / 4096) : 256;
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (aux == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d, pat, steps, slice, per);
                else hipLaunchKernelGGL(k<16>, dim3(grid), dim3(256), 0, 0, d, pat, steps, slice, per);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double payload = (double)grid * steps * 16384.0;
            printf("aux=%2d %-11s grid %d steps %d: %.3f ms  %.1f GB/s atomic payload\n", aux, names[pat], grid, steps, best, payload / best / 1e6);
            fflush(stdout);
        }
    // per-CU or chip-wide limit?  the same private pattern from fewer workgroups (1 WG per CU up to 256)
    for (int grid : {8, 32, 64, 128, 256, 512, 1024}) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d, 1, 2048, slice, per);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("private16K grid %4d: %.3f ms  %.1f GB/s  (%.2f GB/s per WG)\n", grid, best, grid * 2048.0 * 16384 / best / 1e6,
               2048.0 * 16384 / best / 1e6);
    }
    // correctness of cross-XCD sharing with aux=0: chip-shared window, every element must equal grid*steps*(16KB/1MB share)
    hipMemset(d, 0, 1 << 20);
    hipLaunchKernelGGL(k<0>, dim3(4200), dim3(256), 0, 0, d, 3, 64, slice, per);
    std::vector<float> h(262144); hipMemcpy(h.data(), d, 1 << 20, hipMemcpyDeviceToHost);
    double mn = 1e30, mx = -1; for (float v : h) { if (v < mn) mn = v; if (v > mx) mx = v; }
    printf("chip-shared aux=0 check: every element should be %d: min %.0f max %.0f\n", 4200 * 64 * 4096 / 262144, mn, mx);
    return 0;
}
