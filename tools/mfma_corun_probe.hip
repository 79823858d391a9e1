// What slows a matrix-core stream down when other kernels share its CUs?  The `skew` wave shape of mfma_chain_probe.hip (= the search
// kernel's: 2 waves per SIMD, one 512-thread block per CU, 88 KB of LDS so that no second one fits) runs alone and beside one of
// four synthetic co-runners on a second stream, each filling what the first leaves (4 blocks of 256 threads per CU, 48 registers):
//   valu   register-only integer / f32 arithmetic            lds     conflict-free ds_read_b32 streams
//   ldsx   ds_read_b32 with 8-way bank conflicts             mem     global streaming reads (HBM / L2 bound)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_corun_probe mfma_corun_probe.hip && ./mfma_corun_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define MF(acc, f, b) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v8i{(int)(f).x, (int)(f).y, (int)(f).z, (int)(f).w, 0, 0, 0, 0}, b, acc, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F)
__device__ __forceinline__ int tree(const v16f& a) {
    int t[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) t[k] = max(max(__float_as_int(a[3 * k]), __float_as_int(a[3 * k + 1])), __float_as_int(a[3 * k + 2]));
    return max(max(max(__float_as_int(a[15]), t[0]), t[1]), max(max(t[2], t[3]), t[4]));
}
template <int PRIO>
__global__ __launch_bounds__(512) void mfma_stream(int iters, int* out) {
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    extern __shared__ uint4 dyn[];                                  // + 56 KB dynamic: 88 KB in all
    __shared__ uint4 lds[8][4][64];
    const int lane = threadIdx.x & 63;
    v8i b[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 4; ++s) { const uint32_t h = (threadIdx.x * 2654435761u + i * 97u + s) & 0x22222222u; b[i][s] = v8i{(int)h, (int)(h >> 1 & 0x22222222u), (int)h, (int)h, 0, 0, 0, 0}; }
    if (threadIdx.x < 64) for (int r = 0; r < 8; ++r) for (int s = 0; s < 4; ++s) lds[r][s][lane] = make_uint4(0x20202020u * (lane & 1), 0x02020202u, 0x22002200u, 0x00220022u * (r & 1));
    if (iters < 0) dyn[threadIdx.x] = lds[0][0][lane];
    __syncthreads();
    const v16f zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v16f c[2] = {zero, zero};
    int thr = 1 << 30, hits = 0;
    for (int it = 0; it < iters; ++it) {
        const uint4* L = &lds[it & 7][0][lane];
        const uint4 f0 = L[0], f1 = L[64], f2 = L[128], f3 = L[192];
        const int m0 = tree(c[0]);
        c[1] = MF(zero, f0, b[1][0]); c[1] = MF(c[1], f1, b[1][1]); c[1] = MF(c[1], f2, b[1][2]); c[1] = MF(c[1], f3, b[1][3]);
        if (__builtin_amdgcn_ballot_w64(m0 > thr)) ++hits;
        const int m1 = tree(c[1]);
        c[0] = MF(zero, f3, b[0][0]); c[0] = MF(c[0], f2, b[0][1]); c[0] = MF(c[0], f1, b[0][2]); c[0] = MF(c[0], f0, b[0][3]);
        if (__builtin_amdgcn_ballot_w64(m1 > thr)) ++hits;
    }
    if (hits == 12345) out[threadIdx.x] = hits + tree(c[0]) + tree(c[1]);
}
template <int MODE>
__global__ __launch_bounds__(256) void hog(int iters, const uint4* __restrict__ src, size_t n16, int* out) {
    __shared__ uint32_t s[4096];                                    // 16 KB
    for (int i = threadIdx.x; i < 4096; i += 256) s[i] = i * 2654435761u;
    __syncthreads();
    uint32_t a = threadIdx.x, bq = blockIdx.x * 7 + 1, c = 12345, d = 99;
    float f = 1.0f + threadIdx.x, g = 0.5f;
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) { a = a * bq + c; f = f * g + 1.0f; c ^= a >> 3; g = g * 0.999f + 0.001f; d += c & a; }
        }
    } else if (MODE == 1 || MODE == 2) {
        const int stride = MODE == 1 ? 1 : 8;                       // 8 dwords: 8-way conflicts on 32 banks... (64 lanes, 2 passes)
        uint32_t idx = (threadIdx.x * stride) & 4095;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) { a += s[(idx + u * 256 * stride) & 4095]; }
            idx = (idx + (a & 1)) & 4095;
        }
    } else {
        size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { const uint4 v = src[i % n16]; a ^= v.x + v.y + v.z + v.w; i += (size_t)gridDim.x * 256; }
        }
    }
    if (a + c + d == 42 && f + g == 1.5f) out[threadIdx.x] = a;
}
template <int PRIO>
static float run_pair(int mode, hipStream_t sa, hipStream_t sb, int* d, const uint4* src, size_t n16) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int hog_iters = mode == 0 ? 600000 : mode == 1 || mode == 2 ? 400000 : 60000;
    if (mode >= 0) {
        if (mode == 0) hog<0><<<1024, 256, 0, sb>>>(hog_iters, src, n16, d);
        if (mode == 1) hog<1><<<1024, 256, 0, sb>>>(hog_iters, src, n16, d);
        if (mode == 2) hog<2><<<1024, 256, 0, sb>>>(hog_iters, src, n16, d);
        if (mode == 3) hog<3><<<1024, 256, 0, sb>>>(hog_iters, src, n16, d);
    }
    hipEventRecord(e0, sa);
    mfma_stream<PRIO><<<256, 512, 56 * 1024, sa>>>(20000, d);
    hipEventRecord(e1, sa); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEvent_t h1; hipEventCreate(&h1); hipEventRecord(h1, sb); hipEventSynchronize(h1);
    float hm; hipEventElapsedTime(&hm, e0, h1);
    printf("  (co-runner still running %.1f ms after the stream's start)\n", hm);
    return ms;
}
int main() {
    int* d; hipMalloc(&d, 4096 * 4);
    const size_t n16 = (size_t)1 << 26;                            // 1 GiB
    uint4* src; hipMalloc(&src, n16 * 16); hipMemset(src, 1, n16 * 16);
    hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    mfma_stream<0><<<256, 512, 56 * 1024, sa>>>(100, d); hipDeviceSynchronize();
    const char* names[] = {"alone", "valu", "lds", "ldsx", "mem"};
    for (int rep = 0; rep < 2; ++rep)
        for (int m = -1; m <= 3; ++m) {
            const float ms = rep ? run_pair<3>(m, sa, sb, d, src, n16) : run_pair<0>(m, sa, sb, d, src, n16);
            printf("prio %d ", rep ? 3 : 0);
            hipDeviceSynchronize();
            printf("%-6s matrix-core stream %8.3f ms  (%.2f PFLOP/s)\n", names[m + 1], ms, 40000.0 * 4 * 2 * 1024 * 131072.0 / (ms * 1e-3) / 1e15);
        }
    return 0;
}
