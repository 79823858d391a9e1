// Where do the skewed chains of the search lose their issue slots?  A dependent v_mfma_f32_32x32x64_f8f6f4 issued RIGHT behind its predecessor
// goes every ~39 cycles (tools/mfma_agpr_probe.hip); the search interleaves two VALU instructions of the other group's max tree after every
// MFMA of a chain (KT_INTERLEAVE).  The same two-accumulator loop as mfma_chain_probe's 'skew', fragments prefetched one iteration ahead
// (as the kernel does), with the instruction ORDER of a chain + tree forced three ways:
//   il    1 MFMA, 2 VALU, four times      (the kernel's)
//   b2b   4 MFMAs back to back, then the VALU
//   pre   the VALU first, then 4 MFMAs back to back
//   free  the compiler's own order
//   hipcc --offload-arch=gfx950 -O3 -o mfma_order_probe mfma_order_probe.hip && ./mfma_order_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define MF(acc, f, b) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v8i{(int)(f).x, (int)(f).y, (int)(f).z, (int)(f).w, 0, 0, 0, 0}, b, acc, 4, 4, 0, 0, 0, 0)
__device__ __forceinline__ int tree(const v16f& a) {
    int t[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) t[k] = max(max(__float_as_int(a[3 * k]), __float_as_int(a[3 * k + 1])), __float_as_int(a[3 * k + 2]));
    return max(max(max(__float_as_int(a[15]), t[0]), t[1]), max(max(t[2], t[3]), t[4]));
}
template <int MODE> __device__ __forceinline__ void order() {
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
    } else if (MODE == 1) { __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x002, 8, 0); }
    else if (MODE == 2) { __builtin_amdgcn_sched_group_barrier(0x002, 8, 0); __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); }
}
template <int MODE>
__global__ __launch_bounds__(1024) void probe(int iters, int* out) {
    __shared__ uint4 lds[8][4][64];
    const int lane = threadIdx.x & 63;
    v8i b[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 4; ++s) { const uint32_t h = (threadIdx.x * 2654435761u + i * 97u + s) & 0x22222222u; b[i][s] = v8i{(int)h, (int)(h >> 1 & 0x22222222u), (int)h, (int)h, 0, 0, 0, 0}; }
    if (threadIdx.x < 64) for (int r = 0; r < 8; ++r) for (int s = 0; s < 4; ++s) lds[r][s][lane] = make_uint4(0x20202020u * (lane & 1), 0x02020202u, 0x22002200u, 0x00220022u * (r & 1));
    __syncthreads();
    const v16f zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v16f c0 = zero, c1 = zero;
    int thr = 1 << 30, hits = 0;
    const uint4* L = &lds[0][0][lane];
    uint4 f0 = L[0], f1 = L[64], f2 = L[128], f3 = L[192];
    for (int it = 0; it < iters; ++it) {
        const uint4* Ln = &lds[(it + 1) & 7][0][lane];
        const uint4 g0 = Ln[0], g1 = Ln[64], g2 = Ln[128], g3 = Ln[192];               // next iteration's fragments, used a whole iteration later
        if (MODE == 4) {        // chains only: what two alternating chains of four cost with nothing else in the wave
            c1 = MF(zero, f0, b[1][0]); c1 = MF(c1, f1, b[1][1]); c1 = MF(c1, f2, b[1][2]); c1 = MF(c1, f3, b[1][3]);
            c0 = MF(zero, f3, b[0][0]); c0 = MF(c0, f2, b[0][1]); c0 = MF(c0, f1, b[0][2]); c0 = MF(c0, f0, b[0][3]);
            asm volatile("" : "+v"(c0), "+v"(c1));
            f0 = g0; f1 = g1; f2 = g2; f3 = g3;
            continue;
        }
        if (MODE == 5) {        // trees, but their results only folded into a register (no compare, no ballot)
            const int m0 = tree(c0);
            c1 = MF(zero, f0, b[1][0]); c1 = MF(c1, f1, b[1][1]); c1 = MF(c1, f2, b[1][2]); c1 = MF(c1, f3, b[1][3]);
            order<0>();
            asm volatile("" : "+v"(c1));
            const int m1 = tree(c1);
            c0 = MF(zero, f3, b[0][0]); c0 = MF(c0, f2, b[0][1]); c0 = MF(c0, f1, b[0][2]); c0 = MF(c0, f0, b[0][3]);
            order<0>();
            asm volatile("" : "+v"(c0));
            hits = max(hits, max(m0, m1));
            f0 = g0; f1 = g1; f2 = g2; f3 = g3;
            continue;
        }
        if (MODE == 6) {        // the tree of a chain issued behind the SECOND MFMA of the next chain (its last MFMA is then two slots old)
            const int m0 = tree(c0);
            c1 = MF(zero, f0, b[1][0]); c1 = MF(c1, f1, b[1][1]); c1 = MF(c1, f2, b[1][2]); c1 = MF(c1, f3, b[1][3]);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            asm volatile("" : "+v"(c1));
            if (__builtin_amdgcn_ballot_w64(m0 > thr)) ++hits;
            const int m1 = tree(c1);
            c0 = MF(zero, f3, b[0][0]); c0 = MF(c0, f2, b[0][1]); c0 = MF(c0, f1, b[0][2]); c0 = MF(c0, f0, b[0][3]);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            asm volatile("" : "+v"(c0));
            if (__builtin_amdgcn_ballot_w64(m1 > thr)) ++hits;
            f0 = g0; f1 = g1; f2 = g2; f3 = g3;
            continue;
        }
        {
            const int m0 = tree(c0);
            c1 = MF(zero, f0, b[1][0]); c1 = MF(c1, f1, b[1][1]); c1 = MF(c1, f2, b[1][2]); c1 = MF(c1, f3, b[1][3]);
            order<MODE>();
            asm volatile("" : "+v"(c1));
            if (__builtin_amdgcn_ballot_w64(m0 > thr)) ++hits;
        }
        {
            const int m1 = tree(c1);
            c0 = MF(zero, f3, b[0][0]); c0 = MF(c0, f2, b[0][1]); c0 = MF(c0, f1, b[0][2]); c0 = MF(c0, f0, b[0][3]);
            order<MODE>();
            asm volatile("" : "+v"(c0));
            if (__builtin_amdgcn_ballot_w64(m1 > thr)) ++hits;
        }
        f0 = g0; f1 = g1; f2 = g2; f3 = g3;
    }
    if (hits == 12345) out[threadIdx.x] = hits + tree(c0) + tree(c1);
}
template <int MODE> static void run(const char* name, int wps, int* d) {
    const int iters = 20000;                                                   // 8 MFMAs per iteration
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<MODE><<<256, 256 * wps>>>(100, d);
    (void)hipEventRecord(e0); probe<MODE><<<256, 256 * wps>>>(iters, d); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mf = 8.0 * iters * wps;
    printf("%-5s %d waves/SIMD: %8.3f ms  %6.1f ns per MFMA and SIMD  (%.2f PFLOP/s)\n", name, wps, ms, ms * 1e6 / mf, mf * 1024 * 131072.0 / (ms * 1e-3) / 1e15);
}
int main() {
    int* d; (void)hipMalloc(&d, 4096 * 4);
    for (int wps = 1; wps <= 4; ++wps) { run<0>("il", wps, d); run<1>("b2b", wps, d); run<2>("pre", wps, d); run<3>("free", wps, d); run<4>("chain", wps, d); run<5>("notst", wps, d); run<6>("late", wps, d); }
    return 0;
}
