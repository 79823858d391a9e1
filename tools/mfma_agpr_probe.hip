// Does a dependent v_mfma_f32_32x32x64_f8f6f4 issue sooner when its accumulator lives in AccVGPRs?  One chain per wave (srcC = vDst),
// VGPR accumulator (compiler) against AGPR accumulator (inline assembly, "a" constraint), 1 - 4 waves per SIMD; no trees, no LDS.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_agpr_probe mfma_agpr_probe.hip && ./mfma_agpr_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(1024) void probe(int iters, float* out) {
    const uint32_t h = (threadIdx.x * 2654435761u) & 0x22222222u;
    const v4i a4 = {(int)h, (int)(h >> 1 & 0x22222222u), (int)h, (int)h}, b4 = {(int)(h >> 2 & 0x22222222u), (int)h, (int)h, (int)h};
    v16f c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v16f d = c;
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it)
            asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4" : "+v"(c) : "v"(a4), "v"(b4));
    } else if (MODE == 1) {
        for (int it = 0; it < iters; ++it)
            asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4" : "+a"(c) : "v"(a4), "v"(b4));
    } else if (MODE == 2) {          // two independent chains, VGPR
        for (int it = 0; it < iters; it += 2) {
            asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4" : "+v"(c) : "v"(a4), "v"(b4));
            asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4" : "+v"(d) : "v"(a4), "v"(b4));
        }
    } else {                         // two independent chains, AGPR
        for (int it = 0; it < iters; it += 2) {
            asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4" : "+a"(c) : "v"(a4), "v"(b4));
            asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4" : "+a"(d) : "v"(a4), "v"(b4));
        }
    }
    if (c[0] + d[3] == 12345.f) out[threadIdx.x] = c[1];
}
template <int MODE> static void run(const char* name, int wps, float* d) {
    const int iters = 200000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<MODE><<<256, 256 * wps>>>(1000, d);
    (void)hipEventRecord(e0); probe<MODE><<<256, 256 * wps>>>(iters, d); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s %d waves/SIMD: %8.3f ms  %6.1f ns per MFMA and wave, %6.1f per MFMA and SIMD\n", name, wps, ms, ms * 1e6 / iters, ms * 1e6 / iters / wps);
}
int main() {
    float* d; (void)hipMalloc(&d, 4096 * 4);
    for (int wps = 1; wps <= 4; ++wps) { run<0>("1 chain, VGPR acc", wps, d); run<1>("1 chain, AGPR acc", wps, d); run<2>("2 chains, VGPR acc", wps, d); run<3>("2 chains, AGPR acc", wps, d); }
    return 0;
}
