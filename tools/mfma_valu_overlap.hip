// Does VALU work hide under FP4 MFMAs on one SIMD?  Per loop iteration: 8 x v_mfma_scale_f32_32x32x64_f8f6f4
// (two accumulator chains, as in knn_mfma_kernel) + K independent v_max3_i32.  Prints cycles per iteration per
// SIMD for K = 0..64 at 1, 2 and 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int K, bool MFMA>
__global__ __launch_bounds__(256) void probe(int iters, int* out, unsigned long long* cyc) {
    v8i a = {(int)threadIdx.x, 1, 2, 3, 0, 0, 0, 0}, b = {5, 6, 7, (int)threadIdx.x, 0, 0, 0, 0};
    v16f c0 = {0}, c1 = {0};
    int x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * (i + 1);
    const int y = blockIdx.x, z = threadIdx.x ^ 5;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MFMA) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, c1, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k)
            asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x[k & 7]) : "v"(y), "v"(z));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    int acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += x[i];
    for (int i = 0; i < 16; ++i) acc += (int)c0[i] + (int)c1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) atomicAdd(cyc, t1 - t0);
}

template <int K, bool MFMA>
void run(int waves_per_simd, int* out, unsigned long long* cyc) {
    const int iters = 20000;
    const int blocks = 256 * waves_per_simd;          // 256-thread blocks: one wave per SIMD each
    hipMemset(cyc, 0, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<K, MFMA><<<blocks, 256>>>(iters, out, cyc);       // warm-up (clocks)
    hipMemset(cyc, 0, 8);
    hipEventRecord(e0);
    probe<K, MFMA><<<blocks, 256>>>(iters, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h = 0;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("  [wall %.3f ms = %.1f ns per wave-iteration; tick rate %.2f GHz]", ms, ms * 1e6 / iters, (double)h / blocks / (ms * 1e6));
    // wave-cycles per iteration divided by resident waves per SIMD = SIMD cycles per (one iteration of every resident wave) / waves
    printf("  K=%2d mfma=%d waves/SIMD=%d: %.1f cycles per wave-iteration, %.1f SIMD-cycles per iteration\n", K, (int)MFMA, waves_per_simd,
           (double)h / blocks / iters, (double)h / blocks / iters / waves_per_simd);
}

int main() {
    int* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&cyc, 8);
    for (int w : {1, 2, 4}) {
        run<0, true>(w, out, cyc); run<16, true>(w, out, cyc); run<32, true>(w, out, cyc); run<48, true>(w, out, cyc); run<64, true>(w, out, cyc);
        run<32, false>(w, out, cyc); run<64, false>(w, out, cyc);
    }
    return 0;
}
