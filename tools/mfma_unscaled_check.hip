// Is v_mfma_f32_32x32x64_f8f6f4 (what the compiler selects when both scale operands of the builtin are the constant 0) the same
// function as v_mfma_scale_f32_32x32x64_f8f6f4 with unit scales (E8M0 0x7F) on {0,1} FP4 operands?  Every accumulator value of
// both is compared bit for bit, over random operands and random integer-valued accumulators (the search's: dot <= 256).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_unscaled_check mfma_unscaled_check.hip && ./mfma_unscaled_check
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void both(const uint4* a, const uint4* b, const float* c, float* out_s, float* out_u) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    const uint4 x = a[t], y = b[t];
    const v8i va = {(int)x.x, (int)x.y, (int)x.z, (int)x.w, 0, 0, 0, 0}, vb = {(int)y.x, (int)y.y, (int)y.z, (int)y.w, 0, 0, 0, 0};
    v16f acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = c[t * 16 + i];
    const v16f s = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    const v16f u = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc, 4, 4, 0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) { out_s[t * 16 + i] = s[i]; out_u[t * 16 + i] = u[i]; }
}
int main() {
    const int nb = 4096, n = nb * 64;
    uint4 *ha = (uint4*)malloc(n * 16), *hb = (uint4*)malloc(n * 16);
    float *hc = (float*)malloc(n * 64), *hs = (float*)malloc(n * 64), *hu = (float*)malloc(n * 64);
    srand(12345);
    auto nib = [](int density) { uint32_t w = 0; for (int k = 0; k < 8; ++k) if (rand() % 100 < density) w |= 2u << (4 * k); return w; };
    for (int i = 0; i < n; ++i) {
        const int d = (i / 64) % 3 == 0 ? 50 : (i / 64) % 3 == 1 ? 95 : 5;
        ha[i] = make_uint4(nib(d), nib(d), nib(d), nib(d)); hb[i] = make_uint4(nib(d), nib(d), nib(d), nib(d));
        for (int k = 0; k < 16; ++k) hc[i * 16 + k] = (float)(rand() % 193);
    }
    uint4 *da, *db; float *dc, *ds, *du;
    (void)hipMalloc(&da, n * 16); (void)hipMalloc(&db, n * 16); (void)hipMalloc(&dc, n * 64); (void)hipMalloc(&ds, n * 64); (void)hipMalloc(&du, n * 64);
    (void)hipMemcpy(da, ha, n * 16, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, n * 16, hipMemcpyHostToDevice); (void)hipMemcpy(dc, hc, n * 64, hipMemcpyHostToDevice);
    both<<<nb, 64>>>(da, db, dc, ds, du);
    (void)hipMemcpy(hs, ds, n * 64, hipMemcpyDeviceToHost); (void)hipMemcpy(hu, du, n * 64, hipMemcpyDeviceToHost);
    long bad = 0, nonzero = 0; double mx = 0;
    for (long i = 0; i < (long)n * 16; ++i) {
        uint32_t p, q; __builtin_memcpy(&p, &hs[i], 4); __builtin_memcpy(&q, &hu[i], 4);
        if (p != q) { if (bad < 5) printf("mismatch at %ld: scaled %g unscaled %g (c %g)\n", i, hs[i], hu[i], hc[i]); ++bad; }
        if (hs[i] != hc[i]) ++nonzero;
        if (hs[i] > mx) mx = hs[i];
    }
    printf("unscaled vs unit-scaled v_mfma 32x32x64 f8f6f4 (FP4 {0,1} operands): %ld values, %ld changed by the product, max %g, MISMATCHES %ld\n", (long)n * 16, nonzero, mx, bad);
    return bad != 0;
}
