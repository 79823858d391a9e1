// How many waves per SIMD does v_mfma_scale_f32_32x32x64_f8f6f4 need to stay busy, by the SHAPE of a wave's instruction stream?
//   dep    one accumulator: chain of 4 dependent MFMAs, then its max tree (6 VALU), repeat
//   skew   two accumulators: the chain of one while the tree of the other is issued (knn_tile_body<2>, G = 1)
//   pair   four accumulators: two interleaved chains while the trees of the other two are issued (knn_tile_body<4>, G = 2)
//   flat2  two accumulators: two interleaved chains on the same fragments, then both trees (nothing overlapped inside the wave)
//   split  two accumulators, the two chains offset by HALF a chain and interleaved (r06): A.k2 [tree B] B.k0 A.k3 B.k1 | B.k2 [tree A] A'.k0 B.k3 A'.k1 —
//          consecutive MFMAs of one chain are two issue slots apart (a dependent MFMA issues ~80 cycles after its predecessor, a slot is 32)
// A operands re-read from LDS for every chain ({0,1} nibbles), B operands in registers.  Grid = 256 blocks of W x 4 waves.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_chain_probe mfma_chain_probe.hip && ./mfma_chain_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#ifdef MF_UNSCALED      /* scale operands 0: the compiler selects v_mfma_f32_32x32x64_f8f6f4 (no scale load, 64-bit encoding) */
#define MF_SCALE 0
#else
#define MF_SCALE 0x7F7F7F7F
#endif
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define MF(acc, f, b) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v8i{(int)(f).x, (int)(f).y, (int)(f).z, (int)(f).w, 0, 0, 0, 0}, b, acc, 4, 4, 0, MF_SCALE, 0, MF_SCALE)
__device__ __forceinline__ int tree(const v16f& a) {
    int t[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) t[k] = max(max(__float_as_int(a[3 * k]), __float_as_int(a[3 * k + 1])), __float_as_int(a[3 * k + 2]));
    return max(max(max(__float_as_int(a[15]), t[0]), t[1]), max(max(t[2], t[3]), t[4]));
}
template <int MODE>
__global__ __launch_bounds__(1024) void probe(int iters, int* out) {
    __shared__ uint4 lds[8][4][64];
    const int lane = threadIdx.x & 63;
    v8i b[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < 4; ++s) { const uint32_t h = (threadIdx.x * 2654435761u + i * 97u + s) & 0x22222222u; b[i][s] = v8i{(int)h, (int)(h >> 1 & 0x22222222u), (int)h, (int)h, 0, 0, 0, 0}; }
    if (threadIdx.x < 64) for (int r = 0; r < 8; ++r) for (int s = 0; s < 4; ++s) lds[r][s][lane] = make_uint4(0x20202020u * (lane & 1), 0x02020202u, 0x22002200u, 0x00220022u * (r & 1));
    __syncthreads();
    const v16f zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v16f c[4] = {zero, zero, zero, zero};
    int thr = 1 << 30, hits = 0;
    for (int it = 0; it < iters; ++it) {
        const uint4* L = &lds[it & 7][0][lane];
        const uint4 f0 = L[0], f1 = L[64], f2 = L[128], f3 = L[192];
        if (MODE == 0) {
            c[0] = MF(zero, f0, b[0][0]); c[0] = MF(c[0], f1, b[0][1]); c[0] = MF(c[0], f2, b[0][2]); c[0] = MF(c[0], f3, b[0][3]);
            if (__builtin_amdgcn_ballot_w64(tree(c[0]) > thr)) ++hits;
        } else if (MODE == 1) {
            // chain into c[1] while the tree of c[0] (previous iteration's chain) is issued, then the other way round
            const int m0 = tree(c[0]);
            c[1] = MF(zero, f0, b[1][0]); c[1] = MF(c[1], f1, b[1][1]); c[1] = MF(c[1], f2, b[1][2]); c[1] = MF(c[1], f3, b[1][3]);
            if (__builtin_amdgcn_ballot_w64(m0 > thr)) ++hits;
            const int m1 = tree(c[1]);
            c[0] = MF(zero, f3, b[0][0]); c[0] = MF(c[0], f2, b[0][1]); c[0] = MF(c[0], f1, b[0][2]); c[0] = MF(c[0], f0, b[0][3]);
            if (__builtin_amdgcn_ballot_w64(m1 > thr)) ++hits;
        } else if (MODE == 4) {
            // c[0] = chain A with k-steps 0, 1 of this tile done; c[1] = chain B of the previous tile, complete
            c[0] = MF(c[0], f2, b[0][2]);
            const int mB = tree(c[1]);
            if (__builtin_amdgcn_ballot_w64(mB > thr)) ++hits;
            c[1] = MF(zero, f0, b[1][0]);
            c[0] = MF(c[0], f3, b[0][3]);
            c[1] = MF(c[1], f1, b[1][1]);
            c[1] = MF(c[1], f2, b[1][2]);
            const int mA = tree(c[0]);
            if (__builtin_amdgcn_ballot_w64(mA > thr)) ++hits;
            c[0] = MF(zero, f1, b[0][0]);
            c[1] = MF(c[1], f3, b[1][3]);
            c[0] = MF(c[0], f0, b[0][1]);
        } else if (MODE == 3) {
            // two interleaved chains on the SAME fragments, then both trees: no overlap inside the wave (the SIMD's other waves cover)
            c[0] = MF(zero, f0, b[0][0]); c[1] = MF(zero, f0, b[1][0]); c[0] = MF(c[0], f1, b[0][1]); c[1] = MF(c[1], f1, b[1][1]);
            c[0] = MF(c[0], f2, b[0][2]); c[1] = MF(c[1], f2, b[1][2]); c[0] = MF(c[0], f3, b[0][3]); c[1] = MF(c[1], f3, b[1][3]);
            const int m0 = tree(c[0]), m1 = tree(c[1]);
            if (__builtin_amdgcn_ballot_w64(m0 > thr)) ++hits;
            if (__builtin_amdgcn_ballot_w64(m1 > thr)) ++hits;
        } else {
            const int m0 = tree(c[0]), m1 = tree(c[1]);
            c[2] = MF(zero, f0, b[2][0]); c[3] = MF(zero, f0, b[3][0]); c[2] = MF(c[2], f1, b[2][1]); c[3] = MF(c[3], f1, b[3][1]);
            c[2] = MF(c[2], f2, b[2][2]); c[3] = MF(c[3], f2, b[3][2]); c[2] = MF(c[2], f3, b[2][3]); c[3] = MF(c[3], f3, b[3][3]);
            if (__builtin_amdgcn_ballot_w64(max(m0, m1) > thr)) ++hits;
            const int m2 = tree(c[2]), m3 = tree(c[3]);
            c[0] = MF(zero, f3, b[0][0]); c[1] = MF(zero, f3, b[1][0]); c[0] = MF(c[0], f2, b[0][1]); c[1] = MF(c[1], f2, b[1][1]);
            c[0] = MF(c[0], f1, b[0][2]); c[1] = MF(c[1], f1, b[1][2]); c[0] = MF(c[0], f0, b[0][3]); c[1] = MF(c[1], f0, b[1][3]);
            if (__builtin_amdgcn_ballot_w64(max(m2, m3) > thr)) ++hits;
        }
    }
    if (hits == 12345) out[threadIdx.x] = hits + tree(c[0]) + tree(c[1]) + tree(c[2]) + tree(c[3]);
}

// small16: v_mfma_scale_f32_16x16x128_f8f6f4 — 16 x 16 outputs, 4 accumulator registers: four query tiles of 16 per wave (the same 64 queries),
// a train tile of 16 rows; per iteration 8 MFMAs of 4 passes (= the work of four 32x32x64) in four independent chains of two, the trees
// of the other accumulator set (1 max3 + 1 max each) underneath.
typedef float v4f __attribute__((ext_vector_type(4)));
#define MF16(acc, f, b) __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(v8i{(int)(f).x, (int)(f).y, (int)(f).z, (int)(f).w, 0, 0, 0, 0}, b, acc, 4, 4, 0, MF_SCALE, 0, MF_SCALE)
__device__ __forceinline__ int tree4(const v4f& a) { return max(max(max(__float_as_int(a[0]), __float_as_int(a[1])), __float_as_int(a[2])), __float_as_int(a[3])); }
__global__ __launch_bounds__(1024) void probe16(int iters, int* out) {
    __shared__ uint4 lds[8][2][64];
    const int lane = threadIdx.x & 63;
    v8i b[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) { const uint32_t h = (threadIdx.x * 2654435761u + i * 97u + s) & 0x22222222u; b[i][s] = v8i{(int)h, (int)(h >> 1 & 0x22222222u), (int)h, (int)h, 0, 0, 0, 0}; }
    if (threadIdx.x < 64) for (int r = 0; r < 8; ++r) for (int s = 0; s < 2; ++s) lds[r][s][lane] = make_uint4(0x20202020u * (lane & 1), 0x02020202u, 0x22002200u, 0x00220022u * (r & 1));
    __syncthreads();
    const v4f zero = {0, 0, 0, 0};
    v4f c[8] = {zero, zero, zero, zero, zero, zero, zero, zero};
    int thr = 1 << 30, hits = 0;
    for (int it = 0; it < iters; ++it) {
        const uint4* L = &lds[it & 7][0][lane];
        const uint4 f0 = L[0], f1 = L[64];
        const int m0 = max(max(tree4(c[0]), tree4(c[1])), max(tree4(c[2]), tree4(c[3])));
        c[4] = MF16(zero, f0, b[0][0]); c[5] = MF16(zero, f0, b[1][0]); c[6] = MF16(zero, f0, b[2][0]); c[7] = MF16(zero, f0, b[3][0]);
        c[4] = MF16(c[4], f1, b[0][1]); c[5] = MF16(c[5], f1, b[1][1]); c[6] = MF16(c[6], f1, b[2][1]); c[7] = MF16(c[7], f1, b[3][1]);
        if (__builtin_amdgcn_ballot_w64(m0 > thr)) ++hits;
        const int m1 = max(max(tree4(c[4]), tree4(c[5])), max(tree4(c[6]), tree4(c[7])));
        c[0] = MF16(zero, f1, b[0][0]); c[1] = MF16(zero, f1, b[1][0]); c[2] = MF16(zero, f1, b[2][0]); c[3] = MF16(zero, f1, b[3][0]);
        c[0] = MF16(c[0], f0, b[0][1]); c[1] = MF16(c[1], f0, b[1][1]); c[2] = MF16(c[2], f0, b[2][1]); c[3] = MF16(c[3], f0, b[3][1]);
        if (__builtin_amdgcn_ballot_w64(m1 > thr)) ++hits;
    }
    if (hits == 12345) out[threadIdx.x] = hits + tree4(c[0]) + tree4(c[5]);
}
static void run16(int wps, int* d) {
    const int iters = 20000;                                                   // 16 small MFMAs per iteration = 8 big-MFMA equivalents -> 160 k equivalents per wave
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe16<<<256, 256 * wps>>>(100, d);
    hipEventRecord(e0); probe16<<<256, 256 * wps>>>(iters, d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = 160000.0 * wps;
    printf("%-5s %d waves/SIMD: %8.3f ms  %6.1f ns per 32x32x64-equivalent and SIMD  (%.2f PFLOP/s)\n", "sm16", wps, ms, ms * 1e6 / mf, mf * 1024 * 131072.0 / (ms * 1e-3) / 1e15);
}

template <int MODE> static void run(const char* name, int wps, int* d) {
    const int iters = MODE == 0 ? 40000 : (MODE == 1 || MODE == 3 || MODE == 4) ? 20000 : 10000;        // 160 k MFMAs per wave in every mode
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<256, 256 * wps>>>(100, d);
    hipEventRecord(e0); probe<MODE><<<256, 256 * wps>>>(iters, d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = 160000.0 * wps;                                         // MFMAs per SIMD
    printf("%-5s %d waves/SIMD: %8.3f ms  %6.1f ns per MFMA and SIMD  (%.2f PFLOP/s)\n", name, wps, ms, ms * 1e6 / mf, mf * 1024 * 131072.0 / (ms * 1e-3) / 1e15);
}
int main() {
    int* d; hipMalloc(&d, 4096 * 4);
    for (int wps = 1; wps <= 4; ++wps) { run<0>("dep", wps, d); run<1>("skew", wps, d); run<2>("pair", wps, d); run<3>("flat2", wps, d); run<4>("split", wps, d); run16(wps, d); }
    return 0;
}
