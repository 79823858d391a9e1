// colwalk_probe.hip — what HBM rate does a kernel reach that walks DOWN an image (a wave owns a strip of columns, row after row),
// as a function of the strip width per wave and of the read / write mix?  (Why the SIFT blur / extrema kernels sit near 3 TB/s.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/colwalk_probe tools/colwalk_probe.hip && tools/colwalk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// V floats per lane (strip = 64 V columns), R reads (layers) and W writes per row
template <int V, int R, int W>
__global__ __launch_bounds__(256) void walk(const float* __restrict__ src, float* __restrict__ dst, int w, int h, int chunk, int64_t layer) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int strips = w / (64 * V), chunks = (h + chunk - 1) / chunk;
    const int task = blockIdx.x * 4 + wave;
    if (task >= strips * chunks) return;
    const int cy = task / strips, sx = task - cy * strips;
    const int x = sx * 64 * V + lane * V, y0 = cy * chunk, y1 = min(h, y0 + chunk);
    const float* S = src + (int64_t)blockIdx.y * layer * R;
    float* D = dst + (int64_t)blockIdx.y * layer * (W > 0 ? W : 1);
    float acc[V] = {};
    for (int y = y0; y < y1; y += 4) {
        float v[4][R][V];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int l = 0; l < R; ++l) __builtin_memcpy(&v[r][l][0], S + layer * l + (int64_t)min(y + r, h - 1) * w + x, 4 * V);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float o[V];
#pragma unroll
            for (int e = 0; e < V; ++e) { o[e] = 0; for (int l = 0; l < R; ++l) o[e] += v[r][l][e]; acc[e] += o[e]; }
            if (y + r < y1)
#pragma unroll
                for (int l = 0; l < W; ++l) __builtin_memcpy(D + layer * l + (int64_t)(y + r) * w + x, o, 4 * V);
        }
    }
    if (W == 0) { float s = 0; for (int e = 0; e < V; ++e) s += acc[e]; if (s == 12345.678f) dst[0] = s; }
}

template <int V, int R, int W>
int run(const float* src, float* dst, int w, int h, int n, int chunk) {
    const int64_t layer = (int64_t)w * h;
    const int strips = w / (64 * V), chunks = (h + chunk - 1) / chunk;
    dim3 grid((strips * chunks + 3) / 4, n);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    walk<V, R, W><<<grid, 256>>>(src, dst, w, h, chunk, layer);
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i) walk<V, R, W><<<grid, 256>>>(src, dst, w, h, chunk, layer);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    const double bytes = (double)layer * 4 * (R + W) * n;
    printf("V=%d (%4d B/row/wave) reads %d writes %d chunk %4d: %7.3f ms  %6.2f TB/s\n", V, 256 * V, R, W, chunk, ms, bytes / ms / 1e9);
    return 0;
}

int main() {
    const int w = 3840, h = 2160, n = 16;       // 16 frames x up to 6 layers of 33 MB
    float *src, *dst;
    CK(hipMalloc(&src, (size_t)w * h * 4 * 6 * n)); CK(hipMalloc(&dst, (size_t)w * h * 4 * 2 * n));
    CK(hipMemset(src, 0, (size_t)w * h * 4 * 6 * n));
    for (int chunk : {36, 360, 2160}) {
        run<1, 1, 1>(src, dst, w, h, n, chunk); run<2, 1, 1>(src, dst, w, h, n, chunk); run<4, 1, 1>(src, dst, w, h, n, chunk);
        run<1, 6, 0>(src, dst, w, h, n, chunk); run<2, 6, 0>(src, dst, w, h, n, chunk); run<4, 6, 0>(src, dst, w, h, n, chunk);
        run<1, 1, 2>(src, dst, w, h, n, chunk); run<4, 1, 2>(src, dst, w, h, n, chunk);
    }
    return 0;
}
