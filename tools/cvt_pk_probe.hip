// Does v_cvt_pk_u8_f32 round to nearest even and saturate, i.e. equal saturate_cast<uchar>(cvRound(x))?
//   hipcc --offload-arch=gfx950 -O2 -o cvt_pk_probe cvt_pk_probe.hip && ./cvt_pk_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
__global__ void probe(const float* x, unsigned* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 0u, 0u);
}
int main() {
    const int n = 400 * 64;
    float* hx = new float[n]; unsigned* ho = new unsigned[n];
    for (int i = 0; i < n; ++i) hx[i] = -20.f + (float)i / 64.f;      // includes every x.5 tie in [-20, 380)
    float* dx; unsigned* dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    probe<<<(n + 255) / 256, 256>>>(dx, dout, n);
    hipMemcpy(ho, dout, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        long r = lrintf(hx[i]);
        unsigned want = r < 0 ? 0u : (r > 255 ? 255u : (unsigned)r);
        if (ho[i] != want && bad++ < 10) printf("x=%f got %u want %u\n", hx[i], ho[i], want);
    }
    printf("cvt_pk_u8_f32 vs saturate(cvRound): %d mismatches of %d\n", bad, n);
    return 0;
}
