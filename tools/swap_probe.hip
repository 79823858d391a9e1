#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned l = threadIdx.x;
    unsigned a = 1000 + l, b = 2000 + l;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[l * 2] = r[0]; out[l * 2 + 1] = r[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 8); k<<<1, 64>>>(d); unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l : {0, 1, 31, 32, 33, 63}) printf("lane %2d: a=%u b=%u -> r0=%u r1=%u\n", l, 1000 + l, 2000 + l, h[2 * l], h[2 * l + 1]);
    return 0;
}
