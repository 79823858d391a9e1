// Micro-benchmark: issue rate of the integer VALU ops the Hamming kernel could be built from (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 4096, UNROLL = 16;
template <int OP> __global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t s) {
    uint32_t a[UNROLL];
    for (int i = 0; i < UNROLL; ++i) a[i] = threadIdx.x * 2654435761u + i;
    uint32_t b = s;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (OP == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 2) asm volatile("v_med3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 4) asm volatile("v_dot8_u32_u4 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 5) asm volatile("v_dot4_u32_u8 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 6) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 7) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 8) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
            if (OP == 9) asm volatile("v_sad_u8 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 10) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 11) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < UNROLL; ++i) r ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int OP> int run(const char* name, uint32_t* d) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = 256 * 8;
    k<OP><<<blocks, 256>>>(d, 0x12345678u); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); k<OP><<<blocks, 256>>>(d, 0x12345678u); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double ops = (double)blocks * 256 * ITERS * UNROLL;
    printf("%-18s %8.3f ms  %7.2f Tlaneop/s\n", name, ms, ops / ms / 1e9);
    return 0;
}
int main() {
    uint32_t* d; CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));
    run<0>("v_xor_b32", d); run<8>("v_xor_b32 sgpr", d); run<1>("v_bcnt_u32_b32", d); run<2>("v_med3_u32", d); run<3>("v_add_u32", d);
    run<4>("v_dot8_u32_u4", d); run<5>("v_dot4_u32_u8", d); run<6>("v_lshl_or_b32", d); run<7>("v_and_or_b32", d);
    run<9>("v_sad_u8", d); run<10>("v_perm_b32", d); run<11>("v_mul_u32_u24", d);
    return 0;
}
