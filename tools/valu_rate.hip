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
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pa[UNROLL], pb = {1.0f, 1.0f};
    const unsigned long long sp = ((unsigned long long)s << 32) | s;
    for (int i = 0; i < UNROLL; ++i) pa[i] = f2{(float)threadIdx.x, (float)i};
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
            if (OP == 12) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 13) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 14) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 15) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 16) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[i]));
            if (OP == 17) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 18) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(b));
            if (OP == 19) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 20) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(s), "v"(b));
            if (OP == 21) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 22) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (OP == 23) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 24) asm volatile("v_pk_mad_u16 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 25) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 26) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(pa[i]) : "v"(pb));
            if (OP == 27) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa[i]) : "v"(pb));
            if (OP == 28) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pa[i]) : "v"(pb));
            if (OP == 29) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 30) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 31) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 32) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
            if (OP == 33) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 34) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 35) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 36) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
            if (OP == 37) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 38) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 39) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pa[i]) : "v"(pb), "v"(pa[(i + 5) & 15]));
            if (OP == 40) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pa[i]) : "s"(sp), "v"(pa[(i + 5) & 15]));
            if (OP == 41) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(s), "v"(a[(i + 5) & 15]));
            if (OP == 42) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(a[(i + 5) & 15]));
            if (OP == 43) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(pa[i]) : "s"(sp), "v"(pa[(i + 5) & 15]));
            if (OP == 44) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(a[(i + 5) & 15]));
            if (OP == 45) asm volatile("v_pk_add_f32 %0, %1, %2" : "+v"(pa[i]) : "v"(pb), "v"(pa[(i + 5) & 15]));
            if (OP == 46) asm volatile("v_add_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(a[(i + 5) & 15]));
            if (OP == 47) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 3) & 15]), "v"(a[(i + 5) & 15]));
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < UNROLL; ++i) r ^= a[i] ^ __float_as_uint(pa[i].x + pa[i].y);
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
    run<12>("v_fmac_f32 (VOP2)", d); run<20>("v_fmac_f32 sgpr", d); run<13>("v_fma_f32 (VOP3)", d); run<23>("v_add_f32", d);
    run<14>("v_max_f32 (VOP2)", d); run<15>("v_max3_f32", d); run<25>("v_max_i32", d); run<16>("v_cvt_f32_ubyte1", d); run<21>("v_cvt_pk_u8_f32", d);
    run<17>("v_pk_max_u16", d); run<18>("v_pk_sub_u16 clamp", d); run<24>("v_pk_mad_u16", d); run<19>("v_or_b32", d); run<22>("v_mov_b32_dpp", d);
    run<26>("v_pk_fma_f32", d); run<27>("v_pk_mul_f32", d); run<28>("v_pk_add_f32", d); run<29>("v_mul_f32", d); run<30>("v_sub_f32", d);
    run<31>("v_and_b32", d); run<32>("v_lshlrev_b32", d); run<33>("v_min_u32", d); run<34>("v_sub_u32", d); run<35>("v_pk_add_u16", d);
    run<36>("v_cndmask_b32", d); run<37>("v_add3_u32", d); run<38>("v_mov_b32", d);
    run<39>("pk_fma 3 distinct vgpr pairs", d); run<40>("pk_fma sgpr pair + 2 vgpr pairs", d); run<43>("pk_fma sgpr bcast + 2 vgpr", d);
    run<41>("v_fma sgpr + 2 distinct vgpr", d); run<42>("v_fma 3 distinct vgpr", d); run<44>("v_fmac 3 distinct vgpr", d);
    run<45>("pk_add 3 distinct", d); run<46>("v_add_f32 3 distinct", d); run<47>("v_max3_f32 3 distinct", d);
    return 0;
}
