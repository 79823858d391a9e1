// Probe: v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (e2m1) +-1 operands, unit scales.
// Hypothesis: A lane l holds row (l&31), k in [(l>>5)*32, +32) as 32 nibbles (4 dwords, low nibble first);
// B lane l holds col (l&31), same k range; C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void probe(const uint32_t* a, const uint32_t* b, float* c, int fmt) {
    int l = threadIdx.x;
    v8i av = {0,0,0,0,0,0,0,0}, bv = {0,0,0,0,0,0,0,0};
    for (int i = 0; i < 4; ++i) { av[i] = a[l * 4 + i]; bv[i] = b[l * 4 + i]; }
    v16f acc = {0};
    // cbsz = A format, blgp = B format: 4 = FP4 ; scales: E8M0 127 = 1.0
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int r = 0; r < 16; ++r) c[l * 16 + r] = acc[r];
}

int main() {
    static int8_t A[32][64], B[64][32];
    srand(1);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) { A[i][k] = (rand() & 1) ? 1 : -1; B[k][i] = (rand() & 1) ? 1 : -1; }
    uint32_t ha[64 * 4] = {0}, hb[64 * 4] = {0};
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) {
        int k = (l >> 5) * 32 + j;
        uint32_t na = A[l & 31][k] > 0 ? 0x2 : 0xA, nb = B[k][l & 31] > 0 ? 0x2 : 0xA;
        ha[l * 4 + j / 8] |= na << (4 * (j & 7)); hb[l * 4 + j / 8] |= nb << (4 * (j & 7));
    }
    uint32_t *da, *db; float* dc;
    CHECK(hipMalloc(&da, sizeof(ha))); CHECK(hipMalloc(&db, sizeof(hb))); CHECK(hipMalloc(&dc, 64 * 16 * 4));
    CHECK(hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice));
    probe<<<1, 64>>>(da, db, dc, 4); CHECK(hipDeviceSynchronize());
    float hc[64 * 16]; CHECK(hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
        int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        int ref = 0; for (int k = 0; k < 64; ++k) ref += A[row][k] * B[k][col];
        if ((int)hc[l * 16 + r] != ref) { if (bad < 8) printf("mismatch lane %d reg %d: got %g want %d\n", l, r, hc[l * 16 + r], ref); ++bad; }
    }
    printf("fp4 32x32x64 probe: %d mismatches of 1024 (sample c[0]=%g)\n", bad, hc[0]);
    return 0;
}
