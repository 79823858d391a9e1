// homography.hip.h — verify_model 1: the geometric check as an 8-DOF homography.
//
// The reference verifies with estimateAffinePartial2D (crates/matching-opencv/src/image_utils.rs:45-60; verify.hip.h
// ransac_kernel).  BASELINE.json's north_star / configs[4] ask for "RANSAC homography verification", which has no counterpart
// in the reference: this file restates what cv::findHomography(from, to, RANSAC, thr, mask,
// maxIters, confidence) of OpenCV 4.5.2 computes (calib3d/src/fundam.cpp, ptsetreg.cpp, levmarq.cpp; core/src/lapack.cpp —
// recalled, like the oracle's find_homography which is this kernel's parity target):
//   ransac_h_kernel   RANSACPointSetRegistrator::run with 4-point samples: getSubset on the cv::RNG(-1) stream (duplicates
//                     redrawn, the subset redrawn while HomographyEstimatorCallback::checkSubset rejects it, 10000 attempts),
//                     runKernel = normalised DLT (9x9 L^T L in f64, cv::eigen = Jacobi sweep, smallest eigenvector), f32
//                     re-projection error, sequential acceptance with the adaptive iteration count; then the DLT over all
//                     inliers and LMSolver on the 8 parameters (mask not recomputed), every sum in the oracle's order.
// One wave per (candidate page, frame), as in ransac_kernel: lane = RANSAC iteration.  What differs is the cost of a model:
// a 9x9 symmetric eigenproblem per sample.  Each lane runs OpenCV's Jacobi sweep (pivot = largest off-diagonal element,
// tracked per row / column; its own hypot) on its own slice of LDS — upper triangle 36 + eigenvectors 81 + diagonal 9
// doubles, element e of lane L at e * 64 + L, so every access is conflict free whatever pivot a lane is at — in exactly the
// oracle's operation order: the per-sample model, and with it every inlier count, is bit-identical to the CPU restatement.
// The sample schedule is a pure function of the RNG stream and the points, so it is produced 64 attempts at a time (start
// positions by the same prefix-sum fixed point as ransac_kernel, from an LDS window of the stream), valid subsets queue up,
// and a Jacobi phase always runs on a full wave of samples.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "verify.hip.h"

namespace slideo {

// Jacobi slices per block (ocv.hdlt 0).  A wave issues one instruction per 4 cycles whatever its number of active lanes and the
// sweep is issue bound (~1000 instructions per rotation, ~140 rotations per sample), while the LDS bounds the SLICES per CU
// (1008 B each): 32 slices per wave = four waves per CU, one on every SIMD (64 slices: two waves, half the SIMDs idle — 2x slower).
constexpr int HJ = 32;
constexpr int HJ_TRI = 36, HJ_V = 81, HJ_W = 9;
constexpr int HJ_SLICE = HJ_TRI + HJ_V + HJ_W;       // doubles per lane
constexpr int H_WIN = 512;                  // RNG stream entries staged in LDS per sampling round
constexpr int H_QUEUE = 128;                // valid subsets waiting for a Jacobi phase
constexpr int H_MAX_ATTEMPTS = 10000;       // getSubset(..., maxAttempts) as RANSACPointSetRegistrator::run passes it
constexpr int RANSAC_H_TAIL_WAVES = 8;      // waves of ransac_h_tail_kernel on one candidate's sample schedule
constexpr int RANSAC_H_TAIL_BLOCKS = 256;   // its grid: one block per CU (111 KB of LDS), looping over the tail list

__device__ __forceinline__ int tri9(int i, int j) { return ((i * (17 - i)) >> 1) + (j - i - 1); }   // i < j, upper triangle of 9x9
__device__ __forceinline__ int tri9u(int a, int b) { return a < b ? tri9(a, b) : tri9(b, a); }

// core/src/lapack.cpp hypot (OpenCV's own template)
__device__ __forceinline__ double hypot_cv(double a, double b) {
    a = fabs(a); b = fabs(b);
    if (a > b) { b /= a; return a * sqrt(1 + b * b); }
    if (b > 0) { a /= b; return b * sqrt(1 + a * a); }
    return 0;
}

struct HNorm { double cmx, cmy, cMx, cMy, smx, smy, sMx, sMy; };

// Adds one correspondence (from (X0, Y0) -> to (x0, y0), already normalised) to the upper triangle of L^T L, in the operation
// order of fundam.cpp's loop (products with the structural zeros of Lx / Ly included: not foldable, and they keep the bits).
__device__ __forceinline__ void ltl_add(double (&LtL)[45], double X, double Y, double x, double y) {
    const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
    const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
    int e = 0;
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int k = j; k < 9; ++k) { LtL[e] += Lx[j] * Lx[k] + Ly[j] * Ly[k]; ++e; }
}

// JacobiImpl_<double> (core/src/lapack.cpp) on the lane's LDS slice, then the eigenvector of the smallest eigenvalue.
// LtL: upper triangle incl. diagonal, row-major (45).  `active` lanes without a problem idle; the loop is wave-uniform.
template <int HJ>       // stride between consecutive elements of a slice (= slices interleaved in the buffer)
__device__ __forceinline__ void jacobi9_smallest(double* __restrict__ A, double* __restrict__ V, double* __restrict__ W,
                                              const double (&LtL)[45], bool active, double (&h)[9]) {
    // initial state: V = I, W = diagonal, A = strict upper triangle; indR / indC as packed nibbles
    uint64_t indR = 0, indC = 0;
    if (active) {
        int e = 0;
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
            for (int k = j; k < 9; ++k) {
                if (k == j) W[j * HJ] = LtL[e]; else A[tri9(j, k) * HJ] = LtL[e];
                ++e;
            }
#pragma unroll
        for (int i = 0; i < 81; ++i) V[i * HJ] = (i % 10 == 0) ? 1.0 : 0.0;
        auto at = [&](int j, int k) -> double {           // static indices: j < k
            return LtL[j * 9 - (j * (j - 1)) / 2 + (k - j)];
        };
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (k < 8) {
                int m = k + 1; double mv = fabs(at(k, k + 1));
#pragma unroll
                for (int i = k + 2; i < 9; ++i) { const double v = fabs(at(k, i)); if (mv < v) { mv = v; m = i; } }
                indR |= (uint64_t)m << (4 * k);
            }
            if (k > 0) {
                int m = 0; double mv = fabs(at(0, k));
#pragma unroll
                for (int i = 1; i < k; ++i) { const double v = fabs(at(i, k)); if (mv < v) { mv = v; m = i; } }
                indC |= (uint64_t)m << (4 * k);
            }
        }
    }
    // The pivot candidates — the value at each row's / column's tracked maximum — live in registers: rv[i] = A[i][indR[i]]
    // (i = 0..7), cv[i - 1] = A[indC[i]][i] (i = 1..8).  OpenCV re-reads them from the matrix on every sweep; here they are
    // kept current instead: a rotation changes rows / columns k and l only, every changed element passes through registers
    // (na / nb below), so the caches are patched by compare-selects and the LDS is read once per rotation, for the elements
    // the rotation needs.  Same values, same comparisons, same order.
    double rv[8], cv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { rv[i] = 0; cv[i] = 0; }
    if (active) {
        auto at = [&](int j, int k) -> double { return LtL[j * 9 - (j * (j - 1)) / 2 + (k - j)]; };
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = (int)((indR >> (4 * i)) & 15);
            double v = 0;
#pragma unroll
            for (int j = i + 1; j < 9; ++j) v = c == j ? at(i, j) : v;
            rv[i] = v;
        }
#pragma unroll
        for (int i = 1; i < 9; ++i) {
            const int r = (int)((indC >> (4 * i)) & 15);
            double v = 0;
#pragma unroll
            for (int j = 0; j < i; ++j) v = r == j ? at(j, i) : v;
            cv[i - 1] = v;
        }
    }
    bool run = active;
    int iters = 0;
    while (__builtin_amdgcn_ballot_w64(run) != 0ull) {
        if (run) {
            // pivot: the largest of the row maxima, then of the column maxima (strict >, first wins)
            int k = 0, l = (int)(indR & 15); double p = rv[0], mv = fabs(rv[0]);
#pragma unroll
            for (int i = 1; i < 8; ++i) { const double v = fabs(rv[i]); if (mv < v) { mv = v; k = i; l = (int)((indR >> (4 * i)) & 15); p = rv[i]; } }
#pragma unroll
            for (int i = 1; i < 9; ++i) { const double v = fabs(cv[i - 1]); if (mv < v) { mv = v; k = (int)((indC >> (4 * i)) & 15); l = i; p = cv[i - 1]; } }
            if (fabs(p) <= DBL_EPSILON) run = false;
            else {
                // everything the rotation reads, requested together
                const double wk = W[k * HJ], wl = W[l * HJ];
                double a0[9], b0[9], vk[9], vl[9];
                int ik[9], il[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    const bool on = i != k && i != l;
                    ik[i] = tri9u(on ? i : (k == 0 ? 1 : 0), on ? k : (k == 0 ? 0 : k)) * HJ;      // (a valid address for the two skipped slots)
                    il[i] = tri9u(on ? i : (l == 0 ? 1 : 0), on ? l : (l == 0 ? 0 : l)) * HJ;
                    a0[i] = A[ik[i]]; b0[i] = A[il[i]];
                }
#pragma unroll
                for (int i = 0; i < 9; ++i) { vk[i] = V[(k * 9 + i) * HJ]; vl[i] = V[(l * 9 + i) * HJ]; }
                const double y = (wl - wk) * 0.5;
                double t = fabs(y) + hypot_cv(p, y);
                double s = hypot_cv(p, t);
                const double c = t / s;
                s = p / s; t = (p / t) * p;
                if (y < 0) { s = -s; t = -t; }
                A[tri9(k, l) * HJ] = 0;
                W[k * HJ] = wk - t; W[l * HJ] = wl + t;
                // rows and columns k and l; na[i] / nb[i] = the new (i,k) / (i,l) elements, 0 in the slots of k and l themselves
                // (A[k][l] has just become 0: what the refresh below must see there)
                double na[9], nb[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    const bool on = i != k && i != l;
                    na[i] = on ? a0[i] * c - b0[i] * s : 0.0;
                    nb[i] = on ? a0[i] * s + b0[i] * c : 0.0;
                    if (on) { A[ik[i]] = na[i]; A[il[i]] = nb[i]; }
                }
                // eigenvectors
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    V[(k * 9 + i) * HJ] = vk[i] * c - vl[i] * s;
                    V[(l * 9 + i) * HJ] = vk[i] * s + vl[i] * c;
                }
                // tracked elements of the OTHER rows / columns that the rotation changed: value only (OpenCV does not re-search them)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int c_i = (int)((indR >> (4 * i)) & 15);
                    if (i != k && i != l) rv[i] = c_i == k ? na[i] : (c_i == l ? nb[i] : rv[i]);
                }
#pragma unroll
                for (int i = 1; i < 9; ++i) {
                    const int r_i = (int)((indC >> (4 * i)) & 15);
                    if (i != k && i != l) cv[i - 1] = r_i == k ? na[i] : (r_i == l ? nb[i] : cv[i - 1]);
                }
                // rows / columns k and l: searched again (as OpenCV does), over the values just computed
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int idx = j == 0 ? k : l;
                    int mr = idx + 1, mc = 0; double mvr = -1.0, mvc = -1.0, pr = 0, pc = 0;
#pragma unroll
                    for (int i = 0; i < 9; ++i)
                        if (i != idx) {
                            const double e = j == 0 ? na[i] : nb[i];
                            const double v = fabs(e);
                            if (i > idx) { if (mvr < v) { mvr = v; mr = i; pr = e; } }
                            else { if (mvc < v) { mvc = v; mc = i; pc = e; } }
                        }
                    if (idx < 8) {
                        indR = (indR & ~((uint64_t)15 << (4 * idx))) | ((uint64_t)mr << (4 * idx));
#pragma unroll
                        for (int i = 0; i < 8; ++i) rv[i] = i == idx ? pr : rv[i];
                    }
                    if (idx > 0) {
                        indC = (indC & ~((uint64_t)15 << (4 * idx))) | ((uint64_t)mc << (4 * idx));
#pragma unroll
                        for (int i = 1; i < 9; ++i) cv[i - 1] = i == idx ? pc : cv[i - 1];
                    }
                }
                if (++iters >= 9 * 9 * 30) run = false;
            }
        }
    }
    // descending selection sort of the eigenvalues, rows of V following: only the row that ends last is needed
#pragma unroll
    for (int j = 0; j < 9; ++j) h[j] = 0;
    if (active) {
        double w[9]; int id[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) { w[i] = W[i * HJ]; id[i] = i; }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int m = k, im = id[k]; double wm = w[k];
#pragma unroll
            for (int i = k + 1; i < 9; ++i) if (wm < w[i]) { m = i; wm = w[i]; im = id[i]; }
#pragma unroll
            for (int i = k + 1; i < 9; ++i) if (i == m) { w[i] = w[k]; id[i] = id[k]; }
            w[k] = wm; id[k] = im;
        }
        const int row = id[8];
#pragma unroll
        for (int j = 0; j < 9; ++j) h[j] = V[(row * 9 + j) * HJ];
    }
}

// H = (invHnorm * H0 * Hnorm2) / its (2,2) entry, fundam.cpp's order of products
__device__ __forceinline__ void h_denormalise(const double (&h)[9], const HNorm& n, double (&H)[9]) {
    const double ix = 1. / n.smx, iy = 1. / n.smy;
    double T[9];
#pragma unroll
    for (int j = 0; j < 3; ++j) { T[j] = ix * h[j] + n.cmx * h[6 + j]; T[3 + j] = iy * h[3 + j] + n.cmy * h[6 + j]; T[6 + j] = h[6 + j]; }
    const double n2 = -n.cMx * n.sMx, n5 = -n.cMy * n.sMy;
    double R[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) { R[3 * r] = T[3 * r] * n.sMx; R[3 * r + 1] = T[3 * r + 1] * n.sMy; R[3 * r + 2] = T[3 * r] * n2 + T[3 * r + 1] * n5 + T[3 * r + 2]; }
    const double sc = 1. / R[8];
#pragma unroll
    for (int j = 0; j < 9; ++j) H[j] = R[j] * sc;
}

// normalisation of HomographyEstimatorCallback::runKernel for 4 pairs p[i] = (from.x, from.y, to.x, to.y); false: no spread
__device__ __forceinline__ bool norm4(const float4 (&p)[4], HNorm& n) {
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { cmx += p[i].z; cmy += p[i].w; cMx += p[i].x; cMy += p[i].y; }
    cmx /= 4; cmy /= 4; cMx /= 4; cMy /= 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        smx += fabs(p[i].z - cmx); smy += fabs(p[i].w - cmy);
        sMx += fabs(p[i].x - cMx); sMy += fabs(p[i].y - cMy);
    }
    const bool ok = !(fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON);
    n.cmx = cmx; n.cmy = cmy; n.cMx = cMx; n.cMy = cMy; n.smx = 4 / smx; n.smy = 4 / smy; n.sMx = 4 / sMx; n.sMy = 4 / sMy;
    return ok;
}

// ocv.hdlt 0 — HomographyEstimatorCallback::runKernel on 4 pairs: L^T L + Jacobi.  Returns 1 model or 0.
template <int ST>
__device__ __forceinline__ int dlt4(const float4 (&p)[4], double* A, double* V, double* W, bool active, double (&H)[9]) {
    HNorm n{};
    const bool ok = norm4(p, n) && active;
    double LtL[45];
#pragma unroll
    for (int e = 0; e < 45; ++e) LtL[e] = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        ltl_add(LtL, (p[i].x - n.cMx) * n.sMx, (p[i].y - n.cMy) * n.sMy, (p[i].z - n.cmx) * n.smx, (p[i].w - n.cmy) * n.smy);
    double h[9];
    jacobi9_smallest<ST>(A, V, W, LtL, ok, h);
    if (ok) h_denormalise(h, n, H);
    return ok ? 1 : 0;
}

// ocv.hdlt 1 — the same normalisation, then the 8 equations of the 4 pairs with h33 = 1 by Gaussian elimination with partial
// pivoting (the oracle's homography_4pt_direct / solve_gauss_n<8>, operation for operation).  The 8 x 9 augmented matrix
// lives in registers; a pivot row is brought up by compare-selects (rows below the diagonal only, columns >= c only: the
// entries left of the diagonal are never read again).  ~2.4 k instructions per sample against ~140 k for the Jacobi sweep.
__device__ __forceinline__ int dlt4_direct(const float4 (&p)[4], bool active, double (&H)[9]) {
    HNorm n{};
    bool ok = norm4(p, n) && active;
    double M[8][9];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double x = (p[i].z - n.cmx) * n.smx, y = (p[i].w - n.cmy) * n.smy;
        const double X = (p[i].x - n.cMx) * n.sMx, Y = (p[i].y - n.cMy) * n.sMy;
        const double r0[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, x}, r1[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, y};
#pragma unroll
        for (int j = 0; j < 9; ++j) { M[2 * i][j] = r0[j]; M[2 * i + 1][j] = r1[j]; }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        int pr = c; double best = M[c][c];
#pragma unroll
        for (int r = c + 1; r < 8; ++r) if (fabs(M[r][c]) > fabs(best)) { pr = r; best = M[r][c]; }
        if (best == 0.0) ok = false;
#pragma unroll
        for (int j = c; j < 9; ++j) {
            const double top = M[c][j];
            double up = top;
#pragma unroll
            for (int r = c + 1; r < 8; ++r) { up = pr == r ? M[r][j] : up; M[r][j] = pr == r ? top : M[r][j]; }
            M[c][j] = up;
        }
#pragma unroll
        for (int r = c + 1; r < 8; ++r) {
            const double f = M[r][c] / M[c][c];
#pragma unroll
            for (int j = c; j < 9; ++j) M[r][j] -= f * M[c][j];
        }
    }
    double h[9];
#pragma unroll
    for (int i = 7; i >= 0; --i) {
        double sm = M[i][8];
#pragma unroll
        for (int j = i + 1; j < 8; ++j) sm -= M[i][j] * h[j];
        h[i] = sm / M[i][i];
    }
    h[8] = 1;
    if (ok) h_denormalise(h, n, H);
    return ok ? 1 : 0;
}

// ocv.hdlt 2 — the 4-point model in closed form (the oracle's homography_4pt_closed / square_to_quad, operation for operation):
// the projective maps of the unit square onto the two normalised quadrilaterals, H = S_to * adj(S_from).  ~150 f64 instructions
// and two divisions per sample against ~2.4 k for the elimination and ~140 k for the Jacobi sweep, and a few dozen registers.
__device__ __forceinline__ bool square_to_quad(const double (&x)[4], const double (&y)[4], double (&S)[9]) {
    const double dx1 = x[1] - x[2], dx2 = x[3] - x[2], dx3 = x[0] - x[1] + x[2] - x[3];
    const double dy1 = y[1] - y[2], dy2 = y[3] - y[2], dy3 = y[0] - y[1] + y[2] - y[3];
    const double det = dx1 * dy2 - dx2 * dy1;
    const double g = (dx3 * dy2 - dx2 * dy3) / det, hh = (dx1 * dy3 - dx3 * dy1) / det;
    S[0] = x[1] - x[0] + g * x[1]; S[1] = x[3] - x[0] + hh * x[3]; S[2] = x[0];
    S[3] = y[1] - y[0] + g * y[1]; S[4] = y[3] - y[0] + hh * y[3]; S[5] = y[0];
    S[6] = g; S[7] = hh; S[8] = 1.0;
    return det != 0.0;
}
__device__ __forceinline__ int dlt4_closed(const float4 (&p)[4], bool active, double (&H)[9]) {
    HNorm n{};
    bool ok = norm4(p, n) && active;
    double fx[4], fy[4], tx[4], ty[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        tx[i] = (p[i].z - n.cmx) * n.smx; ty[i] = (p[i].w - n.cmy) * n.smy;
        fx[i] = (p[i].x - n.cMx) * n.sMx; fy[i] = (p[i].y - n.cMy) * n.sMy;
    }
    double A[9], B[9];
    ok = square_to_quad(fx, fy, A) && ok;
    ok = square_to_quad(tx, ty, B) && ok;
    const double J[9] = {A[4] * A[8] - A[5] * A[7], A[2] * A[7] - A[1] * A[8], A[1] * A[5] - A[2] * A[4],
                         A[5] * A[6] - A[3] * A[8], A[0] * A[8] - A[2] * A[6], A[2] * A[3] - A[0] * A[5],
                         A[3] * A[7] - A[4] * A[6], A[1] * A[6] - A[0] * A[7], A[0] * A[4] - A[1] * A[3]};
    double h[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c2 = 0; c2 < 3; ++c2) h[3 * r + c2] = B[3 * r] * J[c2] + B[3 * r + 1] * J[3 + c2] + B[3 * r + 2] * J[6 + c2];
    if (ok) h_denormalise(h, n, H);
    return ok ? 1 : 0;
}

// precomp.hpp haveCollinearPoints on 4 points (only the last point is tested against the pairs before it)
__device__ __forceinline__ bool collinear4(float x0, float y0, float x1, float y1, float x2, float y2, float x3, float y3) {
    const float px[3] = {x0, x1, x2}, py[3] = {y0, y1, y2};
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double dx1 = (double)(px[j] - x3), dy1 = (double)(py[j] - y3);
#pragma unroll
        for (int k = 0; k < j; ++k) {
            const double dx2 = (double)(px[k] - x3), dy2 = (double)(py[k] - y3);
            bad = bad || (fabs(dx2 * dy1 - dy2 * dx1) <= (double)FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)));
        }
    }
    return bad;
}
__device__ __forceinline__ double det3_rows(float ax, float ay, float bx, float by, float cx, float cy) {
    const double a00 = ax, a01 = ay, a02 = 1., a10 = bx, a11 = by, a12 = 1., a20 = cx, a21 = cy, a22 = 1.;
    return a00 * (a11 * a22 - a21 * a12) - a01 * (a10 * a22 - a20 * a12) + a02 * (a10 * a21 - a20 * a11);
}
// HomographyEstimatorCallback::checkSubset, count == 4
__device__ __forceinline__ bool check_subset4(const float4 (&p)[4]) {
    if (collinear4(p[0].x, p[0].y, p[1].x, p[1].y, p[2].x, p[2].y, p[3].x, p[3].y)) return false;
    if (collinear4(p[0].z, p[0].w, p[1].z, p[1].w, p[2].z, p[2].w, p[3].z, p[3].w)) return false;
    const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
    int negative = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 a = p[tt[i][0]], b = p[tt[i][1]], c = p[tt[i][2]];
        negative += (det3_rows(a.x, a.y, b.x, b.y, c.x, c.y) * det3_rows(a.z, a.w, b.z, b.w, c.z, c.w) < 0) ? 1 : 0;
    }
    return negative == 0 || negative == 4;
}

__device__ __forceinline__ float h_error(const float (&Hf)[8], float4 p) {
    const float ww = 1.f / (Hf[6] * p.x + Hf[7] * p.y + 1.f);
    const float dx = (Hf[0] * p.x + Hf[1] * p.y + Hf[2]) * ww - p.z;
    const float dy = (Hf[3] * p.x + Hf[4] * p.y + Hf[5]) * ww - p.w;
    return dx * dx + dy * dy;
}

__device__ __forceinline__ int ransac_update_iters4(double p, double ep, int max_iters) {
    p = fmax(p, 0.); p = fmin(p, 1.);
    ep = fmax(ep, 0.); ep = fmin(ep, 1.);
    double num = fmax(1. - p, DBL_MIN);
    double denom = 1. - pow(1. - ep, 4.0);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)rint(num / denom);
}

__device__ __forceinline__ double sel8(const double (&a)[8], int i) {
    double r = a[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) r = i == j ? a[j] : r;
    return r;
}
__device__ __forceinline__ double sel9(const double (&a)[9], int i) {
    double r = a[0];
#pragma unroll
    for (int j = 1; j < 9; ++j) r = i == j ? a[j] : r;
    return r;
}

// HomographyRefineCallback::compute over the inliers: |r|^2, max |r|, and (want_j) the upper triangle of J^T J (36) and
// J^T r (8).  The sums run over the inliers IN POINT ORDER, as the CPU restatement's loops do — floating-point sums in any
// other order would make the refined matrix differ by an amount that the conditioning of the inlier set amplifies (a
// nearly degenerate set: 1e-4 relative).  Parallelism is across the 44 accumulators instead of across the points: lane e
// owns entry e (0..35: J^T J, 36..43: J^T r), every lane evaluates the (wave-uniform) residual and Jacobian rows of the
// point and picks its two factors; the totals are exchanged at the end.  Every lane returns the same values.
__device__ __forceinline__ double lm8_eval(const float4* pts, const uint8_t* mask, int n, int lane, const double (&h)[8], bool want_j,
                                        double (&AU)[36], double (&v)[8], double* rinf) {
    double S = 0, ri = 0, acc = 0;
    int ra = 0, rb = 0;                                   // this lane's entry: (ra, rb) of J^T J, or ra of J^T r
    if (lane < 36) { int e = lane; while (e >= 8 - ra) { e -= 8 - ra; ++ra; } rb = ra + e; }
    else ra = rb = min(lane - 36, 7);
    const bool is_v = lane >= 36;
    for (int i = 0; i < n; ++i) {
        if (!mask[i]) continue;
        const float4 p = pts[i];
        const double Mx = p.x, My = p.y;
        double ww = h[6] * Mx + h[7] * My + 1.;
        ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
        const double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
        const double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
        const double ex = xi - (double)p.z, ey = yi - (double)p.w;
        S += ex * ex; S += ey * ey;
        ri = fmax(ri, fmax(fabs(ex), fabs(ey)));
        if (want_j) {
            const double J0[8] = {Mx * ww, My * ww, ww, 0, 0, 0, -Mx * ww * xi, -My * ww * xi};
            const double J1[8] = {0, 0, 0, Mx * ww, My * ww, ww, -Mx * ww * yi, -My * ww * yi};
            const double ja = sel8(J0, ra), ka = sel8(J1, ra);
            const double fb = is_v ? ex : sel8(J0, rb), gb = is_v ? ey : sel8(J1, rb);
            acc += ja * fb + ka * gb;
        }
    }
    if (want_j) {
#pragma unroll
        for (int e = 0; e < 36; ++e) AU[e] = __shfl(acc, e);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = __shfl(acc, 36 + e);
    }
    if (rinf) *rinf = ri;
    return S;
}

// 8x8 Gaussian elimination with partial pivoting (ocv.lm 0) on the lane's LDS slice (every lane solves the same system:
// no broadcast, no divergence); sc: >= 72 doubles at stride HJ.  AU: upper triangle of the symmetric matrix, dl: added to
// the diagonal (lambda * D).
template <int HJ>
__device__ __forceinline__ bool solve8_lds(double* sc, const double (&AU)[36], const double (&dl)[8], const double (&b)[8], double (&x)[8]) {
    {
        int e = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = i; j < 8; ++j) {
                const double a = AU[e] + (i == j ? dl[i] : 0.0);
                sc[(i * 9 + j) * HJ] = a;
                if (j != i) sc[(j * 9 + i) * HJ] = AU[e];
                ++e;
            }
#pragma unroll
        for (int i = 0; i < 8; ++i) sc[(i * 9 + 8) * HJ] = b[i];
    }
    bool ok = true;
    for (int c = 0; c < 8 && ok; ++c) {
        int p = c;
        double best = fabs(sc[(c * 9 + c) * HJ]);
        for (int r = c + 1; r < 8; ++r) { const double v = fabs(sc[(r * 9 + c) * HJ]); if (v > best) { best = v; p = r; } }
        if (sc[(p * 9 + c) * HJ] == 0.0) { ok = false; break; }
        if (p != c)
            for (int j = 0; j < 9; ++j) { const double t = sc[(p * 9 + j) * HJ]; sc[(p * 9 + j) * HJ] = sc[(c * 9 + j) * HJ]; sc[(c * 9 + j) * HJ] = t; }
        const double piv = sc[(c * 9 + c) * HJ];
        for (int r = c + 1; r < 8; ++r) {
            const double f = sc[(r * 9 + c) * HJ] / piv;
            for (int j = c; j < 9; ++j) sc[(r * 9 + j) * HJ] -= f * sc[(c * 9 + j) * HJ];
        }
    }
    if (!ok) return false;
    double xs[8];
#pragma unroll
    for (int i = 7; i >= 0; --i) {
        double s = sc[(i * 9 + 8) * HJ];
#pragma unroll
        for (int j = i + 1; j < 8; ++j) s -= sc[(i * 9 + j) * HJ] * xs[j];
        xs[i] = s / sc[(i * 9 + i) * HJ];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = xs[i];
    return true;
}

// (refine_h_kernel<1> calls the solver out of line — inlined three times it costs the kernel half its occupancy —, the lane LM of
// refine_h_eigen_kernel inlines it: no scratch there, -17 %)
template <int HJ>
__device__ __noinline__ bool solve8_lds_call(double* sc, const double (&AU)[36], const double (&dl)[8], const double (&b)[8], double (&x)[8]) {
    return solve8_lds<HJ>(sc, AU, dl, b, x);
}

// ---------------------------------------------------------------------------------------------------------------------
// The sample schedule.  getSubset consumes the cv::RNG stream sequentially — an attempt takes 4 draws plus one per duplicate —
// so where attempt j starts depends on every attempt before it.  (ransac_kernel's prefix-sum fixed point needs about one
// round per attempt WITH a duplicate: fine for 2-point samples over hundreds of votes, hopeless for 4-point samples over the
// 5..30 votes of a wrong candidate page, where most attempts have one.)  Here: the stream window is staged in LDS already
// reduced modulo `count`; nxt[p] = where an attempt starting at window position p ends (a pure function of the window, all
// positions in parallel); the tables T[d] = nxt^(4^d) by pointer jumping, radix 4 (three gathers and a store per entry and
// level: 16 LDS operations per entry for 512 attempts where doubling takes 24, and half the barriers); lane j reads off
// start_j = nxt^j(0) from the base-4 digits of j.  The window holds only what the round's attempts are expected to draw
// (h_window_len: 4.4 entries per attempt at 19 votes, 6.4 at 5 — the tables are as long as the window, and they are the work);
// a round whose attempts do not all fit simply takes fewer of them (`na`), the sequence of attempts is the stream's either way.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int HS_END = H_WIN;                      // absorbing "beyond the window" position
constexpr int HS_TAB = H_WIN + 8;                  // entries per table
constexpr int HS_LEVELS = 3;                       // 4^3 = 64 attempts per round
__host__ __device__ constexpr size_t ransac_h_samp_bytes() { return (size_t)H_WIN * 4 + HS_LEVELS * (size_t)HS_TAB * 2; }

// Window entries `attempts` attempts over `count` points are expected to consume (sum of the geometric waits for 4 distinct
// values) + 5 % + a margin of several sigma; at most `cap`.
__device__ __forceinline__ int h_window_len(int count, int attempts, int cap) {
    const float c = (float)count;
    const float mean = 1.f + c / (c - 1.f) + c / (c - 2.f) + c / (c - 3.f);
    return min(cap, (int)((float)attempts * mean * 1.05f) + 40 + (attempts >> 4));
}

// An attempt read from window position p: 4 distinct indices, `end` = the position after its last draw (`END` if it would
// run past the window's `lim` entries).  win[] holds rng % count.
template <int END>
__device__ __forceinline__ void read_attempt(const uint32_t* win, int lim, int p, uint32_t (&idx)[4], int& end) {
    int q = p;
    bool fit = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t v = 0;
        for (;;) {
            if (q >= lim) { fit = false; break; }
            v = win[q]; ++q;
            bool dup = false;
#pragma unroll
            for (int j = 0; j < i; ++j) dup = dup || v == idx[j];
            if (!dup) break;
        }
        idx[i] = v;
    }
    end = fit ? q : END;
}

// One radix-4 jump level over the window: Tn = Tp o Tp o Tp o Tp on [0, lim) and on the absorbing entry END.
template <int END, int NT>
__device__ __forceinline__ void h_jump4(const uint16_t* Tp, uint16_t* Tn, int lim, int tid) {
    for (int p = tid; p <= lim; p += NT) Tn[p] = Tp[Tp[Tp[Tp[p]]]];     // (entry `lim`: where an attempt that ends with the window ends)
    if (tid == 0) Tn[END] = (uint16_t)END;
}

// Scratch layout helper of ransac_h_kernel: jbuf (Jacobi slices; hdlt 0 only) | points | sampler tables | queue.
// With hdlt 0 the sampler tables alias the Jacobi slices (sampling and scoring alternate, never overlap).
__host__ __device__ constexpr size_t ransac_h_jbuf_bytes(int hdlt) { return hdlt ? 0 : (size_t)HJ_SLICE * HJ * 8; }
__host__ __device__ constexpr size_t ransac_h_lds_bytes(int lds_pts, int hdlt) {
    return (hdlt ? ransac_h_samp_bytes() + 64 : ransac_h_jbuf_bytes(0)) + (size_t)lds_pts * 16 + (size_t)H_QUEUE * 8 + 64;
}

// Two instances share the grid, as ransac_kernel's: <RANSAC_SMALL_PTS, 0> and <RANSAC_LDS_PTS, RANSAC_SMALL_PTS + 1>.
// HDLT = slideo_ocv_variants.hdlt: how a minimal sample becomes a model.  Dynamic LDS: ransac_h_lds_bytes(LDS_PTS, HDLT).
// Leaves per candidate: found, inliers, the RANSAC model in fc.M, the inlier mask in gmask (refine_h_kernel reads them).
template <int LDS_PTS, int MIN_COUNT, int HDLT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(HDLT ? 2 : 1, HDLT == 2 ? 4 : (HDLT ? 2 : 1)))) void ransac_h_kernel(VerifyParams vp, const uint32_t* __restrict__ qofs,
                                                      const slideo_keypoint* __restrict__ frame_kp,
                                                      const float2* __restrict__ page_xy,
                                                      const uint2* __restrict__ votes, const uint32_t* __restrict__ rng_tab,
                                                      FrameCands* __restrict__ fcs, float4* __restrict__ gpts,
                                                      uint8_t* __restrict__ gmask, uint32_t* __restrict__ flags,
                                                      uint32_t round_cap, uint32_t* __restrict__ tail_list, uint32_t* __restrict__ tail_count) {
    extern __shared__ __attribute__((aligned(16))) uint8_t hsm[];
    constexpr int ST = HJ;                          // slices interleaved in jbuf (hdlt 0)
    constexpr int CH = HDLT ? 64 : HJ;              // RANSAC iterations scored per phase
    constexpr size_t REGION0 = HDLT ? ransac_h_samp_bytes() + 64 : ransac_h_jbuf_bytes(0);
    double* jbuf = reinterpret_cast<double*>(hsm);
    uint32_t* win = reinterpret_cast<uint32_t*>(hsm);                           // (aliases jbuf when HDLT == 0)
    uint16_t* jt = reinterpret_cast<uint16_t*>(hsm + (size_t)H_WIN * 4);       // HS_LEVELS tables of HS_TAB
    float4* lpts = reinterpret_cast<float4*>(hsm + REGION0);
    uint2* queue = reinterpret_cast<uint2*>(hsm + REGION0 + (size_t)LDS_PTS * 16);
    static_assert(ransac_h_samp_bytes() + 64 <= ransac_h_jbuf_bytes(0), "sampler tables fit in the Jacobi slices");

    const int r = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
    FrameCands& fc = fcs[f];
    if (r >= fc.ncand) return;
    const int count = fc.count[r];
    if (count < MIN_COUNT || (MIN_COUNT == 0 && count > LDS_PTS)) return;      // the other instance's candidate
    const size_t vbase = (size_t)qofs[f] * vp.k + fc.ofs[r];
    const uint2* vt = votes + vbase;
    float4* pts = count <= LDS_PTS ? lpts : gpts + vbase;
    uint8_t* mask = gmask + vbase;
    [[maybe_unused]] double* A = jbuf + (lane & (ST - 1));
    [[maybe_unused]] double* V = A + HJ_TRI * ST;
    [[maybe_unused]] double* W = V + HJ_V * ST;
    const uint32_t qbase_f = qofs[f];
    for (int i0 = lane; i0 < count; i0 += 256) {
        uint2 v[4]; float2 s[4]; float2 kq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = vt[min(i0 + 64 * u, count - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s[u] = page_xy[v[u].y];
            const slideo_keypoint* kp = frame_kp + qbase_f + v[u].x;
            kq[u] = *reinterpret_cast<const float2*>(&kp->x);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + 64 * u < count) pts[i0 + 64 * u] = make_float4(s[u].x, s[u].y, kq[u].x, kq[u].y);   // from = slide pt, to = frame pt
    }
    __syncthreads();
    double bestH[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int found = 0, inl = 0;
    const float thr2 = (float)(vp.thr * vp.thr);
    if (count >= 4) {
        // count == 4 (`npoints == 4` in findHomography): the kernel alone on the four pairs as they are — one "iteration" with
        // the subset (0, 1, 2, 3), no subset check, every pair an inlier whatever its error
        const bool exact4 = count == 4;
        int niters = exact4 ? 1 : max(vp.max_iters, 1), max_good = 0, base = 0;
        uint32_t pos = 0;                  // stream position of the next attempt
        int nq = 0;                        // queued valid subsets
        int fail_run = 0;                  // consecutive rejected attempts since the last valid subset
        bool sched_end = false;            // getSubset gave up (10000 attempts): no further iteration exists
        bool overflow = false;
        uint32_t rounds = 0;               // sampling rounds so far (round_cap: hand the candidate to ransac_h_tail_kernel)
        const int wlen = h_window_len(max(count, 5), 64, H_WIN);   // window entries staged per round
        if (exact4) { if (lane == 0) queue[0] = make_uint2(0u | (1u << 16), 2u | (3u << 16)); nq = 1; sched_end = true; }
        while (base < niters) {
            // ---- sampling rounds until a full phase of subsets is queued (or as many as the loop can still use) ----
            const int need = min(CH, niters - base);
            while (nq < need && !sched_end) {
                if (pos + H_WIN + 64 > vp.rng_len) { overflow = true; break; }
                if (++rounds > round_cap) {
                    // A candidate whose subsets are almost all rejected (votes that share their points) walks through 10^5 - 10^6
                    // attempts, 64 per round: one wave for tens of milliseconds while the other 9000 candidates take 1 ms each.
                    // It starts again in ransac_h_tail_kernel, 512 attempts per round — same stream, same result.
                    if (lane == 0) tail_list[atomicAdd(tail_count, 1u)] = ((uint32_t)f << 6) | (uint32_t)r;
                    return;
                }
                __syncthreads();
                for (int i = lane; i < wlen; i += 64) win[i] = rng_tab[pos + i] % (uint32_t)count;
                __syncthreads();
                for (int p = lane; p <= wlen; p += 64) {
                    uint32_t idx[4]; int end;
                    read_attempt<HS_END>(win, wlen, p, idx, end);         // (p == wlen: nothing fits, HS_END)
                    jt[p] = (uint16_t)end;
                }
                if (lane == 0) jt[HS_END] = (uint16_t)HS_END;
                __syncthreads();
#pragma unroll
                for (int d = 1; d < HS_LEVELS; ++d) {
                    h_jump4<HS_END, 64>(jt + (d - 1) * HS_TAB, jt + d * HS_TAB, wlen, lane);
                    __syncthreads();
                }
                int st = 0;                                             // start of attempt `lane`: nxt^lane(0), by base-4 digits
#pragma unroll
                for (int d = 0; d < HS_LEVELS; ++d) {
                    const int dg = (lane >> (2 * d)) & 3;
#pragma unroll
                    for (int k = 0; k < 3; ++k) if (dg > k) st = jt[d * HS_TAB + st];
                }
                uint32_t idx[4] = {0, 1, 2, 3};
                int end = HS_END;
                if (st < wlen) read_attempt<HS_END>(win, wlen, st, idx, end);
                const bool have = end != HS_END;                        // the attempt lies inside the window
                const unsigned long long hb = __builtin_amdgcn_ballot_w64(have);
                const int na = hb == ~0ull ? 64 : __builtin_ctzll(~hb); // attempts of this round (a prefix of the lanes)
                if (na == 0) { overflow = true; break; }                // (> 500 duplicates in a row: not with count >= 5)
                pos += (uint32_t)__shfl(end, na - 1);
                const float4 p4[4] = {pts[idx[0]], pts[idx[1]], pts[idx[2]], pts[idx[3]]};
                const bool valid = lane < na && check_subset4(p4);
                const unsigned long long vb = __builtin_amdgcn_ballot_w64(valid);
                if (vb == 0ull) {
                    fail_run += na;
                    if (fail_run >= H_MAX_ATTEMPTS) sched_end = true;
                    continue;
                }
                const int first = __builtin_ctzll(vb);
                if (fail_run + first >= H_MAX_ATTEMPTS) { sched_end = true; continue; }
                const int rank = __builtin_popcountll(vb & ((1ull << lane) - 1ull));
                if (valid && nq + rank < H_QUEUE) queue[nq + rank] = make_uint2(idx[0] | (idx[1] << 16), idx[2] | (idx[3] << 16));
                nq = min(nq + __builtin_popcountll(vb), H_QUEUE);     // (nq < CH <= 64 on entry and <= 64 arrive: never clipped)
                fail_run = na - 1 - (63 - __builtin_clzll(vb));        // rejected attempts after the round's last valid one
            }
            if (overflow) break;
            __syncthreads();
            const int nrun = min(min(nq, CH), niters - base);          // iterations this phase scores
            if (nrun <= 0) break;                                      // schedule exhausted
            // ---- model + inlier count, one iteration per lane ----
            const bool mine = lane < nrun;
            const uint2 qs = queue[min(lane, H_QUEUE - 1)];
            float4 p4[4];
            p4[0] = pts[mine ? (qs.x & 0xFFFFu) : 0]; p4[1] = pts[mine ? (qs.x >> 16) : 0];
            p4[2] = pts[mine ? (qs.y & 0xFFFFu) : 0]; p4[3] = pts[mine ? (qs.y >> 16) : 0];
            double Hm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            int nmodels;
            if constexpr (HDLT == 2) nmodels = dlt4_closed(p4, mine, Hm);
            else if constexpr (HDLT == 1) nmodels = dlt4_direct(p4, mine, Hm);
            else nmodels = dlt4<ST>(p4, A, V, W, mine, Hm);
            if (exact4) {
                found = __shfl(nmodels, 0) > 0;
#pragma unroll
                for (int j = 0; j < 9; ++j) bestH[j] = found ? __shfl(Hm[j], 0) : 0.0;
                break;
            }
            int good = -1;
            if (mine && nmodels > 0) {
                float Hf[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) Hf[j] = (float)Hm[j];
                good = 0;
                for (int i = 0; i < count; ++i) good += (h_error(Hf, pts[i]) <= thr2) ? 1 : 0;
            }
            // ---- sequential acceptance over the phase's iterations ----
            int used = nrun;
            if (__builtin_amdgcn_ballot_w64(mine && good > max(max_good, 3)) != 0ull) {
                for (int i = 0; i < nrun; ++i) {
                    if (base + i >= niters) { used = i; break; }
                    const int g = __shfl(good, i);
                    if (g > max(max_good, 3)) {
                        max_good = g;
#pragma unroll
                        for (int j = 0; j < 9; ++j) bestH[j] = __shfl(Hm[j], i);
                        niters = ransac_update_iters4(vp.conf, (double)(count - g) / count, niters);
                    }
                }
            }
            base += used;
            // drop the consumed subsets
            __syncthreads();
            const uint2 a = queue[min(lane + nrun, H_QUEUE - 1)];
            __syncthreads();
            if (lane + nrun < nq) queue[lane] = a;
            nq -= nrun;
            if (sched_end && nq == 0) break;
        }
        if (overflow) { if (lane == 0) atomicOr(flags, 4u); }
        if (exact4) {
            inl = found ? 4 : 0;
            if (lane < 4) mask[lane] = found ? 1 : 0;
        } else {
            found = max_good > 0;
            if (found) {
                float Hf[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) Hf[j] = (float)bestH[j];
                int c = 0;
                for (int i = lane; i < count; i += 64) {
                    const uint8_t m = h_error(Hf, pts[i]) <= thr2;
                    mask[i] = m; c += m;
                }
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
                inl = c;
            } else {
#pragma unroll
                for (int j = 0; j < 9; ++j) bestH[j] = 0;
            }
        }
    }
    if (lane == 0) {
        fc.found[r] = found; fc.inliers[r] = inl;
        for (int j = 0; j < 9; ++j) fc.M[r][j] = bestH[j];
    }
}

// ransac_h_kernel for the candidates it gave up on (round_cap): the same RANSAC from the start, with the SAMPLE SCHEDULE produced
// by NW waves — 64 NW attempts per round (radix-4 jump tables over log4(64 NW) levels) — and the model / scoring / acceptance
// phases on wave 0 exactly as in ransac_h_kernel (64 iterations per phase).  The valid subsets are a function of the stream
// and the points alone and are queued in stream order, so looking further ahead per round changes nothing but the time: a
// candidate that is bound by its rejected attempts (10^5 - 10^6 of them) runs about NW times faster.
// grid RANSAC_H_TAIL_BLOCKS (the list is taken entry by entry off flags-side counter `tail_next`), block 64 NW; HDLT 1 or 2
// (form 0 is bound by its eigen-solver, not by the schedule).
template <int HDLT, int NW>
__global__ __launch_bounds__(64 * NW) void ransac_h_tail_kernel(VerifyParams vp, const uint32_t* __restrict__ qofs,
                                                                const slideo_keypoint* __restrict__ frame_kp, const float2* __restrict__ page_xy,
                                                                const uint2* __restrict__ votes, const uint32_t* __restrict__ rng_tab,
                                                                FrameCands* __restrict__ fcs, float4* __restrict__ gpts, uint8_t* __restrict__ gmask,
                                                                uint32_t* __restrict__ flags, const uint32_t* __restrict__ tail_list,
                                                                const uint32_t* __restrict__ tail_count, uint32_t* __restrict__ tail_next) {
    static_assert(HDLT == 1 || HDLT == 2, "tail kernel: the cheap sample solvers");
    constexpr int NT = 64 * NW, WIN = H_WIN * NW, TAB = WIN + 8, END = WIN;
    constexpr int LV = NW == 1 ? 3 : NW <= 4 ? 4 : 5;                           // 4^LV >= NT
    static_assert(NW == 1 || NW == 2 || NW == 4 || NW == 8 || NW == 16, "NW: a power of two up to 16");
    constexpr int QCAP = 64 + NT;
    extern __shared__ __attribute__((aligned(16))) uint8_t hsm[];
    uint32_t* win = reinterpret_cast<uint32_t*>(hsm);
    uint16_t* jt = reinterpret_cast<uint16_t*>(hsm + (size_t)WIN * 4);                             // LV tables of TAB
    float4* lpts = reinterpret_cast<float4*>(hsm + (size_t)WIN * 4 + (((size_t)LV * TAB * 2 + 15) & ~(size_t)15));
    uint2* queue = reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(lpts) + (size_t)RANSAC_LDS_PTS * 16);
    __shared__ int s_cnt[2 * NW], s_first[NW], s_last[NW], s_end, s_ctl[4];
    __shared__ uint32_t s_te;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t ntail = *tail_count;
    for (;;) {
    __syncthreads();                                                            // (the previous candidate's LDS is done with)
    if (tid == 0) s_te = atomicAdd(tail_next, 1u);
    __syncthreads();
    const uint32_t te = s_te;
    if (te >= ntail) break;
    const uint32_t tl = tail_list[te];
    const int r = (int)(tl & 63u), f = (int)(tl >> 6);
    FrameCands& fc = fcs[f];
    const int count = fc.count[r];
    const size_t vbase = (size_t)qofs[f] * vp.k + fc.ofs[r];
    const uint2* vt = votes + vbase;
    float4* pts = count <= RANSAC_LDS_PTS ? lpts : gpts + vbase;
    uint8_t* mask = gmask + vbase;
    const uint32_t qbase_f = qofs[f];
    for (int i = tid; i < count; i += NT) {
        const uint2 v = vt[i];
        const float2 sp = page_xy[v.y];
        const slideo_keypoint* kp = frame_kp + qbase_f + v.x;
        pts[i] = make_float4(sp.x, sp.y, kp->x, kp->y);                         // from = slide pt, to = frame pt
    }
    __syncthreads();
    const int wlen = h_window_len(max(count, 5), NT, WIN);                      // window entries staged per round
    double bestH[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};                             // (wave 0)
    int found = 0, inl = 0;
    const float thr2 = (float)(vp.thr * vp.thr);
    // count >= 5 here: ransac_h_kernel never hands over the exact-4 case (no sampling rounds)
    int niters = max(vp.max_iters, 1), max_good = 0, base = 0;
    uint32_t pos = 0;
    int nq = 0, fail_run = 0;
    bool sched_end = false, overflow = false;
    while (base < niters) {
        const int need = min(64, niters - base);
        while (nq < need && !sched_end) {
            if (pos + WIN + 64 > vp.rng_len) { overflow = true; break; }
            __syncthreads();
            for (int i = tid; i < wlen; i += NT) win[i] = rng_tab[pos + i] % (uint32_t)count;
            __syncthreads();
            for (int p = tid; p <= wlen; p += NT) {
                uint32_t idx[4]; int end;
                read_attempt<END>(win, wlen, p, idx, end);                      // (p == wlen: nothing fits, END)
                jt[p] = (uint16_t)end;
            }
            if (tid == 0) jt[END] = (uint16_t)END;
            __syncthreads();
#pragma unroll 1
            for (int d = 1; d < LV; ++d) {
                h_jump4<END, NT>(jt + (d - 1) * TAB, jt + d * TAB, wlen, tid);
                __syncthreads();
            }
            int st = 0;                                                         // start of attempt `tid`: nxt^tid(0), by base-4 digits
#pragma unroll
            for (int d = 0; d < LV; ++d) {
                const int dg = (tid >> (2 * d)) & 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) if (dg > k) st = jt[d * TAB + st];
            }
            uint32_t idx[4] = {0, 1, 2, 3};
            int end = END;
            if (st < wlen) read_attempt<END>(win, wlen, st, idx, end);
            const bool have = end != END;                                       // the attempt lies inside the window (a prefix of the threads)
            const unsigned long long hb = __builtin_amdgcn_ballot_w64(have);
            if (lane == 0) s_cnt[wave] = __builtin_popcountll(hb);
            __syncthreads();
            int na = 0;
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) na += s_cnt[w2];
            if (na == 0) { overflow = true; break; }
            if (tid == na - 1) s_end = end;
            const float4 p4[4] = {pts[idx[0]], pts[idx[1]], pts[idx[2]], pts[idx[3]]};
            const bool valid = tid < na && check_subset4(p4);
            const unsigned long long vb = __builtin_amdgcn_ballot_w64(valid);
            if (lane == 0) {
                s_cnt[NW + wave] = __builtin_popcountll(vb);
                s_first[wave] = vb ? 64 * wave + __builtin_ctzll(vb) : -1;
                s_last[wave] = vb ? 64 * wave + 63 - __builtin_clzll(vb) : -1;
            }
            __syncthreads();
            pos += (uint32_t)s_end;
            int nvalid = 0, before = 0, first = -1, last = -1;
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) {
                if (w2 < wave) before += s_cnt[NW + w2];
                nvalid += s_cnt[NW + w2];
                if (first < 0) first = s_first[w2];
                if (s_last[w2] >= 0) last = s_last[w2];
            }
            if (nvalid == 0) {
                fail_run += na;
                if (fail_run >= H_MAX_ATTEMPTS) sched_end = true;
                continue;
            }
            if (fail_run + first >= H_MAX_ATTEMPTS) { sched_end = true; continue; }
            const int rank = before + __builtin_popcountll(vb & ((1ull << lane) - 1ull));
            if (valid && nq + rank < QCAP) queue[nq + rank] = make_uint2(idx[0] | (idx[1] << 16), idx[2] | (idx[3] << 16));
            nq = min(nq + nvalid, QCAP);                                       // (nq < 64 on entry and <= NT arrive: never clipped)
            fail_run = na - 1 - last;                                          // rejected attempts after the round's last valid one
        }
        if (overflow) break;
        __syncthreads();
        const int nrun = min(min(nq, 64), niters - base);                      // iterations this phase scores
        if (nrun <= 0) break;
        if (wave == 0) {
            // ---- model + inlier count, one iteration per lane; sequential acceptance: ransac_h_kernel's code ----
            const bool mine = lane < nrun;
            const uint2 qs = queue[min(lane, QCAP - 1)];
            float4 p4[4];
            p4[0] = pts[mine ? (qs.x & 0xFFFFu) : 0]; p4[1] = pts[mine ? (qs.x >> 16) : 0];
            p4[2] = pts[mine ? (qs.y & 0xFFFFu) : 0]; p4[3] = pts[mine ? (qs.y >> 16) : 0];
            double Hm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            int nmodels;
            if constexpr (HDLT == 2) nmodels = dlt4_closed(p4, mine, Hm);
            else nmodels = dlt4_direct(p4, mine, Hm);
            int good = -1;
            if (mine && nmodels > 0) {
                float Hf[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) Hf[j] = (float)Hm[j];
                good = 0;
                for (int i = 0; i < count; ++i) good += (h_error(Hf, pts[i]) <= thr2) ? 1 : 0;
            }
            int used = nrun;
            if (__builtin_amdgcn_ballot_w64(mine && good > max(max_good, 3)) != 0ull) {
                for (int i = 0; i < nrun; ++i) {
                    if (base + i >= niters) { used = i; break; }
                    const int g = __shfl(good, i);
                    if (g > max(max_good, 3)) {
                        max_good = g;
#pragma unroll
                        for (int j = 0; j < 9; ++j) bestH[j] = __shfl(Hm[j], i);
                        niters = ransac_update_iters4(vp.conf, (double)(count - g) / count, niters);
                    }
                }
            }
            if (lane == 0) { s_ctl[0] = max_good; s_ctl[1] = niters; s_ctl[2] = used; }
        }
        __syncthreads();
        max_good = s_ctl[0]; niters = s_ctl[1]; base += s_ctl[2];
        // drop the consumed subsets
        uint2 keep[(QCAP + NT - 1) / NT];
#pragma unroll
        for (int u = 0; u < (QCAP + NT - 1) / NT; ++u) keep[u] = queue[min(tid + u * NT + nrun, QCAP - 1)];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < (QCAP + NT - 1) / NT; ++u) if (tid + u * NT + nrun < nq) queue[tid + u * NT] = keep[u];
        nq -= nrun;
        if (sched_end && nq == 0) break;
    }
    if (wave != 0) continue;
    if (overflow) { if (lane == 0) atomicOr(flags, 4u); }
    found = max_good > 0;
    if (found) {
        float Hf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) Hf[j] = (float)bestH[j];
        int c = 0;
        for (int i = lane; i < count; i += 64) {
            const uint8_t mk = h_error(Hf, pts[i]) <= thr2;
            mask[i] = mk; c += mk;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
        inl = c;
    } else {
#pragma unroll
        for (int j = 0; j < 9; ++j) bestH[j] = 0;
    }
    if (lane == 0) {
        fc.found[r] = found; fc.inliers[r] = inl;
        for (int j = 0; j < 9; ++j) fc.M[r][j] = bestH[j];
    }
    }
}
__host__ __device__ constexpr size_t ransac_h_tail_lds_bytes(int nw) {
    const int lv = nw == 1 ? 3 : nw <= 4 ? 4 : 5;
    return (size_t)H_WIN * nw * 4 + (((size_t)lv * (H_WIN * nw + 8) * 2 + 15) & ~(size_t)15) + (size_t)RANSAC_LDS_PTS * 16 + (size_t)(64 + 64 * nw) * 8 + 64;
}

// lm8_eval with ONE LANE per candidate: the lane walks its candidate's points in order and owns all 44 accumulators — the sums
// and their order are lm8_eval's (entry (a, b): acc += J0[a] J0[b] + J1[a] J1[b]; J^T r: acc += J0[a] ex + J1[a] ey).
// pts / mask: the lane's column of an LDS tile, element i at [i * HJ].
__device__ __forceinline__ double lm8_eval_lane(const float4* pts, const uint8_t* mask, int n, const double (&h)[8], bool want_j,
                                                double (&AU)[36], double (&v)[8], double* rinf) {
    double S = 0, ri = 0;
    if (want_j) {
#pragma unroll
        for (int e = 0; e < 36; ++e) AU[e] = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0;
    }
    for (int i = 0; i < n; ++i) {
        if (!mask[i * HJ]) continue;
        const float4 p = pts[i * HJ];
        const double Mx = p.x, My = p.y;
        double ww = h[6] * Mx + h[7] * My + 1.;
        ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
        const double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
        const double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
        const double ex = xi - (double)p.z, ey = yi - (double)p.w;
        S += ex * ex; S += ey * ey;
        ri = fmax(ri, fmax(fabs(ex), fabs(ey)));
        if (want_j) {
            const double J0[8] = {Mx * ww, My * ww, ww, 0, 0, 0, -Mx * ww * xi, -My * ww * xi};
            const double J1[8] = {0, 0, 0, Mx * ww, My * ww, ww, -Mx * ww * yi, -My * ww * yi};
            int e = 0;
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b = a; b < 8; ++b) { AU[e] += J0[a] * J0[b] + J1[a] * J1[b]; ++e; }
#pragma unroll
            for (int a = 0; a < 8; ++a) v[a] += J0[a] * ex + J1[a] * ey;
        }
    }
    if (rinf) *rinf = ri;
    return S;
}

// LMSolverImpl::run on 8 parameters, one lane per candidate: refine_h_kernel<1>'s loop with the lane's own solves (its LDS slice,
// stride HJ) and evaluations.  x: in = the start, out = the result.
__device__ __forceinline__ void lm8_run_lane(double* sc, const float4* pts, const uint8_t* mask, int n, int max_iters, double (&x)[8]) {
    const double eps = (double)FLT_EPSILON;
    double xd[8], AU[36], v[8], D[8], d[8], dl[8], rinf = 0;
    double S = lm8_eval_lane(pts, mask, n, x, true, AU, v, &rinf);
    {
        int e = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { D[i] = AU[e]; e += 8 - i; }
    }
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    int iter = 0;
    for (;;) {
#pragma unroll
        for (int i = 0; i < 8; ++i) dl[i] = lambda * D[i];
        if (!solve8_lds<HJ>(sc, AU, dl, v, d)) {
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = 0.0;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) xd[i] = x[i] - d[i];
        double dummyA[36], dummyv[8];
        const double Sd = lm8_eval_lane(pts, mask, n, xd, false, dummyA, dummyv, nullptr);
        double dS = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            double t = 2 * v[i];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int a = i < j ? i : j, b = i < j ? j : i;
                t -= AU[a * 8 - (a * (a - 1)) / 2 + (b - a)] * d[j];
            }
            dS += d[i] * t;
        }
        const double R = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
        if (R > Rhi) { lambda *= 0.5; if (lambda < lc) lambda = 0; }
        else if (R < Rlo) {
            double t = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) t += d[i] * v[i];
            double nu = (Sd - S) / (fabs(t) > DBL_EPSILON ? t : 1) + 2;
            nu = fmin(fmax(nu, 2.), 10.);
            if (lambda == 0) {
                double maxval = DBL_EPSILON;
                const double zero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int i = 0; i < 8; ++i) {
                    double e8[8], col[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) e8[j] = j == i ? 1.0 : 0.0;
                    if (solve8_lds<HJ>(sc, AU, zero8, e8, col)) {
                        double ci = 0;
#pragma unroll
                        for (int j = 0; j < 8; ++j) ci = j == i ? col[j] : ci;
                        maxval = fmax(maxval, fabs(ci));
                    }
                }
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = xd[i];
            lm8_eval_lane(pts, mask, n, x, true, AU, v, &rinf);
        }
        iter++;
        double dinf = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) dinf = fmax(dinf, fabs(d[i]));
        if (!(iter < max_iters && dinf >= eps && rinf >= eps)) break;
    }
}

// fundam.cpp after the RANSAC: `result && npoints > 4` — runKernel over the inliers, then LMSolver (maxIters = refine_iters)
// on H[0..7]; the mask stays RANSAC's.  One wave per (candidate, frame); every sum in point order (lm8_eval); the one
// 8x8 solves run on lane 0's slice.  grid (max_cand, B), block 64, LDS static.
// Three launches (r03): the 9x9 eigenproblem of runKernel is ONE lane's work for ~0.5 ms — 58 % of the 8.9 s of candidate time
// a headline unit's 9000 candidates spent in the single kernel, with 63 lanes waiting.  PHASE 0 (this kernel) normalises,
// accumulates L^T L in point order and leaves both in a scratch record; refine_h_eigen_kernel solves 32 candidates' problems per
// wave, a lane and an LDS slice each (ransac_h_kernel's Jacobi, unchanged); PHASE 1 (this kernel) runs the LM from the record's
// matrix.  The arithmetic of a candidate is what the single kernel did, operation for operation.
struct RefineRec { double H[9]; double LtL[45]; HNorm n; double ok; double lm_done; };      // 64 doubles per (frame, candidate)
template <int PHASE>
__global__ __launch_bounds__(64) void refine_h_kernel(VerifyParams vp, const uint32_t* __restrict__ qofs,
                                                      const slideo_keypoint* __restrict__ frame_kp,
                                                      const float2* __restrict__ page_xy, const uint2* __restrict__ votes,
                                                      FrameCands* __restrict__ fcs, float4* __restrict__ gpts,
                                                      const uint8_t* __restrict__ gmask, RefineRec* __restrict__ recs,
                                                      uint32_t* __restrict__ eig_list, uint32_t* __restrict__ eig_count) {
    __shared__ float4 lpts[RANSAC_LDS_PTS];
    __shared__ uint8_t lmask[RANSAC_LDS_PTS];
    __shared__ double jslice[HJ_SLICE];
    const int r = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
    FrameCands& fc = fcs[f];
    if (r >= fc.ncand) return;
    const int count = fc.count[r], inl = fc.inliers[r];
    if (count <= 4 || !fc.found[r] || inl <= 0 || vp.refine_iters <= 0) return;
    RefineRec& rec = recs[(size_t)f * gridDim.x + r];
    if (PHASE == 1 && rec.lm_done != 0.0) return;                                // refine_h_eigen_kernel ran this candidate's LM too
    const size_t vbase = (size_t)qofs[f] * vp.k + fc.ofs[r];
    const uint2* vt = votes + vbase;
    float4* pts = count <= RANSAC_LDS_PTS ? lpts : gpts + vbase;
    const uint8_t* mask = count <= RANSAC_LDS_PTS ? lmask : gmask + vbase;
    const uint32_t qbase_f = qofs[f];
    for (int i = lane; i < count; i += 64) {
        const uint2 v = vt[i];
        const float2 s = page_xy[v.y];
        const slideo_keypoint* kp = frame_kp + qbase_f + v.x;
        pts[i] = make_float4(s.x, s.y, kp->x, kp->y);
        if (count <= RANSAC_LDS_PTS) lmask[i] = gmask[vbase + i];
    }
    __syncthreads();
    constexpr int ST = 1;
    [[maybe_unused]] double* A = jslice;                    // (the 8x8 solves' scratch)
    double bestH[9];
    if constexpr (PHASE == 1) {
#pragma unroll
        for (int j = 0; j < 9; ++j) bestH[j] = rec.ok != 0.0 ? rec.H[j] : fc.M[r][j];       // (runKernel returning 0 leaves H as RANSAC found it)
    } else {
    // runKernel over the inliers.  All sums in point order (see lm8_eval): lanes 0..3 own the four centroid / deviation sums,
    // lanes 0..44 the 45 entries of L^T L.
    HNorm n{};
    const double cnt = (double)inl;
    {
        double acc = 0;
        for (int i = 0; i < count; ++i) {
            if (!mask[i]) continue;
            const float4 p = pts[i];
            acc += (double)(lane == 0 ? p.z : (lane == 1 ? p.w : (lane == 2 ? p.x : p.y)));
        }
        n.cmx = __shfl(acc, 0) / cnt; n.cmy = __shfl(acc, 1) / cnt; n.cMx = __shfl(acc, 2) / cnt; n.cMy = __shfl(acc, 3) / cnt;
        const double cen = lane == 0 ? n.cmx : (lane == 1 ? n.cmy : (lane == 2 ? n.cMx : n.cMy));
        acc = 0;
        for (int i = 0; i < count; ++i) {
            if (!mask[i]) continue;
            const float4 p = pts[i];
            acc += fabs((double)(lane == 0 ? p.z : (lane == 1 ? p.w : (lane == 2 ? p.x : p.y))) - cen);
        }
        n.smx = __shfl(acc, 0); n.smy = __shfl(acc, 1); n.sMx = __shfl(acc, 2); n.sMy = __shfl(acc, 3);
    }
    const bool ok = !(fabs(n.smx) < DBL_EPSILON || fabs(n.smy) < DBL_EPSILON || fabs(n.sMx) < DBL_EPSILON || fabs(n.sMy) < DBL_EPSILON);
    if (lane == 0) { rec.ok = ok ? 1.0 : 0.0; rec.lm_done = 0.0; }
    if (ok) {
        n.smx = cnt / n.smx; n.smy = cnt / n.smy; n.sMx = cnt / n.sMx; n.sMy = cnt / n.sMy;
        int rj = 0, rk = 0;
        { int e = min(lane, 44); while (e >= 9 - rj) { e -= 9 - rj; ++rj; } rk = rj + e; }
        double acc = 0;
        for (int i = 0; i < count; ++i) {
            if (!mask[i]) continue;
            const float4 p = pts[i];
            const double x = (p.z - n.cmx) * n.smx, y = (p.w - n.cmy) * n.smy;
            const double X = (p.x - n.cMx) * n.sMx, Y = (p.y - n.cMy) * n.sMy;
            const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
            const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
            acc += sel9(Lx, rj) * sel9(Lx, rk) + sel9(Ly, rj) * sel9(Ly, rk);
        }
        if (lane < 45) rec.LtL[lane] = acc;
        if (lane == 0) { rec.n = n; eig_list[atomicAdd(eig_count, 1u)] = (uint32_t)f * gridDim.x + (uint32_t)r; }
    }
    return;
    }
    // LMSolverImpl::run, 8 parameters
    const double eps = (double)FLT_EPSILON;
    double x[8], xd[8], AU[36], v[8], D[8], d[8], dl[8], rinf = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = bestH[i];
    double S = lm8_eval(pts, mask, count, lane, x, true, AU, v, &rinf);
    {
        int e = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { D[i] = AU[e]; e += 8 - i; }
    }
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    int iter = 0;
    for (;;) {
#pragma unroll
        for (int i = 0; i < 8; ++i) dl[i] = lambda * D[i];
        {
            int okd = 0;
            if (lane == 0) okd = solve8_lds_call<ST>(A, AU, dl, v, d) ? 1 : 0;
            okd = __shfl(okd, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = okd ? __shfl(d[i], 0) : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) xd[i] = x[i] - d[i];
        double dummyA[36], dummyv[8];
        const double Sd = lm8_eval(pts, mask, count, lane, xd, false, dummyA, dummyv, nullptr);
        double dS = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            double t = 2 * v[i];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int a = i < j ? i : j, b = i < j ? j : i;
                t -= AU[a * 8 - (a * (a - 1)) / 2 + (b - a)] * d[j];
            }
            dS += d[i] * t;
        }
        const double R = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
        if (R > Rhi) { lambda *= 0.5; if (lambda < lc) lambda = 0; }
        else if (R < Rlo) {
            double t = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) t += d[i] * v[i];
            double nu = (Sd - S) / (fabs(t) > DBL_EPSILON ? t : 1) + 2;
            nu = fmin(fmax(nu, 2.), 10.);
            if (lambda == 0) {
                double maxval = DBL_EPSILON;
                const double zero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int i = 0; i < 8; ++i) {
                    double e8[8], col[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) e8[j] = j == i ? 1.0 : 0.0;
                    int okc = 0;
                    double ci = 0;
                    if (lane == 0) {
                        okc = solve8_lds_call<ST>(A, AU, zero8, e8, col) ? 1 : 0;
#pragma unroll
                        for (int j = 0; j < 8; ++j) ci = j == i ? col[j] : ci;
                    }
                    if (__shfl(okc, 0)) maxval = fmax(maxval, fabs(__shfl(ci, 0)));
                }
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = xd[i];
            lm8_eval(pts, mask, count, lane, x, true, AU, v, &rinf);
        }
        iter++;
        double dinf = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) dinf = fmax(dinf, fabs(d[i]));
        if (!(iter < vp.refine_iters && dinf >= eps && rinf >= eps)) break;
    }
    if (lane == 0) {
        for (int i = 0; i < 8; ++i) fc.M[r][i] = x[i];
        fc.M[r][8] = bestH[8];
    }
}

// The eigenproblems of refine_h_kernel<0>'s records: lane l < HJ of block b solves entry b HJ + l of the list on its LDS slice
// (jacobi9_smallest<HJ>, the sweep ransac_h_kernel runs per sample) and leaves the denormalised matrix in the record.  A
// candidate with at most REFINE_LANE_PTS votes — two thirds of all candidates have 5 - 7 inliers — gets its LM in the same lane
// (lm8_run_lane: the points in a lane-major LDS tile, the 8x8 solves on the slice), and refine_h_kernel<1> skips it (lm_done).
// grid ceil(candidates / HJ), block 64, dynamic LDS refine_h_eigen_lds_bytes().
constexpr int REFINE_LANE_PTS = 48;
__host__ __device__ constexpr size_t refine_h_eigen_lds_bytes() { return ransac_h_jbuf_bytes(0) + (size_t)REFINE_LANE_PTS * HJ * 16 + (size_t)REFINE_LANE_PTS * HJ; }
__global__ __launch_bounds__(64) void refine_h_eigen_kernel(VerifyParams vp, const uint32_t* __restrict__ qofs, const slideo_keypoint* __restrict__ frame_kp,
                                                            const float2* __restrict__ page_xy, const uint2* __restrict__ votes,
                                                            FrameCands* __restrict__ fcs, const uint8_t* __restrict__ gmask, int max_cand,
                                                            RefineRec* __restrict__ recs, const uint32_t* __restrict__ eig_list,
                                                            const uint32_t* __restrict__ eig_count, int lane_lm) {
    extern __shared__ __attribute__((aligned(16))) uint8_t hsm[];
    const uint32_t total = *eig_count;
    if (blockIdx.x * HJ >= total) return;
    const int lane = threadIdx.x;
    const uint32_t e = blockIdx.x * HJ + (uint32_t)lane;
    const bool mine = lane < HJ && e < total;
    double* A = reinterpret_cast<double*>(hsm) + (lane & (HJ - 1));
    double* V = A + HJ_TRI * HJ;
    double* W = V + HJ_V * HJ;
    float4* tile = reinterpret_cast<float4*>(hsm + ransac_h_jbuf_bytes(0)) + (lane & (HJ - 1));
    uint8_t* mtile = hsm + ransac_h_jbuf_bytes(0) + (size_t)REFINE_LANE_PTS * HJ * 16 + (lane & (HJ - 1));
    const uint32_t ci = mine ? eig_list[e] : eig_list[blockIdx.x * HJ];
    RefineRec& rec = recs[ci];
    double H[9];
    {
        double LtL[45];
#pragma unroll
        for (int j = 0; j < 45; ++j) LtL[j] = rec.LtL[j];
        const HNorm n = rec.n;
        double h[9];
        jacobi9_smallest<HJ>(A, V, W, LtL, mine, h);
        if (mine) {
            h_denormalise(h, n, H);
#pragma unroll
            for (int j = 0; j < 9; ++j) rec.H[j] = H[j];
        }
    }
    if (!lane_lm) return;
    const int f = (int)(ci / (uint32_t)max_cand), r = (int)(ci - (uint32_t)f * (uint32_t)max_cand);
    FrameCands& fc = fcs[f];
    const int count = fc.count[r];
    const bool small = mine && count <= REFINE_LANE_PTS;
    if (small) {
        const size_t vbase = (size_t)qofs[f] * vp.k + fc.ofs[r];
        const uint32_t qbase_f = qofs[f];
        for (int i = 0; i < count; ++i) {
            const uint2 v = votes[vbase + i];
            const float2 sp = page_xy[v.y];
            const slideo_keypoint* kp = frame_kp + qbase_f + v.x;
            tile[i * HJ] = make_float4(sp.x, sp.y, kp->x, kp->y);
            mtile[i * HJ] = gmask[vbase + i];
        }
        double x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = H[j];
        lm8_run_lane(A, tile, mtile, count, vp.refine_iters, x);
#pragma unroll
        for (int j = 0; j < 8; ++j) fc.M[r][j] = x[j];
        fc.M[r][8] = H[8];
        rec.lm_done = 1.0;
    }
}

}  // namespace slideo
