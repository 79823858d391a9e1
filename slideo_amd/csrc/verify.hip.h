// verify.hip.h — per-frame decision kernels (everything after the kNN).
//
// Restates crates/matching-opencv/src/lib.rs:268-389 on the GPU:
//   vote_kernel       5 % tolerance vote, bucket by page, top-40 pages, ordered
//                     per-page match lists                       (lib.rs:268-295)
//   ransac_kernel     estimateAffinePartial2D: 2-point RANSAC similarity with
//                     OpenCV's adaptive iteration count + LM refine
//                     (image_utils.rs:45-60, [OCV A.9])           (lib.rs:296-313)
//   rate_kernel       sort by inliers, keep <= 10, rating filter  (lib.rs:329-333)
//   reproject_kernel  warpAffine(nearest, inverse map) -> to_small_image ->
//                     compute_similarity, fused: the full-resolution warp is
//                     never materialised ([OCV A.10-A.12])        (lib.rs:335-351)
//   verdict_kernel    sort by similarity, keep > 0.5, first       (lib.rs:370-389)
//   small_image_kernel / ssd_kernel: to_small_image + compute_similarity for
//                     page ingest (lib.rs:128) and the changed-frame mask
//                     (video_capture.rs:86-98).
// Orders the reference leaves to HashMap iteration are canonical:
// pages tie-break by ascending index.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "geom.h"
#include "types.h"

namespace slideo {


// ---------------------------------------------------------------------------
// vote_kernel: grid B, block 256, dynamic LDS: counts[P] u32 | rank[P] u8 (padded) | runcnt[max_cand][256] u32
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vote_kernel(VerifyParams vp, const uint32_t* __restrict__ keys,
                                                   const uint32_t* __restrict__ qofs,
                                                   const int32_t* __restrict__ train_page, int npages,
                                                   FrameCands* __restrict__ fcs, uint2* __restrict__ votes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t* counts = reinterpret_cast<uint32_t*>(smem);
    uint8_t* rank = smem + (size_t)npages * 4;
    uint32_t* runcnt = reinterpret_cast<uint32_t*>(smem + (size_t)npages * 4 + (((size_t)npages + 15) & ~(size_t)15));
    __shared__ unsigned long long red[4];
    __shared__ int32_t s_page[MAXC], s_count[MAXC], s_ofs[MAXC];

    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t q0 = qofs[f], n = qofs[f + 1] - q0;
    const int k = vp.k;
    FrameCands& fc = fcs[f];
    for (int i = tid; i < npages; i += 256) { counts[i] = 0; rank[i] = 0xFF; }
    __syncthreads();
    // tolerance vote: d < best * tol (f32, strict)      lib.rs:275
    // A thread owns a contiguous run of queries.  Per query it fetches the whole sorted key list with 8 x 16-byte
    // loads in flight (entry by entry, every key cost a dependent round trip), and because the list is sorted the
    // entries that pass `d < lim` are a prefix of it.
    const uint32_t runq = (n + 255) / 256;
    const uint32_t qa = min(n, (uint32_t)tid * runq), qb = min(n, qa + runq);
    auto for_each_vote = [&](auto&& fn) {
        for (uint32_t q = qa; q < qb; ++q) {
            const uint4* kq4 = reinterpret_cast<const uint4*>(keys + (size_t)(q0 + q) * VOTE_KLIST);
            uint32_t key[VOTE_KLIST];
#pragma unroll
            for (int i = 0; i < VOTE_KLIST / 4; ++i) { const uint4 v = kq4[i]; key[4 * i] = v.x; key[4 * i + 1] = v.y; key[4 * i + 2] = v.z; key[4 * i + 3] = v.w; }
            if (vp.ratio > 0.f) {                                            // ratio test on the two nearest rows
                if (k >= 2 && key[0] != KNN_EMPTY && key[1] != KNN_EMPTY &&
                    (float)(key[0] >> KNN_KEY_SHIFT) < vp.ratio * (float)(key[1] >> KNN_KEY_SHIFT))
                    fn(q, key[0] & KNN_IDX_MASK);
                continue;
            }
            const float lim = (float)(key[0] >> KNN_KEY_SHIFT) * vp.tol;
#pragma unroll
            for (int r = 0; r < VOTE_KLIST; ++r) {
                const bool pass = r < k && key[r] != KNN_EMPTY && (float)(key[r] >> KNN_KEY_SHIFT) < lim;
                if (!pass) break;                                        // sorted list: nothing further passes
                fn(q, key[r] & KNN_IDX_MASK);
            }
        }
    };
    for_each_vote([&](uint32_t, uint32_t t) { atomicAdd(&counts[train_page[t]], 1u); });
    __syncthreads();
    // top max_cand pages by (count desc, page asc)       lib.rs:284-295
    int nc = 0;
    for (int round = 0; round < vp.max_cand; ++round) {
        unsigned long long best = 0;
        for (int p = tid; p < npages; p += 256) {
            uint32_t c = counts[p];
            if (c && rank[p] == 0xFF) {
                unsigned long long comp = ((unsigned long long)c << 32) | (0xFFFFFFFFu - (uint32_t)p);
                best = comp > best ? comp : best;
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            unsigned long long o = __shfl_xor(best, d);
            best = o > best ? o : best;
        }
        if (lane == 0) red[wave] = best;
        __syncthreads();
        best = red[0];
        for (int w = 1; w < 4; ++w) best = red[w] > best ? red[w] : best;
        __syncthreads();
        if (best == 0) break;
        int p = (int)(0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFu));
        if (tid == 0) { s_page[nc] = p; s_count[nc] = (int32_t)(best >> 32); rank[p] = (uint8_t)nc; }
        ++nc;
        __syncthreads();
    }
    if (tid == 0) {
        int o = 0;
        for (int i = 0; i < nc; ++i) { s_ofs[i] = o; o += s_count[i]; }
        fc.ncand = nc; fc.nsurv = 0;
        for (int i = 0; i < nc; ++i) { fc.page[i] = s_page[i]; fc.count[i] = s_count[i]; fc.ofs[i] = s_ofs[i]; }
    }
    __syncthreads();
    if (nc == 0) return;
    // ordered placement: each thread owns a contiguous run of queries, i.e. of entries in (q asc, r asc) order
    for (int c = 0; c < nc; ++c) runcnt[c * 256 + tid] = 0;
    for_each_vote([&](uint32_t, uint32_t t) { uint8_t r = rank[train_page[t]]; if (r != 0xFF) runcnt[r * 256 + tid]++; });
    __syncthreads();
    // exclusive scan over the 256 threads, per candidate (wave w takes candidates w, w+4, ...)
    for (int c = wave; c < nc; c += 4) {
        uint32_t v[4], s = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = runcnt[c * 256 + lane * 4 + j]; s += v[j]; }
        uint32_t inc = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d); if (lane >= d) inc += o; }
        uint32_t ex = inc - s;
#pragma unroll
        for (int j = 0; j < 4; ++j) { runcnt[c * 256 + lane * 4 + j] = ex; ex += v[j]; }
    }
    __syncthreads();
    uint2* vout = votes + (size_t)q0 * k;
    for_each_vote([&](uint32_t q, uint32_t t) {
        uint8_t r = rank[train_page[t]];
        if (r != 0xFF) {
            uint32_t pos = (uint32_t)s_ofs[r] + runcnt[r * 256 + tid]++;
            vout[pos] = make_uint2(q, t);
        }
    });
}

// ---------------------------------------------------------------------------
// ransac_kernel: one wave per (candidate, frame).  grid (max_cand, B), block 64.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void similarity_from_2(float4 p, float4 q, double M[6]) {
    // p = (x1,y1,X1,Y1), q = (x2,y2,X2,Y2); from (x,y) -> to (X,Y)     [OCV A.9] runKernel
    double x1 = p.x, y1 = p.y, X1 = p.z, Y1 = p.w, x2 = q.x, y2 = q.y, X2 = q.z, Y2 = q.w;
    double d = 1. / ((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
    double S0 = d * ((X1 - X2) * (x1 - x2) + (Y1 - Y2) * (y1 - y2));
    double S1 = d * ((Y1 - Y2) * (x1 - x2) - (X1 - X2) * (y1 - y2));
    double S2 = d * ((Y1 - Y2) * (x1 * y2 - x2 * y1) - (X1 * y2 - X2 * y1) * (y1 - y2) - (X1 * x2 - X2 * x1) * (x1 - x2));
    double S3 = d * (-(X1 - X2) * (x1 * y2 - x2 * y1) - (Y1 * x2 - Y2 * x1) * (x1 - x2) - (Y1 * y2 - Y2 * y1) * (y1 - y2));
    M[0] = S0; M[4] = S0; M[1] = -S1; M[2] = S2; M[3] = S1; M[5] = S3;
}

__device__ __forceinline__ int ransac_update_iters(double p, double ep, int max_iters) {
    p = fmax(p, 0.); p = fmin(p, 1.);
    ep = fmax(ep, 0.); ep = fmin(ep, 1.);
    double num = fmax(1. - p, DBL_MIN);
    double denom = 1. - pow(1. - ep, 2.0);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)rint(num / denom);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = fmax(v, __shfl_xor(v, d));
    return v;
}

__device__ __forceinline__ bool solve4(const double Ain[16], const double bin[4], double x[4]) {
    double A[4][5];
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) A[i][j] = Ain[i * 4 + j]; A[i][4] = bin[i]; }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r) if (fabs(A[r][c]) > fabs(A[p][c])) p = r;
        if (A[p][c] == 0.0) return false;
        if (p != c) for (int j = 0; j < 5; ++j) { double t = A[p][j]; A[p][j] = A[c][j]; A[c][j] = t; }
        for (int r = c + 1; r < 4; ++r) {
            double fct = A[r][c] / A[c][c];
            for (int j = c; j < 5; ++j) A[r][j] -= fct * A[c][j];
        }
    }
    for (int i = 3; i >= 0; --i) {
        double s = A[i][4];
        for (int j = i + 1; j < 4; ++j) s -= A[i][j] * x[j];
        x[i] = s / A[i][i];
    }
    return true;
}

// residual sums over the inliers for h = (a, b, tx, ty); wave-parallel
__device__ __forceinline__ double lm_eval(const float4* pts, const uint8_t* mask, int n, int lane, const double h[4],
                                          bool want_j, double A[16], double v[4], double* rinf) {
    double S = 0, ri = 0, sq = 0, sx = 0, sy = 0, cnt = 0, v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    for (int i = lane; i < n; i += 64) {
        if (!mask[i]) continue;
        float4 p = pts[i];
        double Mx = p.x, My = p.y;
        double ex = (h[0] * Mx - h[1] * My + h[2]) - (double)p.z;
        double ey = (h[1] * Mx + h[0] * My + h[3]) - (double)p.w;
        S += ex * ex; S += ey * ey;
        ri = fmax(ri, fmax(fabs(ex), fabs(ey)));
        if (want_j) {
            sq += Mx * Mx + My * My; sx += Mx; sy += My; cnt += 1.0;
            v0 += Mx * ex + My * ey; v1 += -My * ex + Mx * ey; v2 += ex; v3 += ey;
        }
    }
    S = wave_sum(S); ri = wave_max(ri);
    if (want_j) {
        sq = wave_sum(sq); sx = wave_sum(sx); sy = wave_sum(sy); cnt = wave_sum(cnt);
        v[0] = wave_sum(v0); v[1] = wave_sum(v1); v[2] = wave_sum(v2); v[3] = wave_sum(v3);
        // J^T J for rows {x,-y,1,0},{y,x,0,1}
        A[0] = sq;  A[1] = 0;   A[2] = sx;  A[3] = sy;
        A[4] = 0;   A[5] = sq;  A[6] = -sy; A[7] = sx;
        A[8] = sx;  A[9] = -sy; A[10] = cnt; A[11] = 0;
        A[12] = sy; A[13] = sx; A[14] = 0;  A[15] = cnt;
    }
    if (rinf) *rinf = ri;
    return S;
}

// The 2-point sample schedule of one chunk of 64 iterations when some iteration redraws (ptsetreg.cpp getSubset: idx = next() %
// count, the second index redrawn while it equals the first).  Where iteration j starts in the stream depends on the redraws of
// every iteration before it.  The window w[0, RS_WIN) = stream % count sits in LDS; nxt[p] = where an iteration starting at p
// ends (all p in parallel); T1 = nxt^4, T2 = nxt^16 by pointer jumping; lane j reads start_j = nxt^j(0) off the base-4 digits
// of j (homography.hip.h's schedule, two draws instead of four).  Returns false — nothing changed — if 64 iterations do not fit
// in the window (the caller's prefix-sum fixed point then takes the chunk); `stream` = rng_tab + pos, RS_WIN entries readable.
constexpr int RS_WIN = 256, RS_END = RS_WIN, RS_TAB = RS_WIN + 8;
__device__ __forceinline__ bool ransac_schedule_from_window(const uint32_t* __restrict__ stream, uint32_t count, uint32_t* win, uint16_t* jt,
                                                            int lane, uint32_t& a, uint32_t& b, uint32_t& pos) {
    __syncthreads();                                                    // (one wave per block: orders the LDS accesses)
#pragma unroll
    for (int u = 0; u < RS_WIN / 64; ++u) win[lane + 64 * u] = stream[lane + 64 * u] % count;
    __syncthreads();
    auto attempt = [&](int p, uint32_t& i0, uint32_t& i1) -> int {      // end position, RS_END if it leaves the window
        if (p >= RS_WIN) return RS_END;
        i0 = win[p];
        int q = p + 1;
        for (;;) {
            if (q >= RS_WIN) return RS_END;
            i1 = win[q];
            if (i1 != i0) return q + 1;
            ++q;
        }
    };
    uint16_t* T0 = jt; uint16_t* T1 = jt + RS_TAB; uint16_t* T2 = jt + 2 * RS_TAB;
    for (int p = lane; p <= RS_WIN; p += 64) { uint32_t x, y; T0[p] = (uint16_t)attempt(p, x, y); }    // (p == RS_WIN == RS_END: absorbing)
    __syncthreads();
    for (int p = lane; p <= RS_WIN; p += 64) T1[p] = T0[T0[T0[T0[p]]]];
    __syncthreads();
    for (int p = lane; p <= RS_WIN; p += 64) T2[p] = T1[T1[T1[T1[p]]]];
    __syncthreads();
    int st = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) if ((lane & 3) > k) st = T0[st];
#pragma unroll
    for (int k = 0; k < 3; ++k) if (((lane >> 2) & 3) > k) st = T1[st];
#pragma unroll
    for (int k = 0; k < 3; ++k) if ((lane >> 4) > k) st = T2[st];
    uint32_t i0 = 0, i1 = 0;
    const int end = attempt(st, i0, i1);
    if (__builtin_amdgcn_ballot_w64(end == RS_END) != 0ull) return false;
    a = i0; b = i1;
    pos += (uint32_t)__shfl(end, 63);
    return true;
}

// Two instances share the grid: <RANSAC_SMALL_PTS> takes the candidates with few votes (most of the <= 40 per frame;
// 4 KB of LDS, so the register file and not LDS bounds the occupancy), <RANSAC_LDS_PTS> the rest; a block whose
// candidate belongs to the other instance exits at once.
template <int LDS_PTS, int MIN_COUNT>
__global__ __launch_bounds__(64) void ransac_kernel(VerifyParams vp, const uint32_t* __restrict__ qofs,
                                                    const slideo_keypoint* __restrict__ frame_kp,
                                                    const float2* __restrict__ page_xy,
                                                    const uint2* __restrict__ votes, const uint32_t* __restrict__ rng_tab,
                                                    FrameCands* __restrict__ fcs, float4* __restrict__ gpts,
                                                    uint8_t* __restrict__ gmask, uint32_t* __restrict__ flags) {
    __shared__ float4 lpts[LDS_PTS];
    __shared__ uint8_t lmask[LDS_PTS];
    __shared__ uint32_t rs_win[RS_WIN];
    __shared__ uint16_t rs_jt[3][RS_TAB];
    const int r = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
    FrameCands& fc = fcs[f];
    if (r >= fc.ncand) return;
    const int count = fc.count[r];
    if (count < MIN_COUNT || (MIN_COUNT == 0 && count > LDS_PTS)) return;      // the other instance's candidate
    const size_t vbase = (size_t)qofs[f] * vp.k + fc.ofs[r];
    const uint2* vt = votes + vbase;
    float4* pts = count <= LDS_PTS ? lpts : gpts + vbase;
    uint8_t* mask = count <= LDS_PTS ? lmask : gmask + vbase;
    // (4 votes per lane and round: the three dependent gathers of a vote — vote, slide point, frame keypoint — are
    // each issued for all four before the first is used)
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
            if (i0 + 64 * u < count) pts[i0 + 64 * u] = make_float4(s[u].x, s[u].y, kq[u].x, kq[u].y);   // from = slide pt, to = frame pt   lib.rs:299-302
    }
    __syncthreads();
    double bestM[6] = {0, 0, 0, 0, 0, 0};
    int found = 0, inl = 0;
    const float thr2 = (float)(vp.thr * vp.thr);
    if (count == 2) {
        similarity_from_2(pts[0], pts[1], bestM);
        found = 1; inl = 2;
    } else if (count > 2) {
        int niters = max(vp.max_iters, 1), max_good = 0;
        uint32_t pos = 0;
        for (int base = 0; base < niters; base += 64) {
            // sample schedule: cv::RNG((uint64)-1) stream, idx = next() % count, second index redrawn while equal
            if (pos + 4 * 64 + 64 > vp.rng_len) { if (lane == 0) atomicOr(flags, 4u); break; }
            uint32_t a = rng_tab[pos + 2 * lane] % (uint32_t)count;
            uint32_t b = rng_tab[pos + 2 * lane + 1] % (uint32_t)count;
            if (__builtin_amdgcn_ballot_w64(a == b) == 0ull) pos += 128;
            else if (vp.sched_window && ransac_schedule_from_window(rng_tab + pos, (uint32_t)count, rs_win, &rs_jt[0][0], lane, a, b, pos)) {
                // (some iteration redraws its second index: the 64 start positions from jump tables over an LDS window of the
                // stream — two table levels whatever the number of redraws, which is what candidates with 3 - 20 votes need:
                // a third of their iterations redraw)
            } else {
                // Some iteration redraws its second index, which shifts the stream position of every later iteration:
                // start_{j+1} = start_j + 2 + e_j, e_j = redraws of iteration j.  Instead of replaying the 64 iterations one
                // after the other, every lane evaluates its iteration from a guessed start (pos + 2 lane + shift) and the
                // shifts are corrected by a prefix sum of the e_j until nothing moves: lane 0 is right from the start, a lane
                // is right one round after all the lanes before it, and a round that changes no shift satisfies the recurrence.
                uint32_t shift = 0, total = 0;
                for (;;) {
                    const uint32_t p0 = pos + 2 * lane + shift;
                    const uint32_t i0 = rng_tab[min(p0, vp.rng_len - 1)] % (uint32_t)count;
                    uint32_t e = 0, i1;
                    for (;;) {
                        i1 = rng_tab[min(p0 + 1 + e, vp.rng_len - 1)] % (uint32_t)count;
                        if (i1 != i0 || p0 + 1 + e >= vp.rng_len - 1) break;
                        ++e;
                    }
                    uint32_t inc = e;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if (lane >= d) inc += o; }
                    const uint32_t nshift = inc - e;
                    const bool changed = nshift != shift;
                    shift = nshift;
                    if (__builtin_amdgcn_ballot_w64(changed) == 0ull) { a = i0; b = i1; total = __shfl(inc, 63); break; }
                }
                pos += 128 + total;
                if (pos >= vp.rng_len) { if (lane == 0) atomicOr(flags, 4u); break; }
            }
            double M[6];
            similarity_from_2(pts[a], pts[b], M);
            const float F0 = (float)M[0], F1 = (float)M[1], F2 = (float)M[2], F3 = (float)M[3], F4 = (float)M[4], F5 = (float)M[5];
            int good = 0;
            for (int i = 0; i < count; ++i) {
                float4 p = pts[i];
                float ea = F0 * p.x + F1 * p.y + F2 - p.z;
                float eb = F3 * p.x + F4 * p.y + F5 - p.w;
                float e = ea * ea + eb * eb;
                good += (e <= thr2) ? 1 : 0;
            }
            // sequential acceptance over the 64 iterations of this chunk.  max_good only grows, so if no iteration of the
            // chunk beats the current best none is accepted and the scan is skipped (the usual case for the many
            // candidates without a consistent model, which otherwise spent most of their time here).
            bool stop = false;
            if (__builtin_amdgcn_ballot_w64(good > max(max_good, 1) && base + lane < niters) == 0ull) {
                if (base + 64 >= niters) break;
                continue;
            }
            for (int i = 0; i < 64; ++i) {
                if (base + i >= niters) { stop = true; break; }
                int g = __shfl(good, i);
                if (g > max(max_good, 1)) {
                    max_good = g;
#pragma unroll
                    for (int j = 0; j < 6; ++j) bestM[j] = __shfl(M[j], i);
                    niters = ransac_update_iters(vp.conf, (double)(count - g) / count, niters);
                }
            }
            if (stop) break;
        }
        found = max_good > 0;
        if (found) {
            const float F0 = (float)bestM[0], F1 = (float)bestM[1], F2 = (float)bestM[2], F3 = (float)bestM[3], F4 = (float)bestM[4], F5 = (float)bestM[5];
            int c = 0;
            for (int i = lane; i < count; i += 64) {
                float4 p = pts[i];
                float ea = F0 * p.x + F1 * p.y + F2 - p.z;
                float eb = F3 * p.x + F4 * p.y + F5 - p.w;
                float e = ea * ea + eb * eb;
                uint8_t m = e <= thr2;
                mask[i] = m; c += m;
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
            inl = c;
            __syncthreads();
            if (vp.refine_iters > 0 && inl > 0) {
                // LMSolverImpl::run over the inliers ([OCV] calib3d/src/levmarq.cpp), 4 parameters
                const double eps = (double)FLT_EPSILON;
                double x[4] = {bestM[0], bestM[3], bestM[2], bestM[5]}, xd[4], A[16], v[4], D[4], d[4], Ap[16], rinf = 0;
                double S = lm_eval(pts, mask, count, lane, x, true, A, v, &rinf);
                for (int i = 0; i < 4; ++i) D[i] = A[i * 4 + i];
                const double Rlo = 0.25, Rhi = 0.75;
                double lambda = 1, lc = 0.75;
                int iter = 0;
                for (;;) {
                    for (int i = 0; i < 16; ++i) Ap[i] = A[i];
                    for (int i = 0; i < 4; ++i) Ap[i * 4 + i] += lambda * D[i];
                    if (!solve4(Ap, v, d)) { d[0] = d[1] = d[2] = d[3] = 0; }
                    for (int i = 0; i < 4; ++i) xd[i] = x[i] - d[i];
                    double Sd = lm_eval(pts, mask, count, lane, xd, false, nullptr, nullptr, nullptr);
                    double dS = 0;
                    for (int i = 0; i < 4; ++i) {
                        double t = 2 * v[i];
                        for (int j = 0; j < 4; ++j) t -= A[i * 4 + j] * d[j];
                        dS += d[i] * t;
                    }
                    double R = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
                    if (R > Rhi) { lambda *= 0.5; if (lambda < lc) lambda = 0; }
                    else if (R < Rlo) {
                        double t = 0;
                        for (int i = 0; i < 4; ++i) t += d[i] * v[i];
                        double nu = (Sd - S) / (fabs(t) > DBL_EPSILON ? t : 1) + 2;
                        nu = fmin(fmax(nu, 2.), 10.);
                        if (lambda == 0) {
                            double maxval = DBL_EPSILON;
                            for (int i = 0; i < 4; ++i) {
                                double e4[4] = {0, 0, 0, 0}, col[4];
                                e4[i] = 1;
                                if (solve4(A, e4, col)) maxval = fmax(maxval, fabs(col[i]));
                            }
                            lambda = lc = 1. / maxval;
                            nu *= 0.5;
                        }
                        lambda *= nu;
                    }
                    if (Sd < S) {
                        S = Sd;
                        for (int i = 0; i < 4; ++i) x[i] = xd[i];
                        lm_eval(pts, mask, count, lane, x, true, A, v, &rinf);
                    }
                    iter++;
                    double dinf = fmax(fmax(fabs(d[0]), fabs(d[1])), fmax(fabs(d[2]), fabs(d[3])));
                    if (!(iter < vp.refine_iters && dinf >= eps && rinf >= eps)) break;
                }
                bestM[0] = bestM[4] = x[0]; bestM[1] = -x[1]; bestM[2] = x[2]; bestM[3] = x[1]; bestM[5] = x[3];
            }
        } else {
            for (int j = 0; j < 6; ++j) bestM[j] = 0;
        }
    }
    if (lane == 0) {
        fc.found[r] = found; fc.inliers[r] = inl;
        for (int j = 0; j < 6; ++j) fc.M[r][j] = bestM[j];
    }
}

// ---------------------------------------------------------------------------
// rate_kernel: one thread per frame.      lib.rs:329-333
// ---------------------------------------------------------------------------
// One wave per frame, lane = candidate slot.  The stable "rating desc" order (lib.rs:329) is each lane's rank =
// number of candidates that beat it (higher rating, or equal rating and a lower slot); both filters are monotone in the
// rating, so the survivors are the first `ns` of that order and survivor s is the lane with rank s.
__global__ __launch_bounds__(64) void rate_kernel(VerifyParams vp, int nframes, FrameCands* __restrict__ fcs,
                                                  const PageInfo* __restrict__ pages,
                                                  PairDesc* __restrict__ pair_list, uint32_t* __restrict__ pair_count) {
    static_assert(MAXC <= 64, "one lane per candidate");
    const int f = blockIdx.x, lane = threadIdx.x;
    if (f >= nframes) return;
    FrameCands& fc = fcs[f];
    const int nc = fc.ncand;
    const bool mine = lane < nc;
    const int inl = mine ? fc.inliers[lane] : -1;
    int rank = 0;
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
        const int o = __shfl(inl, j);
        rank += (j < nc && (o > inl || (o == inl && j < lane))) ? 1 : 0;
    }
    const int top = min(nc, vp.max_rated);
    const unsigned long long first = __builtin_amdgcn_ballot_w64(mine && rank == 0);
    const double best = first ? (double)__shfl(inl, __builtin_ctzll(first)) : 0.0;
    const double rating = (double)inl;
    const bool pass = mine && rank < top && rating > vp.min_rating && rating / best > vp.min_rating_ratio;     // lib.rs:333
    const int ns = __builtin_popcountll(__builtin_amdgcn_ballot_w64(pass));
    uint32_t base = 0;
    if (lane == 0) { fc.nsurv = ns; if (ns > 0) base = atomicAdd(pair_count, (uint32_t)ns); }
    base = __shfl(base, 0);
    if (pass) {       // compact (frame, survivor) work list for reproject_kernel; order is irrelevant (sums are per pair)
        fc.surv[rank] = lane; fc.ssd[rank] = 0ull; fc.sim[rank] = 0.f;
        const PageInfo pg = pages[fc.page[lane]];
        PairDesc d;
        d.f = f; d.s = rank; d.area_idx = pg.area_idx; d._pad = 0; d.small_ofs = pg.small_ofs;
#pragma unroll
        for (int j = 0; j < 9; ++j) d.M[j] = fc.M[lane][j];
        pair_list[base + rank] = d;
    }
}

// ---------------------------------------------------------------------------
// INTER_AREA of a virtual source image given by `fetch(sx, sy, bgr[3])`.
// Same accumulation order as ResizeArea_Invoker ([OCV A.11]): per source row the
// x taps accumulate into buf (f32, mul then add), rows accumulate sum (+)= beta*buf.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int sat_int_d(double v) {
    if (!(v > -2147483648.0)) return INT32_MIN;
    if (v >= 2147483647.0) return INT32_MAX;
    return (int)rint(v);
}
__device__ __forceinline__ uint8_t sat_u8_f(float v) {
    int i = (int)rintf(v);
    return (uint8_t)(i < 0 ? 0 : (i > 255 ? 255 : i));
}

template <class Fetch>
__device__ __forceinline__ void area_pixel(const AreaGeom& ag, const AreaTap* __restrict__ taps, const int32_t* __restrict__ idx,
                                           int dx, int dy, Fetch fetch, uint8_t out[3]) {
    if (ag.fast) {
        int sum0 = 0, sum1 = 0, sum2 = 0;
        for (int yy = 0; yy < ag.iscale_y; ++yy)
            for (int xx = 0; xx < ag.iscale_x; ++xx) {
                uint8_t p[3];
                fetch(min(dx * ag.iscale_x + xx, ag.sw - 1), min(dy * ag.iscale_y + yy, ag.sh - 1), p);
                sum0 += p[0]; sum1 += p[1]; sum2 += p[2];
            }
        if (ag.iscale_x == 2 && ag.iscale_y == 2) {
            out[0] = (uint8_t)((sum0 + 2) >> 2); out[1] = (uint8_t)((sum1 + 2) >> 2); out[2] = (uint8_t)((sum2 + 2) >> 2);
        } else {
            out[0] = sat_u8_f((float)sum0 * ag.fast_scale); out[1] = sat_u8_f((float)sum1 * ag.fast_scale);
            out[2] = sat_u8_f((float)sum2 * ag.fast_scale);
        }
        return;
    }
    const int xb = idx[ag.xidx_ofs + dx], xe = idx[ag.xidx_ofs + dx + 1];
    const int yb = idx[ag.yidx_ofs + dy], ye = idx[ag.yidx_ofs + dy + 1];
    float s0 = 0, s1 = 0, s2 = 0;
    constexpr int XB = 8;                       // x taps fetched per batch (a shrink by s needs <= ceil(s) + 2 taps)
    if (xe - xb <= XB) {
        // common case: all x taps of a source row in ONE batch -> their gathers are in flight together.
        // Taps beyond xe get alpha 0 and a clamped (valid) source index; adding 0.f * p leaves the sum unchanged
        // bit for bit (the partial sums are non-negative, so no -0 can arise).
        AreaTap tx[XB];
#pragma unroll
        for (int k = 0; k < XB; ++k) {
            const bool in = xb + k < xe;
            tx[k] = taps[ag.xtap_ofs + (in ? xb + k : xe - 1)];
            if (!in) tx[k].alpha = 0.f;
        }
        for (int j = yb; j < ye; ++j) {
            const AreaTap ty = taps[ag.ytap_ofs + j];
            uint8_t p[XB][3];
#pragma unroll
            for (int k = 0; k < XB; ++k) fetch(tx[k].si, ty.si, p[k]);
            float b0 = 0, b1 = 0, b2 = 0;
#pragma unroll
            for (int k = 0; k < XB; ++k) {
                if (xb + k < xe) {
                    b0 = b0 + (float)p[k][0] * tx[k].alpha; b1 = b1 + (float)p[k][1] * tx[k].alpha; b2 = b2 + (float)p[k][2] * tx[k].alpha;
                }
            }
            if (j == yb) { s0 = ty.alpha * b0; s1 = ty.alpha * b1; s2 = ty.alpha * b2; }
            else { s0 += ty.alpha * b0; s1 += ty.alpha * b1; s2 += ty.alpha * b2; }
        }
    } else {
        for (int j = yb; j < ye; ++j) {
            const AreaTap ty = taps[ag.ytap_ofs + j];
            float b0 = 0, b1 = 0, b2 = 0;
            for (int kx = xb; kx < xe; ++kx) {
                const AreaTap tx = taps[ag.xtap_ofs + kx];
                uint8_t p[3];
                fetch(tx.si, ty.si, p);
                b0 = b0 + (float)p[0] * tx.alpha; b1 = b1 + (float)p[1] * tx.alpha; b2 = b2 + (float)p[2] * tx.alpha;
            }
            if (j == yb) { s0 = ty.alpha * b0; s1 = ty.alpha * b1; s2 = ty.alpha * b2; }
            else { s0 += ty.alpha * b0; s1 += ty.alpha * b1; s2 += ty.alpha * b2; }
        }
    }
    out[0] = sat_u8_f(s0); out[1] = sat_u8_f(s1); out[2] = sat_u8_f(s2);
}

// Same arithmetic, no batching (compact code) — used by reproject's rare fallback paths.
template <class Fetch>
__device__ __noinline__ void area_pixel_simple(const AreaGeom& ag, const AreaTap* __restrict__ taps, const int32_t* __restrict__ idx,
                                               int dx, int dy, Fetch fetch, uint8_t out[3]) {
    if (ag.fast) {
        int sum0 = 0, sum1 = 0, sum2 = 0;
        for (int yy = 0; yy < ag.iscale_y; ++yy)
            for (int xx = 0; xx < ag.iscale_x; ++xx) {
                uint8_t p[3];
                fetch(min(dx * ag.iscale_x + xx, ag.sw - 1), min(dy * ag.iscale_y + yy, ag.sh - 1), p);
                sum0 += p[0]; sum1 += p[1]; sum2 += p[2];
            }
        if (ag.iscale_x == 2 && ag.iscale_y == 2) {
            out[0] = (uint8_t)((sum0 + 2) >> 2); out[1] = (uint8_t)((sum1 + 2) >> 2); out[2] = (uint8_t)((sum2 + 2) >> 2);
        } else {
            out[0] = sat_u8_f((float)sum0 * ag.fast_scale); out[1] = sat_u8_f((float)sum1 * ag.fast_scale);
            out[2] = sat_u8_f((float)sum2 * ag.fast_scale);
        }
        return;
    }
    const int xb = idx[ag.xidx_ofs + dx], xe = idx[ag.xidx_ofs + dx + 1];
    const int yb = idx[ag.yidx_ofs + dy], ye = idx[ag.yidx_ofs + dy + 1];
    float s0 = 0, s1 = 0, s2 = 0;
    for (int j = yb; j < ye; ++j) {
        const AreaTap ty = taps[ag.ytap_ofs + j];
        float b0 = 0, b1 = 0, b2 = 0;
        for (int kx = xb; kx < xe; ++kx) {
            const AreaTap tx = taps[ag.xtap_ofs + kx];
            uint8_t p[3];
            fetch(tx.si, ty.si, p);
            b0 = b0 + (float)p[0] * tx.alpha; b1 = b1 + (float)p[1] * tx.alpha; b2 = b2 + (float)p[2] * tx.alpha;
        }
        if (j == yb) { s0 = ty.alpha * b0; s1 = ty.alpha * b1; s2 = ty.alpha * b2; }
        else { s0 += ty.alpha * b0; s1 += ty.alpha * b1; s2 += ty.alpha * b2; }
    }
    out[0] = sat_u8_f(s0); out[1] = sat_u8_f(s1); out[2] = sat_u8_f(s2);
}

constexpr int SM_TW = AREA_TW, SM_TH = AREA_TH;   // small-image tile per 256-thread block (32 x 8)

// to_small_image of n equally sized images.  grid (tiles, n), block 256.
__global__ __launch_bounds__(256) void small_image_kernel(AreaGeom ag, const AreaTap* __restrict__ taps,
                                                          const int32_t* __restrict__ idx,
                                                          const uint8_t* __restrict__ imgs, int64_t img_stride, int stride,
                                                          uint8_t* __restrict__ out, int64_t out_stride) {
    const int tiles_x = (ag.dw + SM_TW - 1) / SM_TW;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int dx = tx * SM_TW + (threadIdx.x & (SM_TW - 1)), dy = ty * SM_TH + (threadIdx.x / SM_TW);
    if (dx >= ag.dw || dy >= ag.dh) return;
    const uint8_t* img = imgs + (int64_t)blockIdx.y * img_stride;
    uint8_t o[3];
    area_pixel(ag, taps, idx, dx, dy, [&](int sx, int sy, uint8_t* p) {
        const uint8_t* s = img + (int64_t)sy * stride + 3 * sx;
        p[0] = s[0]; p[1] = s[1]; p[2] = s[2];
    }, o);
    uint8_t* d = out + (int64_t)blockIdx.y * out_stride + ((int64_t)dy * ag.dw + dx) * 3;
    d[0] = o[0]; d[1] = o[1]; d[2] = o[2];
}

// sum of squared differences between consecutive small images: pair i = (a_i, b_i).  grid n, block 256.
__global__ __launch_bounds__(256) void ssd_kernel(const uint8_t* __restrict__ a, int64_t a_stride,
                                                  const uint8_t* __restrict__ b, int64_t b_stride, int64_t nbytes,
                                                  unsigned long long* __restrict__ out) {
    __shared__ unsigned long long red[4];
    const uint8_t* pa = a + (int64_t)blockIdx.x * a_stride;
    const uint8_t* pb = b + (int64_t)blockIdx.x * b_stride;
    unsigned long long s = 0;
    for (int64_t i = threadIdx.x; i < nbytes; i += 256) { int d = (int)pa[i] - (int)pb[i]; s += (unsigned)(d * d); }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// reproject_kernel: grid (tiles, max_rated, B), block 256.
// For survivor s of frame f: warp the frame into the slide's space (nearest, inverse map,
// 10-bit fixed point, [OCV A.10]), area-resize to the slide's small size, accumulate the
// squared difference against the slide's small image.
// The fixed-point warp terms of imgwarp.cpp — adelta[x], bdelta[x] per source column and X0[y], Y0[y] per
// source row — are computed once per block into LDS for the source window of the block's 32x8 small pixels,
// so the per-tap work is two LDS reads, two adds/shifts and one 3-byte gather.
constexpr int RP_SPAN_X = 512, RP_SPAN_Y = 160;
#ifndef RP_WIN_KB
#define RP_WIN_KB 32
#endif
constexpr int RP_WIN_BYTES = RP_WIN_KB * 1024;      // LDS window of the frame, 4 bytes per pixel (B,G,R,0)
constexpr int RP_PRE = 6;                    // window groups (of 4 pixels) per thread that are fetched one tile ahead
constexpr int RP_MAX_TILES = 64;             // tiles per strip with a precomputed descriptor (small images up to 2048 px wide)
constexpr int RP_XY_MAX = 6144;              // PERSP: source pixels of a tile whose warped coordinates are tabled in LDS (141 x 37 for 2001 -> 461)
struct RpTile { int sx_lo, sx_hi, tabled, windowed, all_in, wx0, wy0, wpitch, npx4, nrows; };

// warpPerspective(nearest, WARP_INVERSE_MAP), imgproc/src/imgwarp.cpp WarpPerspectiveInvoker (recalled; the oracle's
// WarpSampler::src_xy, persp form): the destination is walked in blocks of bw0 columns and the block origin enters the
// floating-point association: X0 = M0 xb + M1 y + M2, W = W0 + M6 x1, W = W ? 1/W : 0, X = saturate(clamp((X0 + M0 x1) W)).
__device__ __forceinline__ void persp_src(const double (&M)[9], int bw0, int x, int y, int& X, int& Y) {
    const int xb = (x / bw0) * bw0, x1 = x - xb;
    const double X0 = M[0] * xb + M[1] * y + M[2], Y0 = M[3] * xb + M[4] * y + M[5], W0 = M[6] * xb + M[7] * y + M[8];
    double W = W0 + M[6] * x1;
    W = W != 0.0 ? 1. / W : 0;
    const double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + M[0] * x1) * W));
    const double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + M[3] * x1) * W));
    int Xi = (int)rint(fX), Yi = (int)rint(fY);
    X = Xi < -32768 ? -32768 : (Xi > 32767 ? 32767 : Xi);
    Y = Yi < -32768 ? -32768 : (Yi > 32767 ? 32767 : Yi);
}

// grid (max tile rows, min(B, 65535)), block 256.  Block (ty, y) walks the pairs y, y + gridDim.y, ... of the compact
// list built by rate_kernel and, for each, the whole row `ty` of 32x8 output tiles: the per-pair constants are
// fetched once (one flat descriptor instead of a chain of dependent loads) and amortised over the strip.
// PERSP (verify_model 1): the pair's transform is a 3x3 homography and the warp is warpPerspective.  The separable tables
// (adelta / bdelta per column, X0 / Y0 per row) do not exist for a projective map; instead the warped coordinate of EVERY
// source pixel of the tile's span is computed once (one f64 division each) into an LDS table of packed shorts, which the
// taps index — a source pixel is a tap of up to four output pixels.  Window, prefetch and strip structure are shared.
template <bool PERSP>
__global__ __launch_bounds__(256) void reproject_kernel(const AreaGeom* __restrict__ ags, const AreaTap* __restrict__ taps,
                                                        const int32_t* __restrict__ idx,
                                                        const uint8_t* __restrict__ page_small,
                                                        const uint8_t* __restrict__ frames, int64_t frame_stride, int stride,
                                                        int fw, int fh, FrameCands* __restrict__ fcs,
                                                        const PairDesc* __restrict__ pair_list, const uint32_t* __restrict__ pair_count) {
    __shared__ unsigned long long red[4];
    __shared__ int s_tab[PERSP ? RP_XY_MAX : 2 * RP_SPAN_X + 2 * RP_SPAN_Y];
    int* const s_ad = s_tab; int* const s_bd = s_tab + RP_SPAN_X; int* const s_x0 = s_tab + 2 * RP_SPAN_X; int* const s_y0 = s_tab + 2 * RP_SPAN_X + RP_SPAN_Y;
    uint32_t* const s_xy = reinterpret_cast<uint32_t*>(s_tab);       // PERSP: (X & 0xFFFF) | Y << 16 per source pixel of the tile span
    __shared__ __attribute__((aligned(16))) uint8_t win[RP_WIN_BYTES + 16];   // + a zero pixel at RP_WIN_BYTES
    __shared__ RpTile s_tile[RP_MAX_TILES];
    const uint32_t npairs = *pair_count;
    if (threadIdx.x < 4) reinterpret_cast<uint32_t*>(win + RP_WIN_BYTES)[threadIdx.x] = 0;
    const int AB_BITS = 10, AB_SCALE = 1 << AB_BITS, round_delta = AB_SCALE / 2;
    for (uint32_t pi = blockIdx.y; pi < npairs; pi += gridDim.y) {
        const PairDesc pd = pair_list[pi];
        const int f = pd.f;
        const AreaGeom ag = ags[pd.area_idx];
        const int tiles_x = (ag.dw + SM_TW - 1) / SM_TW, tiles_y = (ag.dh + SM_TH - 1) / SM_TH;
        if (ag.vt_ok && tiles_x <= RP_MAX_TILES) continue;            // (uniform per block) reproject_vt_kernel's pair
        const int ty = blockIdx.x;
        if (ty >= tiles_y) continue;                                  // uniform per block
        double M[PERSP ? 9 : 6];
#pragma unroll
        for (int j = 0; j < (PERSP ? 9 : 6); ++j) M[j] = pd.M[j];
        [[maybe_unused]] const int bw0 = max(1, min(1024 / max(min(16, ag.sh), 1), ag.sw));   // PERSP: WarpPerspectiveInvoker's block width
        const uint8_t* frame = frames + (int64_t)f * frame_stride;
        const int dy = ty * SM_TH + (threadIdx.x / SM_TW);
        const int dya = ty * SM_TH, dyb = min(ag.dh, dya + SM_TH) - 1;
        int sy_lo, sy_hi;
        if (ag.fast) { sy_lo = dya * ag.iscale_y; sy_hi = min((dyb + 1) * ag.iscale_y - 1, ag.sh - 1); }
        else { sy_lo = taps[ag.ytap_ofs + idx[ag.yidx_ofs + dya]].si; sy_hi = taps[ag.ytap_ofs + idx[ag.yidx_ofs + dyb + 1] - 1].si; }
        const bool tabled_y = PERSP ? true : (sy_hi - sy_lo) < RP_SPAN_Y;
        [[maybe_unused]] const int sph = sy_hi - sy_lo + 1;
        __syncthreads();                                              // previous pair's readers of the LDS tables are done
        if (!PERSP && tabled_y)
            for (int i = threadIdx.x; i <= sy_hi - sy_lo; i += 256) {
                const int y = sy_lo + i;
                s_x0[i] = sat_int_d((M[1] * y + M[2]) * AB_SCALE) + round_delta;
                s_y0[i] = sat_int_d((M[4] * y + M[5]) * AB_SCALE) + round_delta;
            }
        // ---- per-strip tile descriptors, one thread per tile: source span, frame-space bounding box of its taps and the
        // LDS window layout.  (Computed per tile inside the loop, the two dependent table loads and the corner arithmetic
        // sat on every tile's critical path, and the window could not be requested before the tile began.)
        __syncthreads();                                              // s_x0 / s_y0 are complete
        if ((int)threadIdx.x < tiles_x && (int)threadIdx.x < RP_MAX_TILES) {
            const int tx = threadIdx.x;
            RpTile T{};
            const int dxa = tx * SM_TW, dxb = min(ag.dw, dxa + SM_TW) - 1;
            if (ag.fast) { T.sx_lo = dxa * ag.iscale_x; T.sx_hi = min((dxb + 1) * ag.iscale_x - 1, ag.sw - 1); }
            else { T.sx_lo = taps[ag.xtap_ofs + idx[ag.xidx_ofs + dxa]].si; T.sx_hi = taps[ag.xtap_ofs + idx[ag.xidx_ofs + dxb + 1] - 1].si; }
            if constexpr (PERSP) {
                T.tabled = (T.sx_hi - T.sx_lo + 1) * sph <= RP_XY_MAX;
                // The denominator is linear in (x, y): if it keeps its sign at the four corners of the source span it keeps it
                // inside, the map is then projective on the whole rectangle (lines to lines, convex to convex) and the corner
                // images bound every tap (+- 1 px for the rounding of each coordinate).
                const double w00 = M[6] * T.sx_lo + M[7] * sy_lo + M[8], w10 = M[6] * T.sx_hi + M[7] * sy_lo + M[8];
                const double w01 = M[6] * T.sx_lo + M[7] * sy_hi + M[8], w11 = M[6] * T.sx_hi + M[7] * sy_hi + M[8];
                const bool same = (w00 > 0 && w10 > 0 && w01 > 0 && w11 > 0) || (w00 < 0 && w10 < 0 && w01 < 0 && w11 < 0);
                if (T.tabled && same) {
                    int X00, Y00, X10, Y10, X01, Y01, X11, Y11;
                    persp_src(M, bw0, T.sx_lo, sy_lo, X00, Y00); persp_src(M, bw0, T.sx_hi, sy_lo, X10, Y10);
                    persp_src(M, bw0, T.sx_lo, sy_hi, X01, Y01); persp_src(M, bw0, T.sx_hi, sy_hi, X11, Y11);
                    const int ux0 = min(min(X00, X10), min(X01, X11)) - 1, ux1 = max(max(X00, X10), max(X01, X11)) + 1;
                    const int uy0 = min(min(Y00, Y10), min(Y01, Y11)) - 1, uy1 = max(max(Y00, Y10), max(Y01, Y11)) + 1;
                    const int bx0 = max(ux0, 0), bx1 = min(ux1, fw - 1);
                    const int by0 = max(uy0, 0), by1 = min(uy1, fh - 1);
                    if (bx1 >= bx0 && by1 >= by0) {
                        const int px0 = bx0 & ~3, npx4 = (bx1 - px0 + 4) >> 2;
                        const int wpitch = npx4 * 16, nrows = by1 - by0 + 1;
                        if ((int64_t)wpitch * nrows <= RP_WIN_BYTES && (((uintptr_t)frames | (uintptr_t)frame_stride | (uintptr_t)stride) & 3) == 0) {
                            T.windowed = 1; T.wx0 = px0; T.wy0 = by0; T.wpitch = wpitch; T.npx4 = npx4; T.nrows = nrows;
                        }
                    }
                }
            } else {
            T.tabled = tabled_y && (T.sx_hi - T.sx_lo) < RP_SPAN_X;
            if (T.tabled) {
                // X, Y are monotone in x and in y, so the corners of the source span bound every tap
                const int ny = sy_hi - sy_lo;
                const int ad0 = sat_int_d(M[0] * T.sx_lo * AB_SCALE), adn = sat_int_d(M[0] * T.sx_hi * AB_SCALE);
                const int bd0 = sat_int_d(M[3] * T.sx_lo * AB_SCALE), bdn = sat_int_d(M[3] * T.sx_hi * AB_SCALE);
                auto XY = [&](int ad, int bd, int iy, int& X, int& Y) {
                    X = (int)((uint32_t)s_x0[iy] + (uint32_t)ad) >> AB_BITS;
                    Y = (int)((uint32_t)s_y0[iy] + (uint32_t)bd) >> AB_BITS;
                };
                int X00, Y00, X10, Y10, X01, Y01, X11, Y11;
                XY(ad0, bd0, 0, X00, Y00); XY(adn, bdn, 0, X10, Y10); XY(ad0, bd0, ny, X01, Y01); XY(adn, bdn, ny, X11, Y11);
                const int ux0 = min(min(X00, X10), min(X01, X11)), ux1 = max(max(X00, X10), max(X01, X11));
                const int uy0 = min(min(Y00, Y10), min(Y01, Y11)), uy1 = max(max(Y00, Y10), max(Y01, Y11));
                T.all_in = ux0 >= 0 && ux1 < fw && uy0 >= 0 && uy1 < fh;      // no tap of this tile leaves the frame
                const int bx0 = max(ux0, 0), bx1 = min(ux1, fw - 1);
                const int by0 = max(uy0, 0), by1 = min(uy1, fh - 1);
                if (bx1 >= bx0 && by1 >= by0) {
                    // window columns start at a multiple of 4 pixels so that 4 pixels = 3 aligned source dwords;
                    // stored as 4 bytes per pixel (B,G,R,0): one aligned ds_read_b32 per tap later
                    const int px0 = bx0 & ~3, npx4 = (bx1 - px0 + 4) >> 2;        // groups of 4 pixels per row
                    const int wpitch = npx4 * 16, nrows = by1 - by0 + 1;
                    if ((int64_t)wpitch * nrows <= RP_WIN_BYTES && (((uintptr_t)frames | (uintptr_t)frame_stride | (uintptr_t)stride) & 3) == 0) {
                        T.windowed = 1; T.wx0 = px0; T.wy0 = by0; T.wpitch = wpitch; T.npx4 = npx4; T.nrows = nrows;
                    }
                }
            }
            }
            s_tile[tx] = T;
        }
        __syncthreads();
        // window groups of a tile: RP_PRE per thread travel through registers one tile ahead, the rest (large windows)
        // are loaded when the tile starts
        const int last_dw = (stride >> 2) - 1;                        // stay inside the row allocation
        auto load_group = [&](const RpTile& T, int i, uint32_t& a, uint32_t& b, uint32_t& c) {
            // i / npx4 without the integer-division sequence: i < 2^11, so the f32 product is off by < (2^11 / npx4) * 2^-22,
            // far below the 0.5 / npx4 distance of (i + 0.5) / npx4 to the nearest integer
            const int ry = (int)(((float)i + 0.5f) * (1.0f / (float)max(T.npx4, 1))), g = i - __mul24(ry, T.npx4);
            const uint32_t* row = reinterpret_cast<const uint32_t*>(frame + (int64_t)(T.wy0 + ry) * stride);
            const int d0 = ((T.wx0 >> 2) + g) * 3;
            a = row[min(d0, last_dw)]; b = row[min(d0 + 1, last_dw)]; c = row[min(d0 + 2, last_dw)];
        };
        auto store_group = [&](int i, uint32_t a, uint32_t b, uint32_t c) {
            uint4 o4;                                                 // b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3
            o4.x = a & 0x00FFFFFFu;
            o4.y = (a >> 24) | ((b & 0xFFFFu) << 8);
            o4.z = (b >> 16) | ((c & 0xFFu) << 16);
            o4.w = c >> 8;
            reinterpret_cast<uint4*>(win)[i] = o4;
        };
        uint32_t pa[RP_PRE], pb[RP_PRE], pc[RP_PRE];
        auto prefetch = [&](const RpTile& T) {                        // (always issued, clamped: `p*` stay plain registers)
            const int total = max(T.npx4 * T.nrows, 1);
#pragma unroll
            for (int u = 0; u < RP_PRE; ++u) load_group(T, min((int)threadIdx.x + 256 * u, total - 1), pa[u], pb[u], pc[u]);
        };
        const bool use_tiles = tiles_x <= RP_MAX_TILES;
        prefetch(s_tile[0]);
        unsigned long long acc = 0;
        for (int tx = 0; tx < tiles_x; ++tx) {
            const int dx = tx * SM_TW + (threadIdx.x & (SM_TW - 1));
            const RpTile T = s_tile[min(tx, RP_MAX_TILES - 1)];
            const int sx_lo = T.sx_lo, sx_hi = T.sx_hi;
            const bool tabled = use_tiles && T.tabled;
            const bool windowed = tabled && T.windowed, all_in = T.all_in;
            const int wx0 = T.wx0, wy0 = T.wy0, wpitch = T.wpitch;
            __syncthreads();                                          // previous tile's readers of s_ad / win are done
            [[maybe_unused]] const int spw = sx_hi - sx_lo + 1;
            if constexpr (PERSP) {
                if (tabled) {
                    const float inv = 1.0f / (float)spw;
                    for (int i = threadIdx.x; i < spw * sph; i += 256) {
                        int yy = (int)(((float)i + 0.5f) * inv);          // i / spw (i < 2^13: exact, see load_group)
                        yy -= (yy * spw > i) ? 1 : 0; yy += ((yy + 1) * spw <= i) ? 1 : 0;
                        int X, Y;
                        persp_src(M, bw0, sx_lo + (i - yy * spw), sy_lo + yy, X, Y);
                        s_xy[i] = ((uint32_t)X & 0xFFFFu) | ((uint32_t)Y << 16);
                    }
                }
            } else {
            if (tabled)
                for (int i = threadIdx.x; i <= sx_hi - sx_lo; i += 256) {
                    const int x = sx_lo + i;
                    s_ad[i] = sat_int_d(M[0] * x * AB_SCALE); s_bd[i] = sat_int_d(M[3] * x * AB_SCALE);
                }
            }
            if (windowed) {
                const int total = T.npx4 * T.nrows;
#pragma unroll
                for (int u = 0; u < RP_PRE; ++u) {
                    const int i = (int)threadIdx.x + 256 * u;
                    if (i < total) store_group(i, pa[u], pb[u], pc[u]);
                }
                for (int i = (int)threadIdx.x + 256 * RP_PRE; i < total; i += 256) {      // windows beyond RP_PRE * 256 groups
                    uint32_t a, b, c;
                    load_group(T, i, a, b, c);
                    store_group(i, a, b, c);
                }
            }
            prefetch(s_tile[min(tx + 1, min(tiles_x, RP_MAX_TILES) - 1)]);   // in flight during this tile's taps
            __syncthreads();
            if (dx < ag.dw && dy < ag.dh) {
                uint8_t o[3];
                const int xb = ag.fast ? 0 : idx[ag.xidx_ofs + dx], xe = ag.fast ? 0 : idx[ag.xidx_ofs + dx + 1];
                if constexpr (PERSP) {
                    if (windowed && !ag.fast && ag.max_xtaps <= 8) {
                        // LDS coordinate table + LDS window: per tap one table read, the in-frame test and one aligned 4-byte
                        // window read (out-of-frame taps read the zero pixel; padding taps repeat the last real tap with weight 0)
                        const int yb = idx[ag.yidx_ofs + dy], ye = idx[ag.yidx_ofs + dy + 1];
                        auto run = [&](auto xb_tag) {
                            constexpr int XB = decltype(xb_tag)::value;
                            float al[XB]; int xo[XB];
#pragma unroll
                            for (int k = 0; k < XB; ++k) {
                                const bool in = xb + k < xe;
                                const AreaTap t = taps[ag.xtap_ofs + (in ? xb + k : xe - 1)];
                                al[k] = in ? t.alpha : 0.f;
                                xo[k] = t.si - sx_lo;
                            }
                            float s0 = 0, s1 = 0, s2 = 0;
                            const int wbase = -wy0 * wpitch - 4 * wx0;
                            for (int j = yb; j < ye; ++j) {
                                const AreaTap tyv = taps[ag.ytap_ofs + j];
                                const int rowb = (tyv.si - sy_lo) * spw;
                                float b0 = 0, b1 = 0, b2 = 0;
#pragma unroll
                                for (int k = 0; k < XB; ++k) {
                                    const uint32_t e = s_xy[rowb + xo[k]];
                                    const int X = (int)(short)(e & 0xFFFFu), Y = (int)e >> 16;
                                    int ofs = (int)__mul24(Y, wpitch) + 4 * X + wbase;
                                    ofs = ((unsigned)X < (unsigned)fw && (unsigned)Y < (unsigned)fh) ? ofs : RP_WIN_BYTES;
                                    const uint32_t px = *reinterpret_cast<const uint32_t*>(win + ofs);
                                    b0 = b0 + (float)(px & 255u) * al[k]; b1 = b1 + (float)((px >> 8) & 255u) * al[k]; b2 = b2 + (float)((px >> 16) & 255u) * al[k];
                                }
                                if (j == yb) { s0 = tyv.alpha * b0; s1 = tyv.alpha * b1; s2 = tyv.alpha * b2; }
                                else { s0 += tyv.alpha * b0; s1 += tyv.alpha * b1; s2 += tyv.alpha * b2; }
                            }
                            o[0] = sat_u8_f(s0); o[1] = sat_u8_f(s1); o[2] = sat_u8_f(s2);
                        };
                        using T6 = std::integral_constant<int, 6>; using T8 = std::integral_constant<int, 8>;
                        if (ag.max_xtaps <= 6) run(T6{}); else run(T8{});
                    } else {
                        area_pixel_simple(ag, taps, idx, dx, dy, [&](int x, int y, uint8_t* p) {
                            int X, Y;
                            if (tabled) { const uint32_t e = s_xy[(y - sy_lo) * spw + (x - sx_lo)]; X = (int)(short)(e & 0xFFFFu); Y = (int)e >> 16; }
                            else persp_src(M, bw0, x, y, X, Y);
                            if ((unsigned)X < (unsigned)fw && (unsigned)Y < (unsigned)fh) {
                                const uint8_t* sp = frame + (int64_t)Y * stride + 3 * X;
                                p[0] = sp[0]; p[1] = sp[1]; p[2] = sp[2];
                            } else { p[0] = p[1] = p[2] = 0; }
                        }, o);
                    }
                } else {
                if (windowed && !ag.fast && ag.max_xtaps <= 8) {
                    // common case: LDS tables + LDS window.  Per output pixel: x-tap terms once, then per source row
                    // 2 LDS reads and per tap 2 add/shift pairs, a bounds test and ONE aligned 4-byte LDS read.
                    // Out-of-frame taps and the padding taps (alpha 0) read the zero pixel kept at the end of `win`.
                    const int yb = idx[ag.yidx_ofs + dy], ye = idx[ag.yidx_ofs + dy + 1];
                    // CHECK = false: every tap of the tile is inside the frame (all_in), so the in-frame test and the
                    // zero-pixel select drop out; padding taps then repeat the last real tap with weight 0.
                    auto run = [&](auto xb_tag, auto check_tag) {
                        constexpr int XB = decltype(xb_tag)::value;
                        constexpr bool CHECK = decltype(check_tag)::value;
                        float al[XB]; int adk[XB], bdk[XB];
#pragma unroll
                        for (int k = 0; k < XB; ++k) {
                            const bool in = xb + k < xe;
                            const AreaTap t = taps[ag.xtap_ofs + (in ? xb + k : xe - 1)];
                            al[k] = in ? t.alpha : 0.f;
                            adk[k] = s_ad[t.si - sx_lo];
                            bdk[k] = (in || !CHECK) ? s_bd[t.si - sx_lo] : (int)0x40000000;   // CHECK: padding taps land far outside the frame
                        }
                        float s0 = 0, s1 = 0, s2 = 0;
                        const int wbase = -wy0 * wpitch - 4 * wx0;
                        for (int j = yb; j < ye; ++j) {
                            const AreaTap tyv = taps[ag.ytap_ofs + j];
                            const int X0 = s_x0[tyv.si - sy_lo], Y0 = s_y0[tyv.si - sy_lo];
                            float b0 = 0, b1 = 0, b2 = 0;
#pragma unroll
                            for (int k = 0; k < XB; ++k) {
                                const int X = (int)((uint32_t)X0 + (uint32_t)adk[k]) >> AB_BITS;
                                const int Y = (int)((uint32_t)Y0 + (uint32_t)bdk[k]) >> AB_BITS;
                                // (saturate_cast<short> of imgwarp.cpp cannot change the in-frame test for frames < 32768 px)
                                int ofs = (int)__mul24(Y, wpitch) + 4 * X + wbase;
                                if (CHECK) ofs = ((unsigned)X < (unsigned)fw && (unsigned)Y < (unsigned)fh) ? ofs : RP_WIN_BYTES;
                                const uint32_t px = *reinterpret_cast<const uint32_t*>(win + ofs);
                                b0 = b0 + (float)(px & 255u) * al[k]; b1 = b1 + (float)((px >> 8) & 255u) * al[k]; b2 = b2 + (float)((px >> 16) & 255u) * al[k];
                            }
                            if (j == yb) { s0 = tyv.alpha * b0; s1 = tyv.alpha * b1; s2 = tyv.alpha * b2; }
                            else { s0 += tyv.alpha * b0; s1 += tyv.alpha * b1; s2 += tyv.alpha * b2; }
                        }
                        o[0] = sat_u8_f(s0); o[1] = sat_u8_f(s1); o[2] = sat_u8_f(s2);
                    };
                    using T6 = std::integral_constant<int, 6>; using T8 = std::integral_constant<int, 8>;
                    if (all_in) { if (ag.max_xtaps <= 6) run(T6{}, std::false_type{}); else run(T8{}, std::false_type{}); }
                    else
                    { if (ag.max_xtaps <= 6) run(T6{}, std::true_type{}); else run(T8{}, std::true_type{}); }
                } else {
                    area_pixel_simple(ag, taps, idx, dx, dy, [&](int x, int y, uint8_t* p) {
                        const int adelta = sat_int_d(M[0] * x * AB_SCALE), bdelta = sat_int_d(M[3] * x * AB_SCALE);
                        const int X0 = sat_int_d((M[1] * y + M[2]) * AB_SCALE) + round_delta;
                        const int Y0 = sat_int_d((M[4] * y + M[5]) * AB_SCALE) + round_delta;
                        int X = (int)((uint32_t)X0 + (uint32_t)adelta) >> AB_BITS;
                        int Y = (int)((uint32_t)Y0 + (uint32_t)bdelta) >> AB_BITS;
                        X = X < -32768 ? -32768 : (X > 32767 ? 32767 : X);
                        Y = Y < -32768 ? -32768 : (Y > 32767 ? 32767 : Y);
                        if ((unsigned)X < (unsigned)fw && (unsigned)Y < (unsigned)fh) {
                            const uint8_t* sp = frame + (int64_t)Y * stride + 3 * X;
                            p[0] = sp[0]; p[1] = sp[1]; p[2] = sp[2];
                        } else { p[0] = p[1] = p[2] = 0; }
                    }, o);
                }
                }
                const uint8_t* ref = page_small + pd.small_ofs + ((int64_t)dy * ag.dw + dx) * 3;
                int d0 = (int)o[0] - ref[0], d1 = (int)o[1] - ref[1], d2 = (int)o[2] - ref[2];
                acc += (unsigned)(d0 * d0 + d1 * d1 + d2 * d2);
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(&fcs[f].ssd[pd.s], red[0] + red[1] + red[2] + red[3]);
    }
}

// reproject_vt_kernel: the same sum through a tile of the WARPED image instead of a window of the frame.
// reproject_kernel evaluates the warp per TAP (a small pixel of a 2001 -> 461 shrink has up to 6 x 6 taps: 9216 per tile, each
// with its own fixed-point coordinate pair, window offset and in-frame test), although the taps of a tile are only the
// spw x sph (141 x 37 = 5217) pixels of its source span, most of them shared by no one but a few by two outputs.  Here
//   pass 1  warps every source pixel of the span ONCE: nearest frame pixel -> one unaligned 4-byte global load (B, G, R, x),
//           issued one tile ahead into registers (VT_CG x VT_RI per thread), stored as a dword of the LDS tile `vt`;
//   pass 2  is INTER_AREA over that tile: the taps of an output are consecutive dwords of consecutive rows (geom.h checks
//           it), so a row of taps is three ds_read2_b32 at immediate offsets and the per-tap work is the arithmetic only —
//           3 conversions, 3 multiplies, 3 adds, same order as ResizeArea_Invoker ([OCV A.11]).
// Per tile ~470 VALU instructions per thread instead of ~650, 24 KB of LDS instead of 41.  Pairs of a size class outside the
// limits (AreaGeom::vt_ok: shrink factors above ~5) stay with reproject_kernel; each kernel skips the other's pairs.
// grid / pair walk / strip structure: as reproject_kernel.
struct VtTile { int sx_lo, spw, all_in; };

#ifndef VT_WAVES_EU
#define VT_WAVES_EU 4      /* 128 registers: four blocks per CU (3: 1.58 ms, 4: 1.37 ms for the headline launch) */
#endif
#ifdef VT_VGPRS            /* experiment hook: hard register cap (amdgpu_num_vgpr counts in units of two) */
#define VT_KERNEL_ATTR __attribute__((amdgpu_num_vgpr(VT_VGPRS)))
#else
#define VT_KERNEL_ATTR
#endif
template <bool PERSP>
__global__ VT_KERNEL_ATTR __launch_bounds__(256, VT_WAVES_EU) void reproject_vt_kernel(const AreaGeom* __restrict__ ags, const AreaTap* __restrict__ taps,
                                                           const int32_t* __restrict__ idx, const AreaRec* __restrict__ recs,
                                                           const uint8_t* __restrict__ page_small,
                                                           const uint8_t* __restrict__ frames, int64_t frame_stride, int stride,
                                                           int fw, int fh, FrameCands* __restrict__ fcs,
                                                           const PairDesc* __restrict__ pair_list, const uint32_t* __restrict__ pair_count) {
#ifdef VERIFY_PRIO       /* compile-time experiment hook (tools/ab_matrix.sh): the re-projection's waves at instruction priority VERIFY_PRIO */
    __builtin_amdgcn_s_setprio(VERIFY_PRIO);
#endif
    __shared__ unsigned long long red[4];
    __shared__ uint32_t vt[VT_PX + 8];                                // + slack: a padding tap (weight 0) may read past the last pixel
    __shared__ int2 s_col[2][32 * VT_CG];                             // (adelta, bdelta) per source column of a tile, double buffered
    __shared__ int2 s_rowt[8 * VT_RI];                                // (X0, Y0) fixed-point row terms per source row of the strip
    __shared__ VtTile s_tile[RP_MAX_TILES];
    const uint32_t npairs = *pair_count;
    constexpr int AB_BITS = 10, AB_SCALE = 1 << AB_BITS, round_delta = AB_SCALE / 2;
    constexpr int NP = VT_CG * VT_RI;
    const int c32 = threadIdx.x & 31, hw = threadIdx.x >> 5;         // pass 1: column within a 32-column group, row within an 8-row step
    if (threadIdx.x < 8) vt[VT_PX + threadIdx.x] = 0;
    for (uint32_t pi = blockIdx.y; pi < npairs; pi += gridDim.y) {
        const PairDesc pd = pair_list[pi];
        const AreaGeom ag = ags[pd.area_idx];
        if (!ag.vt_ok) continue;                                      // (uniform per block) reproject_kernel's pair
        const int tiles_x = (ag.dw + SM_TW - 1) / SM_TW, tiles_y = (ag.dh + SM_TH - 1) / SM_TH;
        const int ty = blockIdx.x;
        if (ty >= tiles_y || tiles_x > RP_MAX_TILES) continue;        // (uniform per block; wider small images: reproject_kernel)
        const int f = pd.f;
        double M[PERSP ? 9 : 6];
#pragma unroll
        for (int j = 0; j < (PERSP ? 9 : 6); ++j) M[j] = pd.M[j];
        [[maybe_unused]] const int bw0 = max(1, min(1024 / max(min(16, ag.sh), 1), ag.sw));   // PERSP: WarpPerspectiveInvoker's block width
        const uint8_t* frame = frames + (int64_t)f * frame_stride;
        const int dy = ty * SM_TH + (threadIdx.x / SM_TW);
        const int dya = ty * SM_TH, dyb = min(ag.dh, dya + SM_TH) - 1;
        const int sy_lo = taps[ag.ytap_ofs + idx[ag.yidx_ofs + dya]].si, sy_hi = taps[ag.ytap_ofs + idx[ag.yidx_ofs + dyb + 1] - 1].si;
        const int sph = sy_hi - sy_lo + 1;
        // the rows of this thread's small pixel (the same for every tile of the strip): weights, count and the first row of the
        // span; rows are consecutive (geom.h), a row past the count repeats the last one with weight 0 (s + 0 * b = s exactly)
        constexpr int YB = 7;
        float be[YB]; int ny, ry0;
        {
            const uint4* rp = reinterpret_cast<const uint4*>(recs + ag.yrec_ofs + min(dy, ag.dh - 1));
            const uint4 r0 = rp[0], r1 = rp[1];
            ny = (int)(r0.x >> 24); ry0 = (int)(r0.x & 0xFFFFFFu) - sy_lo;
            be[0] = __uint_as_float(r0.y); be[1] = __uint_as_float(r0.z); be[2] = __uint_as_float(r0.w);
            be[3] = __uint_as_float(r1.x); be[4] = __uint_as_float(r1.y); be[5] = __uint_as_float(r1.z); be[6] = __uint_as_float(r1.w);
        }
        const int ny_wave = __builtin_amdgcn_readfirstlane(max(ny, __shfl_xor(ny, 32)));      // (a wave holds two rows of small pixels)
        // this thread's source rows (hw, hw + 8, ...) and their fixed-point row terms; rows past the span repeat its last one
        // (their pixels are fetched and dropped)
        // (kept in LDS, not in registers: ten registers that would be live through the taps of every tile, where the prefetched
        // pixels, the weights and a row of taps already fill the 128 the launch bounds allow — a spill there puts a scratch
        // reload, and with it a wait for the whole prefetch, into the tap rows)
        __syncthreads();                                              // previous pair's readers of s_tile / s_col / s_rowt / vt are done
        if constexpr (!PERSP) {
            if ((int)threadIdx.x < 8 * VT_RI) {
                const int y = sy_lo + min((int)threadIdx.x, sph - 1);
                s_rowt[threadIdx.x] = make_int2(sat_int_d((M[1] * y + M[2]) * AB_SCALE) + round_delta,
                                                sat_int_d((M[4] * y + M[5]) * AB_SCALE) + round_delta);
            }
        }
        if ((int)threadIdx.x < tiles_x) {
            const int tx = threadIdx.x;
            VtTile T{};
            const int dxa = tx * SM_TW, dxb = min(ag.dw, dxa + SM_TW) - 1;
            T.sx_lo = taps[ag.xtap_ofs + idx[ag.xidx_ofs + dxa]].si;
            const int sx_hi = taps[ag.xtap_ofs + idx[ag.xidx_ofs + dxb + 1] - 1].si;
            T.spw = sx_hi - T.sx_lo + 1;
            if constexpr (!PERSP) {
                // X, Y are monotone in x and in y: the corners of the source span bound every pixel's image.  all_in: no in-frame
                // test needed — and none of them is the frame's last pixel, whose 4-byte load would end one byte past the frame
                auto XY = [&](int x, int y, int& X, int& Y) {
                    X = (int)((uint32_t)(sat_int_d((M[1] * y + M[2]) * AB_SCALE) + round_delta) + (uint32_t)sat_int_d(M[0] * x * AB_SCALE)) >> AB_BITS;
                    Y = (int)((uint32_t)(sat_int_d((M[4] * y + M[5]) * AB_SCALE) + round_delta) + (uint32_t)sat_int_d(M[3] * x * AB_SCALE)) >> AB_BITS;
                };
                int X00, Y00, X10, Y10, X01, Y01, X11, Y11;
                XY(T.sx_lo, sy_lo, X00, Y00); XY(sx_hi, sy_lo, X10, Y10); XY(T.sx_lo, sy_hi, X01, Y01); XY(sx_hi, sy_hi, X11, Y11);
                const int ux0 = min(min(X00, X10), min(X01, X11)), ux1 = max(max(X00, X10), max(X01, X11));
                const int uy0 = min(min(Y00, Y10), min(Y01, Y11)), uy1 = max(max(Y00, Y10), max(Y01, Y11));
                T.all_in = ux0 >= 0 && ux1 < fw && uy0 >= 0 && uy1 < fh && !(ux1 == fw - 1 && uy1 == fh - 1);
            }
            s_tile[tx] = T;
        }
        __syncthreads();
        [[maybe_unused]] auto col_table = [&](int tile, int buf) {   // (adelta, bdelta) of the tile's columns
            if (tile < tiles_x) {
                const VtTile T = s_tile[tile];
                for (int i = threadIdx.x; i < T.spw; i += 256) {
                    const int x = T.sx_lo + i;
                    s_col[buf][i] = make_int2(sat_int_d(M[0] * x * AB_SCALE), sat_int_d(M[3] * x * AB_SCALE));
                }
            }
        };
        if constexpr (!PERSP) { col_table(0, 0); col_table(1, 1); __syncthreads(); }
        // pass 1, issue side: the tile's pixels into registers.  oob / lastp: one bit per register — the pixel lies outside the
        // frame (stored as 0, the load goes to the frame's first pixel) / is the frame's last pixel (loaded one byte early).
        uint32_t P[NP], oob = 0, lastp = 0;
        const int lim = max(fh * stride - 4, 0);
        auto fetch_tile = [&](int tile) {
            const VtTile T = s_tile[tile];
            oob = 0; lastp = 0;
#pragma unroll
            for (int cg = 0; cg < VT_CG; ++cg) {
                if (cg * 32 >= T.spw) continue;                         // (uniform)
                const int col = min(cg * 32 + c32, T.spw - 1);
                [[maybe_unused]] int2 ab = make_int2(0, 0);
                if constexpr (!PERSP) ab = s_col[tile & 1][col];
#pragma unroll
                for (int ri = 0; ri < VT_RI; ++ri) {
                    if (8 * ri >= sph) continue;                        // (uniform)
                    const int k = cg * VT_RI + ri;
                    int X, Y;
                    if constexpr (PERSP) persp_src(M, bw0, T.sx_lo + col, sy_lo + min(hw + 8 * ri, sph - 1), X, Y);
                    else {
                        const int2 rt = s_rowt[hw + 8 * ri];              // (rows past the span hold its last row's terms)
                        X = (int)((uint32_t)rt.x + (uint32_t)ab.x) >> AB_BITS;
                        Y = (int)((uint32_t)rt.y + (uint32_t)ab.y) >> AB_BITS;
                        // (saturate_cast<short> of imgwarp.cpp cannot change the in-frame test for frames < 32768 px)
                    }
                    int o = (int)__mul24(Y, stride) + 3 * X;
                    if (PERSP || !T.all_in) {
                        const bool in = (unsigned)X < (unsigned)fw && (unsigned)Y < (unsigned)fh;
                        o = in ? o : 0;
                        oob |= in ? 0u : (1u << k);
                        lastp |= o > lim ? (1u << k) : 0u;
                        o = min(o, lim);
                    }
                    __builtin_memcpy(&P[k], frame + (uint32_t)o, 4);
                }
            }
        };
        fetch_tile(0);
        unsigned long long acc = 0;
        for (int tx = 0; tx < tiles_x; ++tx) {
            const VtTile T = s_tile[tx];
            const int spw = T.spw;
            // this tile's small pixel of the thread: its x taps and the slide's pixel, requested now, used after barrier B
            const int dx = tx * SM_TW + (threadIdx.x & (SM_TW - 1));
            const bool live = dx < ag.dw && dy < ag.dh;
            constexpr int XBM = 7;
            const int xtaps_max = ag.max_xtaps;
            float al[XBM]; int x_first; uint32_t refpx;
            {
                const int dxc = min(dx, ag.dw - 1);
                const uint4* rp = reinterpret_cast<const uint4*>(recs + ag.xrec_ofs + dxc);
                const uint4 r0 = rp[0], r1 = rp[1];
                x_first = (int)(r0.x & 0xFFFFFFu) - T.sx_lo;
                al[0] = __uint_as_float(r0.y); al[1] = __uint_as_float(r0.z); al[2] = __uint_as_float(r0.w);
                al[3] = __uint_as_float(r1.x); al[4] = __uint_as_float(r1.y); al[5] = __uint_as_float(r1.z); al[6] = __uint_as_float(r1.w);
                // (padding taps: weight 0 on whatever follows in the row)
                __builtin_memcpy(&refpx, page_small + pd.small_ofs + (uint32_t)((min(dy, ag.dh - 1) * ag.dw + dxc) * 3), 4);   // B, G, R (+ a byte of the next pixel)
            }
            __syncthreads();                                          // A: the previous tile's taps are done with vt
#pragma unroll
            for (int cg = 0; cg < VT_CG; ++cg) {
                if (cg * 32 >= spw) continue;
#pragma unroll
                for (int ri = 0; ri < VT_RI; ++ri) {
                    if (8 * ri >= sph) continue;
                    const int k = cg * VT_RI + ri, col = cg * 32 + c32, r = hw + 8 * ri;
                    uint32_t v = P[k];
                    if (PERSP || !T.all_in) { v = ((lastp >> k) & 1u) ? v >> 8 : v; v = ((oob >> k) & 1u) ? 0u : v; }
                    if (col < spw && r < sph) vt[r * spw + col] = v;
                }
            }
            if (tx + 1 < tiles_x) fetch_tile(tx + 1);                 // in flight during this tile's taps
            __syncthreads();                                          // B: vt is complete, tile tx + 1's column table has been read
            if constexpr (!PERSP) col_table(tx + 2, tx & 1);
            if (live) {
                // One instance per tap count of the class (max_xtaps; a pixel with fewer taps has weight 0 on the rest, as before):
                // a row's reads are issued together and its arithmetic is one straight run.  Testing `k >= xtaps_max` after every
                // tap cost a scalar branch and a wait for that tap's own LDS read per tap (and scalar-register spills).
                auto area_rows = [&](auto xb_tag) {
                    constexpr int XB = decltype(xb_tag)::value;
                    const uint32_t* row = vt + (x_first + ry0 * spw);
                    float s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
                    for (int j = 0; j < YB; ++j) {
                        if (j >= ny_wave) break;                            // (uniform)
                        uint32_t px[XB];
#pragma unroll
                        for (int k = 0; k < XB; ++k) px[k] = row[k];
                        // (weights and pixels are >= +0, so every product is: 0 + x = x bit for bit, and the first tap / the first row
                        // start the sums without the addition)
                        float b0 = (float)(px[0] & 255u) * al[0], b1 = (float)((px[0] >> 8) & 255u) * al[0], b2 = (float)((px[0] >> 16) & 255u) * al[0];
#pragma unroll
                        for (int k = 1; k < XB; ++k) {
                            if (PERSP && k >= xtaps_max) break;             // (PERSP: one instance, the uniform test per tap — see below)
                            b0 = b0 + (float)(px[k] & 255u) * al[k]; b1 = b1 + (float)((px[k] >> 8) & 255u) * al[k]; b2 = b2 + (float)((px[k] >> 16) & 255u) * al[k];
                        }
                        if (j == 0) { s0 = be[0] * b0; s1 = be[0] * b1; s2 = be[0] * b2; }
                        else { s0 += be[j] * b0; s1 += be[j] * b1; s2 += be[j] * b2; }
                        row += j + 1 < ny ? spw : 0;
                    }
                    const int d0 = (int)sat_u8_f(s0) - (int)(refpx & 255u), d1 = (int)sat_u8_f(s1) - (int)((refpx >> 8) & 255u), d2 = (int)sat_u8_f(s2) - (int)((refpx >> 16) & 255u);
                    acc += (unsigned)(d0 * d0 + d1 * d1 + d2 * d2);
                };
                // (the projective kernel keeps the one generic instance: its eighteen registers of matrix push the specialised rows
                // into scratch)
                if (PERSP) area_rows(std::integral_constant<int, XBM>{});
                else if (xtaps_max <= 4) area_rows(std::integral_constant<int, 4>{});
                else if (xtaps_max == 5) area_rows(std::integral_constant<int, 5>{});
                else if (xtaps_max == 6) area_rows(std::integral_constant<int, 6>{});
                else area_rows(std::integral_constant<int, XBM>{});
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(&fcs[f].ssd[pd.s], red[0] + red[1] + red[2] + red[3]);
    }
}

// compute_similarity (image_utils.rs:22-27): 1 - (f32)sqrt(ssd) / sqrt(255*255*3 * p) (f32)
__device__ __forceinline__ float similarity_from_ssd(unsigned long long ssd, int sw, int sh) {
    double err = sqrt((double)ssd);
    float max_error = sqrtf((255.0f * 255.0f * 3.0f) * (float)(sw * sh));
    return 1.0f - (float)err / max_error;
}

// verdict_kernel: one thread per frame.      lib.rs:370-389
__global__ void verdict_kernel(VerifyParams vp, int nframes, const uint32_t* __restrict__ qofs,
                               const PageInfo* __restrict__ pages, FrameCands* __restrict__ fcs,
                               slideo_verdict* __restrict__ out) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nframes) return;
    FrameCands& fc = fcs[f];
    slideo_verdict v;
    v.page_idx = -1; v.similarity = 0.f; v.inliers = 0; v.n_keypoints = (int32_t)(qofs[f + 1] - qofs[f]);
    if (v.n_keypoints == 0) { fc.ncand = 0; fc.nsurv = 0; }
    // stable: first maximal similarity among survivors in rating order; verdict_rule 1 (opt-in departure, slideo_amd.h): the
    // first survivor in rating order whose similarity is accepted
    float best = 0.f; int bi = -1;
    for (int s = 0; s < fc.nsurv; ++s) {
        const PageInfo pg = pages[fc.page[fc.surv[s]]];
        float sim = similarity_from_ssd(fc.ssd[s], pg.sw, pg.sh);
        fc.sim[s] = sim;
        if (vp.verdict_rule == 1) { if (bi < 0 && sim > vp.min_similarity) { best = sim; bi = s; } }
        else if (bi < 0 || sim > best) { best = sim; bi = s; }
    }
    if (bi >= 0 && best > vp.min_similarity) {
        v.page_idx = fc.page[fc.surv[bi]]; v.similarity = best; v.inliers = fc.inliers[fc.surv[bi]];
    }
    out[f] = v;
}

}  // namespace slideo
