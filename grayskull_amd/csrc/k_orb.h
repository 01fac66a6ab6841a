/*
 * k_orb.h -- the device halves of gs_orb_extract / gs_match_orb (grayskull.h:608-699).
 *
 *  k_orient_moments  intensity-centroid moments over the disc dx^2+dy^2 <= r^2 (ref :608-621).
 *                    Accumulated as integers: every partial sum is < 2^24 in magnitude, so the
 *                    reference's float accumulation is exact and int -> float gives the same
 *                    bits.  atan2f itself runs on the HOST libm, because the
 *                    reference's angle is whatever glibc returns (ref :100) and the BRIEF bits
 *                    depend on it through (int) truncation.
 *  k_brief           rotated BRIEF-256 given host-computed sin/cos (ref :623-637): float32, no
 *                    FMA contraction, truncation toward zero, out-of-image taps read 0.
 *  k_match           brute-force Hamming NN + ratio test (ref :680-699) with XOR + popcount;
 *                    one wave per query, lanes stride over the train descriptors, butterfly
 *                    merge of (best, second, first-argmin).
 */
#ifndef GS_K_ORB_H
#define GS_K_ORB_H
#include "k_compact.h"

namespace gs {

#ifdef GS_EMU
#define GS_CONST_TABLE static const
#else
#define GS_CONST_TABLE __constant__ const
#endif
GS_CONST_TABLE int8_t k_brief_pattern[1024] = {
#include "brief_pattern.inc"
};

struct KpIn { unsigned x, y; float sin_a, cos_a; }; /* host -> device per keypoint */

/* grid nkp (or an upper bound, with the true count in *count_dev), block 64.
 * pts: (x,y) pairs; out: (m01, m10) int pairs */
/* blockIdx.y = frame of a batch (frames frame_bytes apart; pts / out / count_dev hold gridDim.x
 * slots per frame); single images launch with gridDim.y = 1 */
__global__ __launch_bounds__(64) void k_orient_moments(const uint8_t *img, unsigned w, unsigned h,
                                                       const unsigned *pts, unsigned pt_stride,
                                                       unsigned r, int *out, const unsigned *count_dev,
                                                       size_t frame_bytes = 0) {
  if (count_dev && blockIdx.x >= count_dev[blockIdx.y]) return;
  img += (size_t)blockIdx.y * frame_bytes;
  pts += (size_t)blockIdx.y * gridDim.x * pt_stride;
  out += (size_t)blockIdx.y * gridDim.x * 2;
  const unsigned x = pts[(size_t)blockIdx.x * pt_stride], y = pts[(size_t)blockIdx.x * pt_stride + 1];
  const int side = 2 * (int)r + 1, total = side * side, rr = (int)(r * r);
  int m01 = 0, m10 = 0;
  for (int t = (int)threadIdx.x; t < total; t += 64) {
    const int dy = t / side - (int)r, dx = t % side - (int)r;
    if (dx * dx + dy * dy <= rr) {
      const unsigned sx = x + (unsigned)dx, sy = y + (unsigned)dy; /* gs_get: wrap => 0 */
      const int I = (sx < w && sy < h) ? img[(size_t)sy * w + sx] : 0;
      m01 += dy * I, m10 += dx * I;
    }
  }
  m01 = wave_sum_i(m01), m10 = wave_sum_i(m10);
  if (threadIdx.x == 0) out[2 * blockIdx.x] = m01, out[2 * blockIdx.x + 1] = m10;
}

/* The integer sums above equal the reference's float32 accumulation (ref :610-618) only while every
 * partial sum stays below 2^24: 255 * sum |dy| over the disc ~ 255 * (4/3) r^3, i.e. r <= kOrientExactR
 * (gs_orb_extract always uses r = 15).  For larger radii the drop-in gs_compute_orientation runs the
 * reference's loop as it stands: ONE thread, dy outer / dx inner, float32 adds in that order. */
constexpr unsigned kOrientExactR = 36;
__global__ void k_orient_moments_seq(const uint8_t *img, unsigned w, unsigned h, unsigned x, unsigned y,
                                     unsigned r, float *out) {
#ifndef GS_EMU
#pragma clang fp contract(off)
#endif
  if (threadIdx.x || blockIdx.x) return;
  float m01 = 0, m10 = 0;
  const int R = (int)r, rr = (int)(r * r);
  for (int dy = -R; dy <= R; dy++)
    for (int dx = -R; dx <= R; dx++)
      if (dx * dx + dy * dy <= rr) {
        const unsigned sx = x + (unsigned)dx, sy = y + (unsigned)dy;
        const int I = (sx < w && sy < h) ? img[(size_t)sy * w + sx] : 0;
        m01 += (float)(dy * I);
        m10 += (float)(dx * I);
      }
  out[0] = m01, out[1] = m10;
}

/* grid nkp, block 256 (thread = one of the 256 point pairs); desc: nkp x 8 u32 */
__global__ __launch_bounds__(256) void k_brief(const uint8_t *img, unsigned w, unsigned h,
                                               const KpIn *kin, uint32_t *desc) {
#ifndef GS_EMU
#pragma clang fp contract(off)
#endif
  const KpIn kp = kin[blockIdx.x];
  const unsigned i = threadIdx.x;
  const int p0 = k_brief_pattern[4 * i], p1 = k_brief_pattern[4 * i + 1];
  const int p2 = k_brief_pattern[4 * i + 2], p3 = k_brief_pattern[4 * i + 3];
  const float m00 = (float)p0 * kp.cos_a, m01 = (float)p1 * kp.sin_a;
  const float m10 = (float)p0 * kp.sin_a, m11 = (float)p1 * kp.cos_a;
  const float n00 = (float)p2 * kp.cos_a, n01 = (float)p3 * kp.sin_a;
  const float n10 = (float)p2 * kp.sin_a, n11 = (float)p3 * kp.cos_a;
  const float dx1 = m00 - m01, dy1 = m10 + m11, dx2 = n00 - n01, dy2 = n10 + n11;
  const unsigned x1 = (unsigned)((int)kp.x + (int)dx1), y1 = (unsigned)((int)kp.y + (int)dy1);
  const unsigned x2 = (unsigned)((int)kp.x + (int)dx2), y2 = (unsigned)((int)kp.y + (int)dy2);
  const unsigned I1 = (x1 < w && y1 < h) ? img[(size_t)y1 * w + x1] : 0u;
  const unsigned I2 = (x2 < w && y2 < h) ? img[(size_t)y2 * w + x2] : 0u;
  const uint64_t m = ballot(I1 > I2);
  if (lane_id() == 0) {
    const unsigned wv = i >> 6;
    desc[(size_t)blockIdx.x * 8u + 2 * wv] = (uint32_t)m;
    desc[(size_t)blockIdx.x * 8u + 2 * wv + 1] = (uint32_t)(m >> 32);
  }
}

/* keypoint record = 12 dwords, descriptor at dword 4 (grayskull.h:42-47).
 * One WAVE per query: the 64 lanes stride over the train descriptors, each keeping the
 * reference's (best, second, first-argmin) state; states merge with
 *   best = min(b1,b2), second = min(max(b1,b2), s1, s2), arg = arg of the smaller best (lower
 *   index on ties)
 * which is the sequential loop's result for any visiting order (ref :686-694: two smallest of
 * the multiset {init, init, d_0, d_1, ...} with strict '<', first index attaining the minimum).
 * grid ceil(n1/4), block 256.  mask / chunk_count pre-zeroed; bit i of the mask = query i accepted. */
__global__ __launch_bounds__(256) void k_match(const uint32_t *k1, unsigned n1, const uint32_t *k2,
                                               unsigned n2, float max_distance, unsigned *best_idx,
                                               unsigned *best_dist, unsigned long long *mask,
                                               unsigned *chunk_count) {
  const unsigned lane = threadIdx.x & 63u;
  const unsigned i = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (i >= n1) return; /* whole wave */
  uint32_t d1[8];
#pragma unroll
  for (int q = 0; q < 8; q++) d1[q] = k1[(size_t)i * 12u + 4 + q];
  float best = max_distance + 1, second = max_distance + 1;
  unsigned arg = 0;
  for (unsigned j = lane; j < n2; j += 64u) {
    const uint32_t *d2 = k2 + (size_t)j * 12u + 4;
    unsigned bits = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) bits += (unsigned)__popc(d1[q] ^ d2[q]);
    const float d = (float)bits;
    if (d < best) second = best, best = d, arg = j;
    else if (d < second) second = d;
  }
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) { /* butterfly merge of the 64 lane states */
    const int src = (int)(lane ^ (unsigned)sft);
    const float ob = __builtin_bit_cast(float, shfl(__builtin_bit_cast(uint32_t, best), src));
    const float os = __builtin_bit_cast(float, shfl(__builtin_bit_cast(uint32_t, second), src));
    const unsigned oa = shfl(arg, src);
    const float hi = ob > best ? ob : best; /* the larger of the two bests */
    float s2 = os < second ? os : second;
    s2 = hi < s2 ? hi : s2;
    if (ob < best || (ob == best && oa < arg)) arg = oa;
    best = ob < best ? ob : best;
    second = s2;
  }
  if (lane == 0) {
    best_idx[i] = arg;
    best_dist[i] = (unsigned)best;
    if (best <= max_distance && best < 0.8f * second) {
      atomicOr(&mask[i >> 6], 1ull << (i & 63u));
      atomicAdd(&chunk_count[(i >> 6) / kChunkWords], 1u);
    }
  }
}

/* compaction functor: query i -> gs_match {i, best_idx, distance} (ref :696) */
struct MatchEmit {
  const unsigned *best_idx, *best_dist;
  unsigned *matches; /* max_matches x 3 u32 */
  GS_DEV void operator()(unsigned, size_t item, unsigned r) const {
    unsigned *o = matches + (size_t)r * 3u;
    o[0] = (unsigned)item, o[1] = best_idx[item], o[2] = best_dist[item];
  }
};

}  // namespace gs
#endif
