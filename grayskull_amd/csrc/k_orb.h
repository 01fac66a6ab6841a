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
#include <type_traits>
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

/* gs_get (ref :23-26): the pixel, 0 outside the image.  Frames below 2 GiB go through an unconditional buffer load (outside:
 * the out-of-range offset), so that a thread's pixels are requested together -- hipcc turns `in ? img[..] : 0` into a branch
 * around the load and waits for each one before the next is issued; larger frames keep that form. */
struct PxReader {
  const uint8_t *img;
  BufRsrc B;
  unsigned w, h;
  bool small;
  GS_DEV PxReader(const uint8_t *p, unsigned w_, unsigned h_) : img(p), B(make_buf(p, (size_t)w_ * h_)), w(w_), h(h_), small((size_t)w_ * h_ < 0x7fffffffull) {}
  template <bool SMALL> GS_DEV unsigned get(unsigned sx, unsigned sy, bool also = true) const {
    const bool in = also && sx < w && sy < h;
    if constexpr (SMALL) return buf_load1(B, in ? sy * w + sx : kOOB);
    else return in ? img[(size_t)sy * w + sx] : 0u;
  }
  /* f(std::true_type / false_type): the caller's whole group of reads in one of the two forms (block-uniform choice) */
  template <class F> GS_DEV void with(F &&f) const {
    if (small) f(std::true_type{});
    else f(std::false_type{});
  }
};

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
  /* The patch is read eight pixels per lane at a time with unconditional buffer loads (a pixel outside the disc or the
   * image: the out-of-range offset, which reads 0 = gs_get's value there).  With the load inside the tests every one of a
   * lane's 16 pixels (r = 15) was a memory latency of its own: 20.7 us per 32 x 500 keypoints (profiles/r05n_orb_kernels.log). */
  const PxReader px(img, w, h);
  int m01 = 0, m10 = 0;
  px.with([&](auto SM) {
    for (int t0 = (int)threadIdx.x; t0 < total; t0 += 64 * 8) { /* wave-uniform trip count */
      int I[8], dyv[8], dxv[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int t = t0 + 64 * u;
        dyv[u] = t / side - (int)r, dxv[u] = t % side - (int)r;
        const unsigned sx = x + (unsigned)dxv[u], sy = y + (unsigned)dyv[u]; /* gs_get: wrap => 0 */
        I[u] = (int)px.template get<decltype(SM)::value>(sx, sy, t < total && dxv[u] * dxv[u] + dyv[u] * dyv[u] <= rr);
      }
#pragma unroll
      for (int u = 0; u < 8; u++) m01 += dyv[u] * I[u], m10 += dxv[u] * I[u];
    }
  });
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

/* grid nkp, block 256 (thread = one of the 256 point pairs); desc: nkp x 8 u32.
 * blockIdx.y = frame of a batch (frames frame_bytes apart; kin / desc hold gridDim.x slots per frame, count_dev[frame] of them
 * filled); single images launch with gridDim.y = 1 and no count */
__global__ __launch_bounds__(256) void k_brief(const uint8_t *img, unsigned w, unsigned h,
                                               const KpIn *kin, uint32_t *desc, const unsigned *count_dev = nullptr,
                                               size_t frame_bytes = 0) {
#ifndef GS_EMU
#pragma clang fp contract(off)
#endif
  if (count_dev && blockIdx.x >= count_dev[blockIdx.y]) return; /* whole block */
  img += (size_t)blockIdx.y * frame_bytes;
  kin += (size_t)blockIdx.y * gridDim.x, desc += (size_t)blockIdx.y * gridDim.x * 8u;
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
  const PxReader px(img, w, h); /* both pixels requested before either is used */
  unsigned I1 = 0, I2 = 0;
  px.with([&](auto SM) { I1 = px.template get<decltype(SM)::value>(x1, y1), I2 = px.template get<decltype(SM)::value>(x2, y2); });
  const uint64_t m = ballot(I1 > I2);
  if (lane_id() == 0) {
    const unsigned wv = i >> 6;
    desc[(size_t)blockIdx.x * 8u + 2 * wv] = (uint32_t)m;
    desc[(size_t)blockIdx.x * 8u + 2 * wv + 1] = (uint32_t)(m >> 32);
  }
}

/* ------------------------------------------------------------------ device-resident ORB (GS_NO_STDLIB trig) */
/* The reference's own trig for builds without libm (ref :70-88, used by examples/wasm/grayskull.c:32):
 * two float32 polynomials made of + - x / and compares only, so the GPU reproduces them bit for bit
 * (fp contract off; hipcc's float division is correctly rounded by default).  With them the whole of
 * gs_orb_extract (ref :651-669) runs on the device with no host round trip.  (Host-callable too: the drop-in's
 * single-keypoint entry points of the GS_NO_STDLIB flavour return their float to the host anyway; the library's host
 * code is built -ffp-contract=off like the reference, so both sides round alike.) */
GS_HD float gs_atan2_poly(float y, float x) { /* ref :70-78 */
  if (x == 0.0f) return y > 0.0f ? 1.570796f : (y < 0.0f ? -1.570796f : 0.0f);
  float r, angle;
  const float abs_y = y >= 0.0f ? y : -y;
  if (x >= 0.0f) {
    r = (x - abs_y) / (x + abs_y);
    angle = 0.785398f - 0.785398f * r;
  } else {
    r = (x + abs_y) / (abs_y - x);
    angle = 3.0f * 0.785398f - 0.785398f * r;
  }
  return y < 0.0f ? -angle : angle;
}
GS_HD float gs_sin_poly(float x) { /* ref :80-88 */
  while (x > 3.141592f) x -= 6.283185f;
  while (x < -3.141592f) x += 6.283185f;
  int sign = 1;
  if (x < 0) x = -x, sign = -1;
  if (x > 1.570796f) x = 3.141592f - x;
  const float x2 = x * x, res = x * (1.0f - x2 * (0.16666667f - 0.0083333310f * x2));
  return (float)sign * res;
}

/* Selection step of gs_orb_extract (ref :657-667) for one frame per WAVE: the FAST candidates arrive in
 * scan order; the reference bubble-sorts them by response, descending and stable, then keeps the first
 * nkps that lie >= 15 px inside the image.  Filtering commutes with a stable sort, and a response is
 * a scoremap byte, so: count the in-border candidates per response (256 LDS counters), rank of a
 * candidate = (in-border candidates with a larger response) + (earlier in-border candidates with the
 * same response), keep rank < nkps.  The "earlier, same response" part is resolved 64 candidates at a
 * time: a lane finds the lanes that share its response with one ballot per response bit.
 * grid n frames, block 64.  cand: n x cap records (12 u32), out: n x nkps records. */
__global__ __launch_bounds__(64) void k_orb_select(const unsigned *cand, const unsigned *cand_count, unsigned cap,
                                                  unsigned w, unsigned h, unsigned nkps, unsigned *out,
                                                  unsigned *out_count) {
  __shared__ unsigned cnt[256], base[256];
  const unsigned lane = threadIdx.x, f = blockIdx.x, r = 15u;
  const unsigned n = cand_count[f] < cap ? cand_count[f] : cap;
  const unsigned *c = cand + (size_t)f * cap * 12u;
  unsigned *o = out + (size_t)f * nkps * 12u;
  for (unsigned i = lane; i < 256u; i += 64u) cnt[i] = 0;
  __syncthreads();
  /* candidate records are read eight trips of 64 at a time, from clamped indices (no load inside a bounds test: a trip is one
   * memory latency, and hipcc does not overlap the trips of a loop whose loads sit in branches) */
  constexpr unsigned U = 8;
  for (unsigned i0 = 0; i0 < n; i0 += 64u * U) { /* wave-uniform */
    unsigned x[U], y[U], resp[U];
#pragma unroll
    for (unsigned u = 0; u < U; u++) {
      const unsigned i = i0 + 64u * u + lane, ic = i < n ? i : n - 1u;
      x[u] = c[(size_t)ic * 12u], y[u] = c[(size_t)ic * 12u + 1], resp[u] = c[(size_t)ic * 12u + 2] & 255u;
    }
#pragma unroll
    for (unsigned u = 0; u < U; u++)
      if (i0 + 64u * u + lane < n && x[u] >= r && y[u] >= r && x[u] < w - r && y[u] < h - r) atomicAdd(&cnt[resp[u]], 1u);
  }
  __syncthreads();
  /* base[v] = in-border candidates with a response > v: suffix sums, 4 bins per lane (lane 0 = bins 252..255) */
  {
    unsigned own[4], tot = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) own[k] = cnt[255u - (lane * 4u + (unsigned)k)], tot += own[k];
    unsigned run = wave_incl_scan(tot) - tot; /* candidates in the lanes before this one = larger responses */
    const unsigned total = readlane_last(wave_incl_scan(tot));
#pragma unroll
    for (int k = 0; k < 4; k++) base[255u - (lane * 4u + (unsigned)k)] = run, run += own[k];
    if (lane == 0) out_count[f] = total < nkps ? total : nkps;
  }
  __syncthreads();
  for (unsigned j0 = 0; j0 < n; j0 += 64u * U) { /* wave-uniform */
    unsigned xs[U], ys[U], rs[U];
#pragma unroll
    for (unsigned u = 0; u < U; u++) {
      const unsigned i = j0 + 64u * u + lane, ic = i < n ? i : n - 1u;
      xs[u] = c[(size_t)ic * 12u], ys[u] = c[(size_t)ic * 12u + 1], rs[u] = c[(size_t)ic * 12u + 2];
    }
#pragma unroll
    for (unsigned u = 0; u < U; u++) { /* the trips in scan order */
      if (j0 + 64u * u >= n) break; /* wave-uniform */
      const unsigned x = xs[u], y = ys[u], resp = rs[u];
      const bool ok = j0 + 64u * u + lane < n && x >= r && y >= r && x < w - r && y < h - r;
      /* the in-border lanes with this lane's response: eight ballots, one per bit of the response (rounds 2-4 peeled off one
       * distinct response per iteration with two barriers each -- ~1000 iterations per 2000 candidates) */
      const unsigned rv = resp & 255u;
      uint64_t same = ballot(ok);
#pragma unroll
      for (unsigned b = 0; b < 8u; b++) {
        const bool bit = ((rv >> b) & 1u) != 0u;
        const uint64_t m = ballot(ok && bit);
        same &= bit ? m : ~m;
      }
      const unsigned start = base[ok ? rv : 0u];
      const unsigned before = (unsigned)__popcll(same & ((1ull << lane) - 1ull));
      const unsigned rank = ok ? start + before : 0xffffffffu;
      __syncthreads(); /* every lane has read base[] */
      if (ok && before == 0u) base[rv] = start + (unsigned)__popcll(same); /* the first lane of each response's group */
      __syncthreads();
      if (ok && rank < nkps) {
        unsigned *q = o + (size_t)rank * 12u;
        q[0] = x, q[1] = y, q[2] = resp;
#pragma unroll
        for (int k = 3; k < 12; k++) q[k] = 0;
      }
    }
  }
}

/* Orientation (ref :608-621) and rotated BRIEF (ref :623-637) of the selected keypoints with the
 * GS_NO_STDLIB trig, one 256-thread block per keypoint: the disc moments are integer sums (exact,
 * r = 15), thread 0 evaluates the two polynomials, then thread i tests point pair i.
 * grid (nkps, n frames); kps: n x nkps records (12 u32), count: n. */
__global__ __launch_bounds__(256) void k_orb_describe(const uint8_t *img, unsigned w, unsigned h, size_t frame_bytes,
                                                     unsigned *kps, const unsigned *count, unsigned nkps) {
#ifndef GS_EMU
#pragma clang fp contract(off)
#endif
  __shared__ int part[2][4];
  __shared__ float trig[2];
  const unsigned f = blockIdx.y, k = blockIdx.x, i = threadIdx.x;
  if (k >= count[f]) return; /* whole block */
  img += (size_t)f * frame_bytes;
  unsigned *kp = kps + ((size_t)f * nkps + k) * 12u;
  const unsigned x = kp[0], y = kp[1];
  const int r = 15, side = 2 * r + 1, total = side * side;
  const PxReader px(img, w, h);
  int m01 = 0, m10 = 0;
  px.with([&](auto SM) { /* a thread's four pixels of the 31 x 31 patch, requested together (see k_orient_moments) */
    int I[4], dyv[4], dxv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int t = (int)i + 256 * u;
      dyv[u] = t / side - r, dxv[u] = t % side - r;
      const unsigned sx = x + (unsigned)dxv[u], sy = y + (unsigned)dyv[u];
      I[u] = (int)px.template get<decltype(SM)::value>(sx, sy, t < total && dxv[u] * dxv[u] + dyv[u] * dyv[u] <= r * r);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) m01 += dyv[u] * I[u], m10 += dxv[u] * I[u];
  });
  m01 = wave_sum_i(m01), m10 = wave_sum_i(m10);
  if ((i & 63u) == 0) part[0][i >> 6] = m01, part[1][i >> 6] = m10;
  __syncthreads();
  if (i == 0) {
    const int a = part[0][0] + part[0][1] + part[0][2] + part[0][3], b = part[1][0] + part[1][1] + part[1][2] + part[1][3];
    const float angle = gs_atan2_poly((float)a, (float)b);                  /* ref :620 */
    kp[3] = __builtin_bit_cast(uint32_t, angle);
    trig[0] = gs_sin_poly(angle), trig[1] = gs_sin_poly((float)(angle + 1.57079f)); /* ref :626 */
  }
  __syncthreads();
  const float sin_a = trig[0], cos_a = trig[1];
  const int p0 = k_brief_pattern[4 * i], p1 = k_brief_pattern[4 * i + 1];
  const int p2 = k_brief_pattern[4 * i + 2], p3 = k_brief_pattern[4 * i + 3];
  const float m00 = (float)p0 * cos_a, m01f = (float)p1 * sin_a;
  const float m10f = (float)p0 * sin_a, m11 = (float)p1 * cos_a;
  const float n00 = (float)p2 * cos_a, n01 = (float)p3 * sin_a;
  const float n10 = (float)p2 * sin_a, n11 = (float)p3 * cos_a;
  const float dx1 = m00 - m01f, dy1 = m10f + m11, dx2 = n00 - n01, dy2 = n10 + n11;
  const unsigned x1 = (unsigned)((int)x + (int)dx1), y1 = (unsigned)((int)y + (int)dy1);
  const unsigned x2 = (unsigned)((int)x + (int)dx2), y2 = (unsigned)((int)y + (int)dy2);
  unsigned I1 = 0, I2 = 0;
  px.with([&](auto SM) { I1 = px.template get<decltype(SM)::value>(x1, y1), I2 = px.template get<decltype(SM)::value>(x2, y2); });
  const uint64_t m = ballot(I1 > I2);
  if (lane_id() == 0) {
    const unsigned wv = i >> 6;
    kp[4 + 2 * wv] = (uint32_t)m, kp[4 + 2 * wv + 1] = (uint32_t)(m >> 32);
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
  /* four train descriptors per lane and trip, all thirty-two dwords requested before the first is used (index clamped, the
   * surplus ones ignored): a trip is one memory latency, and 2500 train descriptors were 40 of them in a row (round 5:
   * 2500 x 2500 62 -> 45 us, 500 x 500 27 -> 25; eight per trip: no further gain, profiles/r05n_match_*.log) */
  for (unsigned j0 = lane; j0 < n2; j0 += 256u) { /* a lane's indices in ascending order: the first minimum wins (ref :690) */
    uint32_t d2[4][8];
#pragma unroll
    for (unsigned u = 0; u < 4u; u++) {
      const unsigned j = j0 + 64u * u, jc = j < n2 ? j : n2 - 1u;
#pragma unroll
      for (int q = 0; q < 8; q++) d2[u][q] = k2[(size_t)jc * 12u + 4 + q];
    }
#pragma unroll
    for (unsigned u = 0; u < 4u; u++) {
      const unsigned j = j0 + 64u * u;
      unsigned bits = 0;
#pragma unroll
      for (int q = 0; q < 8; q++) bits += (unsigned)__popc(d1[q] ^ d2[u][q]);
      const float d = (float)bits;
      if (j < n2) {
        if (d < best) second = best, best = d, arg = j;
        else if (d < second) second = d;
      }
    }
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
