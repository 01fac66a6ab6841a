/*
 * k_compact.h -- deterministic ORDERED stream compaction with a cap.
 *
 * gs_fast, gs_lbp_detect and gs_match_orb all emit "the first N hits in scan order"
 * (grayskull.h:530, :824-831, :695-696).  atomicAdd-append would scramble the order, so:
 *
 *   1. the producing kernel publishes one 64-bit ballot word per wave (item i = bit i%64 of
 *      word i/64) and adds popcount(word) to the counter of its chunk (kChunkWords words);
 *   2. k_chunk_scan: exclusive prefix sum of the chunk counters (one block per frame);
 *   3. k_emit<F>: one thread per non-empty chunk walks its words in order and writes hit
 *      number r (< cap) through the functor F(frame, item, r).
 *
 * Hits are rare in all three users, so steps 2-3 are noise next to step 1.
 */
#ifndef GS_K_COMPACT_H
#define GS_K_COMPACT_H
#include "prims.h"

namespace gs {

constexpr unsigned kChunkWords = 32;                  /* 2048 items per chunk */
constexpr unsigned kChunkItems = kChunkWords * 64u;

/* called by ALL lanes of a wave, `word` wave-uniform */
GS_DEV void publish_flags(bool flag, unsigned long long *mask, unsigned *chunk_count,
                          size_t word) {
  const uint64_t m = ballot(flag);
  if (lane_id() == 0) {
    mask[word] = m;
    if (m) atomicAdd(&chunk_count[word / kChunkWords], (unsigned)__popcll(m));
  }
}

/* grid n frames, block 1024.  prefix[c] = sum of count[<c]; total[f] = min(sum, cap) */
__global__ __launch_bounds__(1024) void k_chunk_scan(const unsigned *count, unsigned nchunks,
                                                     unsigned *prefix, unsigned *total,
                                                     unsigned cap) {
  __shared__ unsigned wsum[16];
  __shared__ unsigned carry_s;
  const unsigned tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const unsigned *cf = count + (size_t)blockIdx.x * nchunks;
  unsigned *pf = prefix + (size_t)blockIdx.x * nchunks;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (unsigned base = 0; base < nchunks; base += 1024) {
    const unsigned i = base + tid;
    const unsigned v = i < nchunks ? cf[i] : 0u;
    const unsigned inc = wave_incl_scan(v);
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    unsigned woff = 0;
    for (unsigned k = 0; k < wv; k++) woff += wsum[k];
    const unsigned carry = carry_s;
    if (i < nchunks) pf[i] = carry + woff + inc - v;
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + inc;
    __syncthreads();
  }
  if (tid == 0) total[blockIdx.x] = carry_s < cap ? carry_s : cap;
}

/* grid (ceil(nchunks/256), n frames), block 256 */
template <class F>
__global__ __launch_bounds__(256) void k_emit(const unsigned long long *mask,
                                              const unsigned *count, const unsigned *prefix,
                                              unsigned nchunks, unsigned cap, F emit) {
  const unsigned c = blockIdx.x * 256u + threadIdx.x;
  if (c >= nchunks) return;
  const size_t fc = (size_t)blockIdx.y * nchunks + c;
  if (count[fc] == 0) return;
  unsigned r = prefix[fc];
  if (r >= cap) return;
  const unsigned long long *mw = mask + fc * kChunkWords;
  for (unsigned k = 0; k < kChunkWords && r < cap; k++) {
    unsigned long long m = mw[k];
    while (m && r < cap) {
      const unsigned b = (unsigned)__builtin_ctzll(m);
      m &= m - 1;
      emit(blockIdx.y, (size_t)c * kChunkItems + k * 64u + b, r);
      r++;
    }
  }
}

}  // namespace gs
#endif
