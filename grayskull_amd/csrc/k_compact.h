/*
 * k_compact.h -- deterministic ORDERED stream compaction with a cap.
 *
 * gs_fast, gs_lbp_detect and gs_match_orb all emit "the first N hits in scan order"
 * (grayskull.h:530, :824-831, :695-696).  atomicAdd-append would scramble the order, so:
 *
 *   1. the producing kernel publishes one 64-bit ballot word per wave (item i = bit i%64 of
 *      word i/64) and adds popcount(word) to the counter of its chunk (kChunkWords words);
 *   2. k_chunk_scan: exclusive prefix sum of the chunk counters (one block per frame) -- or, for frames of at
 *      most kEmitSelfScan chunks, no pass at all: each k_emit wave adds up the counters before its own chunk;
 *   3. k_emit<F>: one wave per non-empty chunk; a lane owns 32 consecutive items, ranks its hits behind those of the
 *      lanes before it and writes hit number r (< cap) through the functor F(frame, item, r).
 */
#ifndef GS_K_COMPACT_H
#define GS_K_COMPACT_H
#include "prims.h"

namespace gs {

constexpr unsigned kChunkWords = 32;                  /* 2048 items per chunk */
constexpr unsigned kChunkItems = kChunkWords * 64u;
constexpr unsigned kEmitSelfScan = 1024;             /* chunks per frame up to which k_emit does its own prefix sums */

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

/* One WAVE per chunk (grid (ceil(nchunks/4), n frames), block 256): lane k < 32 fetches word k of
 * the chunk once, then every lane takes one 32-item half of a word (see the loop at the end).  (One thread per
 * chunk walked up to 2048 items serially: 100 us for a chunk full of FAST corners.) */
/* QUAD: the producer's lanes own 4 consecutive items each (k_fast_nms), so a group of 256 items is
 * 4 words in SLOT-major form -- word s of the group, bit l = item 256 g + 4 l + s -- i.e. the four
 * ballots as they come.  Scan order inside a group is then lane-major: lane l's hits rank after
 * all hits of lanes < l in the four words, slot by slot. */
template <class F, bool QUAD = false>
__global__ __launch_bounds__(256) void k_emit(const unsigned long long *mask,
                                              const unsigned *count, const unsigned *prefix,
                                              unsigned nchunks, unsigned cap, F emit, unsigned *total = nullptr) {
  const unsigned lane = threadIdx.x & 63u;
  const unsigned c = uniform(blockIdx.x * 4u + (threadIdx.x >> 6));
  if (c >= nchunks) return; /* whole wave */
  const size_t fc = (size_t)blockIdx.y * nchunks + c;
  unsigned r;
  unsigned long long mine_early = 0;
  if (prefix) {
    if (uniform(count[fc]) == 0) return;
    r = uniform(prefix[fc]);
  } else {
    /* no scan pass (few chunks per frame): the wave sums the counters of the chunks before its own; the wave of
     * chunk 0 also leaves the frame's total */
    const unsigned *cf = count + (size_t)blockIdx.y * nchunks;
    if (c == 0) {
      unsigned t = 0;
      for (unsigned i = lane; i < nchunks; i += 64u) t += cf[i];
      t = wave_sum(t);
      if (lane == 0) total[blockIdx.y] = t < cap ? t : cap;
    }
    /* the pass is a chain of memory round trips for a handful of hits: the counters before the chunk, its own
     * counter and its mask words are requested together (the words of an empty chunk are a wasted 256 bytes) */
    mine_early = lane < kChunkWords ? mask[fc * kChunkWords + lane] : 0ull;
    unsigned before = 0;
    for (unsigned i = lane; i < c; i += 64u) before += cf[i];
    if (uniform(cf[c]) == 0) return;
    r = wave_sum(before);
  }
  if (r >= cap) return;
  const unsigned long long mine = prefix ? (lane < kChunkWords ? mask[fc * kChunkWords + lane] : 0ull) : mine_early;
  const uint32_t mlo = (uint32_t)mine, mhi = (uint32_t)(mine >> 32);
  if constexpr (QUAD) {
    for (unsigned g = 0; g < kChunkWords / 4 && r < cap; g++) { /* wave-uniform */
      uint64_t m[4];
#pragma unroll
      for (unsigned s = 0; s < 4; s++) m[s] = ((uint64_t)readlane_at(mhi, 4 * g + s) << 32) | readlane_at(mlo, 4 * g + s);
      if (!(m[0] | m[1] | m[2] | m[3])) continue;
      const uint64_t lower = (1ull << lane) - 1ull;
      unsigned rank = r + (unsigned)(__popcll(m[0] & lower) + __popcll(m[1] & lower) + __popcll(m[2] & lower) + __popcll(m[3] & lower));
#pragma unroll
      for (unsigned s = 0; s < 4; s++)
        if ((m[s] >> lane) & 1ull) {
          if (rank < cap) emit(blockIdx.y, (size_t)c * kChunkItems + g * 256u + lane * 4u + s, rank);
          rank++;
        }
      r += (unsigned)(__popcll(m[0]) + __popcll(m[1]) + __popcll(m[2]) + __popcll(m[3]));
    }
    return;
  }
  /* Lane l takes half l & 1 of word l >> 1: its hits rank after all hits of the lower halves (one wave scan of the 64
   * popcounts) and are written one per trip, every lane at once -- max popcount(half) trips for the chunk.  (Visiting
   * the words one after the other, 64 lanes per word, took one memory round trip per NON-EMPTY word -- the functor
   * loads before it stores -- i.e. up to 32 in a row for a chunk full of corners: the launch's tail.) */
  const uint32_t hlo = shfl(mlo, (int)(lane >> 1)), hhi = shfl(mhi, (int)(lane >> 1));
  uint32_t half = (lane & 1u) ? hhi : hlo;
  const unsigned pc = (unsigned)__popc(half);
  unsigned rank = r + wave_incl_scan(pc) - pc;
  const size_t item0 = (size_t)c * kChunkItems + lane * 32u;
  while (ballot(half != 0u && rank < cap) != 0) { /* wave-uniform */
    if (half != 0u) {
      const unsigned b = (unsigned)__ffs((int)half) - 1u;
      half &= half - 1u;
      if (rank < cap) emit(blockIdx.y, item0 + b, rank);
      rank++;
    }
  }
}

}  // namespace gs
#endif
