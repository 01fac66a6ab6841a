/*
 * k_pointwise.h -- gs_threshold, gs_histogram, gs_otsu_threshold, checksums, synthetic frames.
 * Reference semantics: grayskull.h:199-228.  All HBM-bound: threshold 2 B/px (in-place R+W),
 * histogram 1 B/px.
 */
#ifndef GS_K_POINTWISE_H
#define GS_K_POINTWISE_H
#include "k_stencil.h" /* U4 */

namespace gs {

/* The 16-byte chunks of [base, base+n) are addressed relative to base rounded down to 16 B, so
 * the body is aligned dwordx4 traffic for ANY base; only the first/last chunk go bytewise. */
struct Chunking {
  uintptr_t a0; /* aligned-down base */
  size_t lo, hi; /* valid byte range relative to a0 */
  size_t nchunks;
};
__host__ __device__ inline Chunking make_chunking(const void *base, size_t n) {
  Chunking c;
  c.a0 = (uintptr_t)base & ~(uintptr_t)15;
  c.lo = (uintptr_t)base - c.a0;
  c.hi = c.lo + n;
  c.nchunks = (c.hi + 15) / 16;
  return c;
}

/* per-byte x > t ? 0xff : 0 on four bytes at once (SWAR, no cross-byte carries) */
GS_DEV uint32_t swar_gt_u8(uint32_t x, uint32_t trep) {
  const uint32_t Hb = 0x80808080u;
  uint32_t d = (trep | Hb) - (x & ~Hb);                  /* bit7 = (t&0x7f) >= (x&0x7f) */
  uint32_t ge = ((trep & ~x) | (~(trep ^ x) & d)) & Hb;  /* bit7 = t >= x */
  uint32_t gt = ~ge & Hb;                                /* bit7 = x > t  */
  return (gt >> 7) * 0xffu;
}

/* ref :225-228, in place.  grid (blocks, n frames); thr_dev (per-frame) overrides thr_const */
__global__ __launch_bounds__(256) void k_threshold(uint8_t *img, size_t frame_bytes,
                                                   const uint8_t *thr_dev, unsigned thr_const) {
  uint8_t *base = img + (size_t)blockIdx.y * frame_bytes;
  const unsigned t = thr_dev ? thr_dev[blockIdx.y] : thr_const;
  const uint32_t trep = t * 0x01010101u;
  const Chunking c = make_chunking(base, frame_bytes);
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < c.nchunks;
       i += (size_t)gridDim.x * 256u) {
    const size_t b0 = i * 16, b1 = b0 + 16;
    uint8_t *p = (uint8_t *)(c.a0 + b0);
    if (b0 >= c.lo && b1 <= c.hi) {
      /* in-place stream with no reuse: non-temporal load and store (6.6 TB/s vs 5.9 TB/s) */
#ifndef GS_EMU
      typedef unsigned int v4u __attribute__((ext_vector_type(4)));
      const v4u q = __builtin_nontemporal_load((const v4u *)p);
      U4 v{q.x, q.y, q.z, q.w};
#else
      U4 v = *(U4 *)p;
#endif
      v.x = swar_gt_u8(v.x, trep), v.y = swar_gt_u8(v.y, trep);
      v.z = swar_gt_u8(v.z, trep), v.w = swar_gt_u8(v.w, trep);
#ifndef GS_EMU
      __builtin_nontemporal_store(v4u{v.x, v.y, v.z, v.w}, (v4u *)p);
#else
      *(U4 *)p = v;
#endif
    } else {
      const size_t s = b0 < c.lo ? c.lo : b0, e = b1 > c.hi ? c.hi : b1;
      for (size_t k = s; k < e; k++) {
        uint8_t *q = (uint8_t *)(c.a0 + k);
        *q = *q > t ? 255 : 0;
      }
    }
  }
}

/* ref :199-203.  LDS-privatised histogram: 32 copies of the 256 bins, copy = lane & 31, so a
 * wave's 64 ds_add_u32 never collide on a bank whatever the pixel values are (flat image
 * regions would otherwise serialise 64-way).  grid (bpf, n frames); each block writes its
 * 256 partial counts to partial[(frame*bpf + block)*256 ..]; k_hist_reduce sums them
 * (no global atomics, deterministic).
 *   The LDS atomic unit takes a ds_add_u32 every ~4.1 cycles per CU whatever the lanes' bins are
 * (9.4 T pixel/s chip-wide, profiles/r02i_ubench_new_ops.log), above what HBM delivers at 1 B/px; the
 * first form of this kernel (load 16 B, wait, 16 atomics) sat at 4.2 Tpx/s on the load latency, with only
 * 20 waves per CU next to the 32 KB table.  So the loads are software-pipelined: every lane keeps DEPTH
 * 16-byte loads in flight ahead of the one whose bytes it is counting. */
#ifndef GS_HIST_DEPTH
#define GS_HIST_DEPTH 3
#endif
constexpr unsigned kHistDepth = GS_HIST_DEPTH;
/* NT: every byte is read once, and when the launch covers more than the Infinity Cache keeps, its loads stream (round 6,
 * same-box A/B of the load policy: 512 x 4K 0.67 -> 0.70 of 8 TB/s, 64 x 4096^2 0.685 -> 0.77; a handful of frames that
 * the benchmark loop finds in that cache again: 8 x 1080p 14.2 -> 15.8 us -- the launcher decides;
 * profiles/r06l_hist_nt_ab.log.  The strip kernels, which find their halo rows in the caches, lose 15-25 % to the same
 * policy: profiles/r06k_bench_nt_loads_ab.log.) */
constexpr size_t kHistMaxFrame = (size_t)1 << 30; /* bytes per k_hist_partial frame (32-bit buffer offsets) */
GS_DEV void hist_count16(unsigned *lh, unsigned copy, const U4 &v, unsigned inc) {
  const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 16; k++) atomicAdd(&lh[((d[k >> 2] >> (8 * (k & 3))) & 0xffu) * 32u + copy], inc);
}
/* BT threads share the block's 32 KB of counters: more waves per CU next to the same LDS footprint */
template <unsigned BT, bool NT = false>
__global__ __launch_bounds__(BT) void k_hist_partial(const uint8_t *img, size_t frame_bytes,
                                                     unsigned *partial) {
  __shared__ unsigned lh[256 * 32];
  const unsigned tid = threadIdx.x, copy = tid & 31u;
  for (unsigned i = tid; i < 256 * 32 / 4; i += BT) ((U4 *)lh)[i] = U4{0, 0, 0, 0};
  __syncthreads();
  const uint8_t *base = img + (size_t)blockIdx.y * frame_bytes;
  const Chunking c = make_chunking(base, frame_bytes);
  /* whole 16-byte chunks [c0, c1); the (at most two) partial ones at the ends of the frame go byte by byte */
  const size_t c0 = c.lo ? 1 : 0, c1 = c.hi / 16;
  const size_t stride = (size_t)gridDim.x * BT, first = c0 + (size_t)blockIdx.x * BT;
  if (first < c1) { /* block-uniform */
    /* The loop is branch-free and the loads are buffer loads (vmcnt only; a flat load would also tick the
     * LDS counter) so that the compiler's s_waitcnt leaves DEPTH - 1 loads in flight instead of draining
     * them: trips are padded to a multiple of DEPTH, loads past the last whole chunk return 0 from the
     * buffer bounds check, and the lanes (or padded trips) without a chunk of their own add 0.
     * kHistMaxFrame keeps every offset of the padded trips below 2^31. */
    const uintptr_t b0 = c.a0 + c0 * 16; /* block-uniform, but 64-bit selects end up in VGPRs: say so (else: waterfall loops) */
    const uint32_t end = uniform((uint32_t)(c1 - c0) * 16u);
    const BufRsrc src = make_buf((const void *)(((uintptr_t)uniform((uint32_t)(b0 >> 32)) << 32) | uniform((uint32_t)b0)), end);
    const unsigned iters = (unsigned)((c1 - first + stride - 1) / stride);
    const unsigned padded = (iters + kHistDepth - 1) / kHistDepth * kHistDepth;
    const uint32_t mine = ((uint32_t)blockIdx.x * BT + tid) * 16u, step = (uint32_t)stride * 16u;
    U4 q[kHistDepth];
#pragma unroll
    for (unsigned k = 0; k < kHistDepth; k++) {
      q[k] = buf_load16_pol<NT>(src, mine + k * step);
      sched_fence(); /* issue order = consumption order, or the first trip (hence every trip) waits for all of them */
    }
    uint32_t off = mine;
    for (unsigned it = 0; it < padded; it += kHistDepth) {
#pragma unroll
      for (unsigned k = 0; k < kHistDepth; k++) { /* static register names: the queue is a rotation of q[] */
        hist_count16(lh, copy, q[k], off < end ? 1u : 0u);
        q[k] = buf_load16_pol<NT>(src, off + kHistDepth * step); /* straight into the slot just consumed: no copies to wait for */
        off += step;
        sched_fence(); /* keep the steps apart: the scheduler would otherwise hoist all 48 address computations above
                          one wait for every load */
      }
    }
  }
  if (blockIdx.x == 0 && tid < 16) { /* ragged ends: head [lo, min(16, hi)) when lo > 0, tail [16 c1, hi) when c1 >= c0 */
    const size_t head_end = c.hi < 16 ? c.hi : 16;
    if (c.lo && c.lo + tid < head_end) atomicAdd(&lh[(unsigned)*(const uint8_t *)(c.a0 + c.lo + tid) * 32u + copy], 1u);
    if (c1 >= c0 && c1 * 16 + tid < c.hi) atomicAdd(&lh[(unsigned)*(const uint8_t *)(c.a0 + c1 * 16 + tid) * 32u + copy], 1u);
  }
  __syncthreads();
  if (tid < 256u) {
    unsigned s = 0;
#pragma unroll 8
    for (unsigned k = 0; k < 32; k++) s += lh[tid * 32u + ((k + tid) & 31u)]; /* rotated: conflict-free */
    partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256u + tid] = s;
  }
}

/* grid n frames, block 256: 8 independent partial sums per thread keep 8 loads in flight (a dependent chain over
 * a few hundred partial histograms would be ~1 us per step) */
__global__ __launch_bounds__(256) void k_hist_reduce(const unsigned *partial, unsigned bpf,
                                                     unsigned *hist, unsigned extra0) {
  const unsigned *p = partial + (size_t)blockIdx.x * bpf * 256u + threadIdx.x;
  unsigned acc[8] = {threadIdx.x == 0 ? extra0 : 0u, 0, 0, 0, 0, 0, 0, 0}; /* extra0: pixels known to be 0 that nobody counted */
  unsigned b = 0;
  for (; b + 8 <= bpf; b += 8) {
#pragma unroll
    for (unsigned k = 0; k < 8; k++) acc[k] += p[(size_t)(b + k) * 256u];
  }
  for (; b < bpf; b++) acc[0] += p[(size_t)b * 256u];
  hist[(size_t)blockIdx.x * 256u + threadIdx.x] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
}

/* ref :205-223.  The reference's scan is a chain of float32 adds (sum / sumB) followed, per t,
 * by an expression of (wb, sumB_t, sum) only.  So: every thread forms its product
 * (float)t*hist[t] (same rounding as the reference's), ONE thread runs the sequential add chain
 * once (prefix p[t] == the reference's sumB after step t, and p[255] == its `sum`: same values,
 * same order), then all 256 between-class variances are evaluated in parallel and reduced with
 * "greater wins, first index on ties" == the reference's strict `>` update.  No FMA contraction;
 * IEEE add/mul/div are correctly rounded on gfx950 as on x86-64.  grid n frames, block 256. */
/* partial != nullptr: the histogram is first folded from `bpf` per-block partial histograms (what k_hist_reduce
 * does, saved as a launch: the pipeline's threshold pass waits for this kernel) and written to `hist` as well;
 * extra0 = pixels known to be 0 that no block counted. */
__global__ __launch_bounds__(256) void k_otsu(unsigned *hist, unsigned npix, uint8_t *thr,
                                              const unsigned *partial = nullptr, unsigned bpf = 0, unsigned extra0 = 0) {
#ifndef GS_EMU
#pragma clang fp contract(off)
#endif
  __shared__ float prod[256];
  __shared__ float pre[256];
  __shared__ unsigned wsum[4];
  __shared__ float bv[4];
  __shared__ unsigned bt[4];
  const unsigned t = threadIdx.x, lane = t & 63u, wv = t >> 6;
  unsigned hv;
  if (partial) {
    const unsigned *p = partial + (size_t)blockIdx.x * bpf * 256u + t;
    unsigned acc[8] = {t == 0 ? extra0 : 0u, 0, 0, 0, 0, 0, 0, 0};
    unsigned b = 0;
    for (; b + 8 <= bpf; b += 8) {
#pragma unroll
      for (unsigned k = 0; k < 8; k++) acc[k] += p[(size_t)(b + k) * 256u];
    }
    for (; b < bpf; b++) acc[0] += p[(size_t)b * 256u];
    hv = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    hist[(size_t)blockIdx.x * 256u + t] = hv;
  } else {
    hv = hist[(size_t)blockIdx.x * 256u + t];
  }
  prod[t] = (float)t * (float)hv;
  const unsigned inc = wave_incl_scan(hv);
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  if (t == 0) { /* the reference's add chain, in its order; 8 LDS reads are in flight ahead of the 8 dependent adds */
    float s = 0;
    for (int i = 0; i < 256; i += 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = prod[i + k];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        s += v[k];
        pre[i + k] = s;
      }
    }
  }
  unsigned wb = inc;
  for (unsigned k = 0; k < wv; k++) wb += wsum[k];
  __syncthreads();
  const float sum = pre[255], sumB = pre[t];
  const unsigned wf = npix - wb;
  float var = -2.0f; /* below the reference's initial varMax = -1: never selected */
  if (wb != 0 && wf != 0) {
    const float mB = sumB / (float)wb;
    const float mF = (sum - sumB) / (float)wf;
    var = (float)wb * (float)wf * (mB - mF) * (mB - mF);
  }
  /* wf == 0 can only hold from some t on (break, ref :215); wb == 0 only up to some t (continue) */
  float v = var;
  unsigned a = t;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const float ov = __builtin_bit_cast(float, shfl(__builtin_bit_cast(uint32_t, v), (int)(lane ^ (unsigned)d)));
    const unsigned oa = shfl(a, (int)(lane ^ (unsigned)d));
    if (ov > v || (ov == v && oa < a)) v = ov, a = oa;
  }
  if (lane == 0) bv[wv] = v, bt[wv] = a;
  __syncthreads();
  if (t == 0) {
    for (int k = 1; k < 4; k++)
      if (bv[k] > v || (bv[k] == v && bt[k] < a)) v = bv[k], a = bt[k];
    thr[blockIdx.x] = (uint8_t)(v > -1.0f ? a : 0u);
  }
}

/* the 1-px frame of every image := 0 (what "gs_sobel into a zeroed image" leaves there).
 * grid (ceil((2w+2h)/256), n frames) */
__global__ __launch_bounds__(256) void k_zero_frame(uint8_t *img, unsigned w, unsigned h,
                                                    size_t frame_bytes) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  uint8_t *f = img + (size_t)blockIdx.y * frame_bytes;
  if (i < w) f[i] = 0;
  else if (i < 2 * w) f[(size_t)(h - 1) * w + (i - w)] = 0;
  else if (i < 2 * w + h) f[(size_t)(i - 2 * w) * w] = 0;
  else if (i < 2 * w + 2 * h) f[(size_t)(i - 2 * w - h) * w + (w - 1)] = 0;
}

/* sums[f] += sum over bytes of (index+1)*(byte+1) mod 2^64 (order independent). */
__global__ __launch_bounds__(256) void k_checksum(const uint8_t *img, size_t frame_bytes,
                                                  unsigned long long *sums) {
  const uint8_t *base = img + (size_t)blockIdx.y * frame_bytes;
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < frame_bytes;
       i += (size_t)gridDim.x * 256u)
    acc += (unsigned long long)(i + 1) * ((unsigned long long)base[i] + 1ull);
  /* wave reduce (two 32-bit halves), then one atomic per wave */
  uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t olo = shfl(lo, (int)(lane_id() ^ (unsigned)d));
    const uint32_t ohi = shfl(hi, (int)(lane_id() ^ (unsigned)d));
    const unsigned long long t =
        (((unsigned long long)hi << 32) | lo) + (((unsigned long long)ohi << 32) | olo);
    lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
  }
  if (lane_id() == 0) atomicAdd(&sums[blockIdx.y], ((unsigned long long)hi << 32) | lo);
}

/* ---- synthetic block-noise frames on device (SURVEY.md 8c generator) -----------------
 * xorshift32 is a linear map over GF(2); jump[k] holds the 32x32 bit matrix of 2^k steps
 * (column images, computed on the host), so each thread jumps straight to its run of RUN
 * pixels and then iterates.  Bit-identical to the CPU generator. */
struct SynthJump { uint32_t col[32][32]; }; /* col[k][b] = M^(2^k) applied to bit b */

GS_DEV uint32_t xs32(uint32_t x) {
  x ^= x << 13;
  x ^= x >> 17;
  x ^= x << 5;
  return x;
}
GS_DEV uint32_t xs_jump(const SynthJump *J, uint32_t s, uint64_t steps) {
  for (int k = 0; k < 32 && steps; k++, steps >>= 1)
    if (steps & 1) {
      uint32_t r = 0;
      for (int b = 0; b < 32; b++)
        if ((s >> b) & 1u) r ^= J->col[k][b];
      s = r;
    }
  return s;
}
constexpr unsigned kSynthRun = 256;
/* grid (ceil(w*h/(256*RUN)), n), block 256.  levels: n * bw*bh bytes scratch */
__global__ __launch_bounds__(256) void k_synth_levels(uint8_t *levels, unsigned nlev,
                                                      uint32_t seed0, const SynthJump *J) {
  const unsigned f = blockIdx.y;
  const unsigned i0 = (blockIdx.x * 256u + threadIdx.x) * kSynthRun;
  if (i0 >= nlev) return;
  uint32_t seed = seed0 + f;
  uint32_t s = xs_jump(J, seed ? seed : 1u, i0);
  for (unsigned i = i0; i < i0 + kSynthRun && i < nlev; i++) {
    s = xs32(s);
    levels[(size_t)f * nlev + i] = (uint8_t)(s & 0xff);
  }
}
__global__ __launch_bounds__(256) void k_synth_pixels(uint8_t *dst, const uint8_t *levels,
                                                      unsigned w, unsigned h, uint32_t seed0,
                                                      const SynthJump *J) {
  const unsigned f = blockIdx.y;
  const unsigned bw = (w + 31) / 32, bh = (h + 31) / 32, nlev = bw * bh;
  const size_t npx = (size_t)w * h;
  const size_t i0 = ((size_t)blockIdx.x * 256u + threadIdx.x) * kSynthRun;
  if (i0 >= npx) return;
  uint32_t seed = seed0 + f;
  uint32_t s = xs_jump(J, seed ? seed : 1u, (uint64_t)nlev + i0);
  const uint8_t *lv = levels + (size_t)f * nlev;
  uint8_t *out = dst + (size_t)f * npx;
  /* i0 is a multiple of 256 and frames are whole dwords apart when w*h % 4 == 0: dword stores */
  const bool aligned = (((uintptr_t)out) & 3) == 0;
  uint32_t packed = 0;
  for (size_t i = i0; i < i0 + kSynthRun && i < npx; i++) {
    s = xs32(s);
    const unsigned x = (unsigned)(i % w), y = (unsigned)(i / w);
    int v = (int)lv[(y / 32) * bw + x / 32] + (int)(s & 15u) - 8;
    const uint32_t b = (uint32_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    if (aligned && i0 + kSynthRun <= npx) {
      packed |= b << (8 * (i & 3));
      if ((i & 3) == 3) *(uint32_t *)(out + i - 3) = packed, packed = 0;
    } else {
      out[i] = (uint8_t)b;
    }
  }
}

}  // namespace gs
#endif
