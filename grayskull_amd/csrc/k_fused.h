/*
 * k_fused.h -- the config-2 pipeline kernel: gs_blur(R) -> gs_sobel -> histogram in one pass
 * (reference semantics grayskull.h:268-283, :306-320, :199-203).  Built by gs_fused.cpp.
 */
#ifndef GS_K_FUSED_H
#define GS_K_FUSED_H
#include "k_strip.h"

namespace gs {

/* ------------------------------------------------------------------ fused blur -> sobel -> histogram */
/* The config-2 chain (gs_blur(R); gs_sobel; histogram for gs_otsu_threshold) in ONE pass over the
 * frame: 1 B/px read + 1 B/px written instead of 2+2+1.  Per source row the lane forms the
 * (2R+1)-tap horizontal sums for pixels -2..17, keeps their running vertical sum, divides (exact
 * 2^24 multipliers, chosen per pixel / per row where the window is clipped: every divisor is
 * rows_in_image * cols_in_image, ref :275-281), and feeds the blurred row -- never written to
 * memory -- straight into the sobel recurrence.  The sobel bytes go to dst and into an
 * LDS-privatised histogram (32 bank-spread copies, see k_hist_partial); each block leaves 256
 * partial counts for k_hist_reduce.  Bit-identical to the separate calls (tests).  Columns 0 and
 * w-1 of dst receive junk here; the launcher zeroes the 1-px frame afterwards (config 2 runs
 * gs_sobel into a zeroed image), and the histogram counts those frame pixels as 0 analytically. */
constexpr uint32_t blur_k24(unsigned cx, unsigned cy) { return (0x1000000u + cx * cy - 1u) / (cx * cy); }

template <int R, unsigned CX>
GS_DEV uint32_t blur_mul_for_rows(unsigned cy) { /* cy in [R+1, 2R+1], wave-uniform */
  uint32_t m = blur_k24(CX, 2 * R + 1);
#pragma unroll
  for (unsigned c = R + 1; c < 2 * R + 1; c++) m = cy == c ? blur_k24(CX, c) : m;
  return m;
}

template <int R>
GS_DEV void blur_hsum10(const uint32_t (&U)[12], uint32_t (&H)[10]) { /* pairs = px -2..17 */
  uint32_t A[13]; /* A[j+1] = pair starting one px after U[j], j = -1..11 (ends zero-extended) */
  if constexpr (R >= 3) A[0] = alignbit(U[0], 0u, 16), A[12] = alignbit(0u, U[11], 16);
#pragma unroll
  for (int j = 0; j <= 10; j++) A[j + 1] = alignbit(U[j + 1], U[j], 16);
  if constexpr (R == 1) {
#pragma unroll
    for (int k = 0; k < 10; k++) H[k] = add2(A[k + 1], U[k + 1], A[k + 2]);
  } else {
    uint32_t Q[11]; /* Q[j] = U[j] + A[j+1]: the 2-px sums starting at both pixels of pair j */
#pragma unroll
    for (int j = 0; j <= 10; j++) Q[j] = add2(U[j], A[j + 1]);
#pragma unroll
    for (int k = 0; k < 10; k++) {
      const int j = k + 1; /* U[j] = pair k */
      uint32_t t = add2(Q[j - 1], Q[j], U[j + 1]);              /* -2 .. +2: one v_add3_u32 */
      if constexpr (R >= 3) t = add2(t, A[j - 1], A[j + 2]);     /* -3, +3 */
      H[k] = t;
    }
  }
}

/* grid like the strip kernels; partial: [frame][blockIdx.y * gridDim.x + blockIdx.x][256] */
/* HIST = false: gs_blur + gs_sobel only (gsh_blur_sobel_batch): no LDS, no histogram, `partial` unused */
/* the spare ring slot (see the kernel).  Without it the kernel needs MORE registers, not fewer (R = 2 without the histogram
 * half: 145 against 138; the odd unroll period keeps the shifted sobel history alive): experiment hook only */
#ifndef GS_FUSED_SPARE
#define GS_FUSED_SPARE(hist) true
#endif
#ifndef GS_FUSED_VGPR_ATTR
#define GS_FUSED_VGPR_ATTR /* experiment hook: -DGS_FUSED_VGPR_ATTR='__attribute__((amdgpu_num_vgpr(144)))' */
#endif
/* RAGGED (w % 16 != 0, without the histogram half): the strips of k_strip.h with the tail strip anchored at w - 16.  The lane
 * left of the tail lane may own pixels -- its own last R and the two it blurs for its right neighbour's sobel taps -- that are
 * less than R columns from the right edge, so the right-hand divisors are chosen per lane from the distance to the edge. */
/* MINW: waves per SIMD the register allocation has to leave room for (__launch_bounds__' second argument) */
template <int R, bool HIST = true, int RG = 0, int MINW = 1>
__global__ __launch_bounds__(256, MINW) GS_FUSED_VGPR_ATTR void k_blur_sobel_hist16(uint8_t *dst, const uint8_t *src,
                                                           unsigned w, unsigned h, unsigned T,
                                                           size_t frame_bytes, unsigned *partial) {
  constexpr bool RAGGED = RG != 0;
  static_assert(!(HIST && RAGGED), "ragged rows take the histogram as a separate pass");
  constexpr int N = 2 * R + 1;
  __shared__ unsigned lh[HIST ? 256 * 32 : 1];
  const unsigned tid = threadIdx.y * blockDim.x + threadIdx.x, copy = tid & 31u;
  if constexpr (HIST) {
    for (unsigned i = tid; i < 256 * 32; i += 256) lh[i] = 0;
    __syncthreads();
  }
  const Strip<false, RG> S(src, dst, w, h, frame_bytes);
  const int y0 = 1 + (int)(S.band * T);
  if (y0 < (int)h - 1 && !S.wave_outside()) { /* wave-uniform; no early return: every wave reaches the barrier */
    const int nrows = ((int)h - 1 - y0) < (int)T ? ((int)h - 1 - y0) : (int)T;
    const bool first = S.x0 == 0, last = S.x0 + 16 == w, inimg = S.in_image();
    /* RAGGED: is own pixel 16 - R + q (q = 0 .. R + 1) exactly R + 1 + i columns' worth from the right edge? (lane constants) */
    bool edge[R + 2][R];
    if constexpr (RAGGED) {
      static_for<R + 2>([&](auto Q) {
        constexpr int q = decltype(Q)::value;
        const int d = (int)w - 1 - (int)(S.x0 + 16 - R + q); /* columns between the pixel and the last one */
        static_for<R>([&](auto I) { edge[q][decltype(I)::value] = inimg && d == decltype(I)::value; });
      });
    }
    /* N+1 ring slots: the new row lands in the free slot and the unroll period N+1 is even, so
     * the sobel history alternates (step<parity>) without register moves */
    constexpr bool SPARE = R <= 2 && GS_FUSED_SPARE(HIST); /* R = 3: the 8th slot would cost the third wave per SIMD */
    constexpr int NS = SPARE ? N + 1 : N, P0 = SPARE ? 1 : 0; /* P0: slot of the first prologue row */
    uint32_t ring[NS][10], V[10];
    SobelState st;
    /* blurred row b as u16 pairs for pixels -2..17, placed where sobel_hpass expects U[1..10] */
    auto blurred = [&](int b, uint32_t(&UB)[12]) {
      const int ya = b - R < 0 ? 0 : b - R, yb = b + R > (int)h - 1 ? (int)h - 1 : b + R;
      const unsigned cy = (unsigned)(yb - ya + 1);
      const uint32_t mC = blur_mul_for_rows<R, N>(cy);
      uint32_t mL[R], mR[R + 2];
      static_for<R>([&](auto Q) {
        constexpr int q = decltype(Q)::value;
        mL[q] = first ? blur_mul_for_rows<R, R + 1 + q>(cy) : mC;
        mR[q] = last ? blur_mul_for_rows<R, 2 * R - q>(cy) : mC;
      });
      mR[R] = mR[R + 1] = mC;
      if constexpr (RAGGED) {
        static_for<R + 2>([&](auto Q) {
          constexpr int q = decltype(Q)::value;
          uint32_t m = mC;
          static_for<R>([&](auto I) { m = edge[q][decltype(I)::value] ? blur_mul_for_rows<R, R + 1 + decltype(I)::value>(cy) : m; });
          mR[q] = m;
        });
      }
      UB[0] = 0, UB[11] = 0;
#pragma unroll
      for (int k = 0; k < 10; k++) { /* pair k = own pixels (2k-2, 2k-1) */
        uint32_t pr[2];
#pragma unroll
        for (int hlf = 0; hlf < 2; hlf++) {
          const int q = 2 * k - 2 + hlf; /* own pixel index -2..17 */
          const uint32_t sv = hlf ? (V[k] >> 16) : (V[k] & 0xffffu);
          constexpr int qr = RAGGED ? 18 : 16; /* RAGGED: pixels 16, 17 (a neighbour's) may lie at the edge as well */
          const uint32_t m = (q >= 0 && q < R) ? mL[(q >= 0 && q < R) ? q : 0]
                             : (q >= 16 - R && q < qr) ? mR[(q >= 16 - R && q < qr) ? q - (16 - R) : 0]
                                                       : mC;
          pr[hlf] = sv * m; /* quotient = byte 3 */
        }
        UB[k + 1] = perm_b32(pr[1], pr[0], 0x0c070c03u);
      }
    };
    /* prologue: source rows y0-1-R .. y0+R give blurred rows y0-1 and y0 */
#pragma unroll
    for (int k = 0; k < 10; k++) V[k] = 0;
    uint32_t UB0[12], UB1[12];
    static_for<N>([&](auto K) {
      constexpr int kk = decltype(K)::value;
      uint32_t U[12];
      S.unpack(S.load(y0 - 1 - R + kk), U);
      blur_hsum10<R>(U, ring[kk + P0]);
#pragma unroll
      for (int k = 0; k < 10; k++) V[k] = add2(V[k], ring[kk + P0][k]);
    });
    blurred(y0 - 1, UB0);
    {
      uint32_t U[12], Hn[10];
      S.unpack(S.load(y0 + R), U);
      blur_hsum10<R>(U, Hn); /* enters slot 0 (SPARE: the free slot); the oldest row (slot P0) leaves */
#pragma unroll
      for (int k = 0; k < 10; k++) V[k] = sub2(add2(V[k], Hn[k]), ring[P0][k]), ring[0][k] = Hn[k];
    }
    blurred(y0, UB1);
    st.init(UB0, UB1);
    const uint32_t cb = copy << 2; /* byte offset of this lane's histogram copy inside a bin */
    /* per-lane increments, fixed for the kernel: lanes outside the image and the two frame columns count 0 */
    const unsigned inc_in = inimg ? 1u : 0u, inc_first = (inimg && !first) ? 1u : 0u, inc_last = (inimg && !last) ? 1u : 0u;

    /* The histogram of row i is added during row i+1 (software pipelining): an LDS atomic takes the
     * CU's LDS pipe ~4 cycles per wave-instruction (scripts/ubench_valu.cpp, pure stream), 16 of them per
     * row and wave.  Issued in a burst after
     * the row's last results (hipcc sinks instructions nobody waits for to the end of the block),
     * they fill the LDS queue and stall the wave.  So the previous row's 16 atomics are threaded
     * through values of THIS row's arithmetic (lds_add_through): the even pixels' through the new
     * vertical sums V[k], the odd pixels' through the sobel column sums C[k+1]. */
    uint32_t Mp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned livep = 0; /* wave-uniform: the previous row counts */
    auto hist_lo = [&](int k, uint32_t &through) {
      const unsigned inc = (k == 0 ? inc_first : inc_in) & livep;
      lds_add_through(lh, mad_u32_u16_lo(Mp[k], 128u, cb), inc, through); /* LDS byte offset = bin*128 + copy*4 */
    };
    auto hist_hi = [&](int k, uint32_t &through) {
      const unsigned inc = (k == 7 ? inc_last : inc_in) & livep;
      lds_add_through(lh, mad_u32_u16_hi(Mp[k], 128u, cb), inc, through);
    };

    strip_rows<NS, false, /*EXITS=*/false>(S, y0, nrows, R + 1, S.load(y0 + R + 1), [&](auto I, int i, const uint32_t(&U)[12]) {
      /* iteration I, SPARE: slot I+1 is free (its row left last iteration), slot I+2 holds the
       * oldest row; otherwise the new row replaces the oldest (slot I+1) */
      constexpr int fr = (decltype(I)::value + 1) % NS, old = (decltype(I)::value + 1 + P0) % NS;
      uint32_t UB[12], M[8], Hn[10];
      blur_hsum10<R>(U, Hn);
#pragma unroll
      for (int k = 0; k < 10; k++) {
        V[k] = sub2(add2(V[k], Hn[k]), ring[old][k]), ring[fr][k] = Hn[k];
        if constexpr (HIST) {
          if (k >= 1 && k <= 8) hist_lo(k - 1, V[k]);
        }
      }
      blurred(y0 + i + 1, UB);
      auto hook = [&](int j, uint32_t &c) { /* C[1..9]: the odd pixels of pairs 0..7 ride on C[1..8] */
        if (j <= 8) hist_hi(j - 1, c);
      };
      U4 o; /* without the histogram only the bytes are needed (saturating-mad form of the clamp) */
      if constexpr (NS % 2 == 0) {
        if constexpr (HIST) o = st.template step<decltype(I)::value & 1>(UB, M, hook);
        else o = st.template step<decltype(I)::value & 1>(UB);
      } else {
        if constexpr (HIST) o = st.step_shift(UB, M, hook);
        else o = st.step_shift(UB);
      }
      /* histogram, branch-free: lanes outside the image, the two frame columns and the dropped
       * rows of the last group add 0. */
      if constexpr (HIST) {
#pragma unroll
        for (int k = 0; k < 8; k++) Mp[k] = M[k];
        livep = i < nrows ? 1u : 0u; /* wave-uniform: rows past the band end are dropped */
      }
      return o;
    });
    if constexpr (HIST) { /* the last row */
      uint32_t t = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) hist_lo(k, t), hist_hi(k, t);
    }
  }
  if constexpr (HIST) {
    lds_drain(); /* the threaded atomics are not in hipcc's lgkmcnt bookkeeping */
    __syncthreads();
    unsigned acc = 0;
#pragma unroll 8
    for (unsigned k = 0; k < 32; k++) acc += lh[tid * 32u + ((k + tid) & 31u)];
    const size_t blk = (size_t)blockIdx.z * gridDim.x * gridDim.y + (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    partial[blk * 256u + tid] = acc;
  }
}

}  // namespace gs
#endif
