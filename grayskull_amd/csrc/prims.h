/*
 * prims.h -- wave64 / packed-math primitives used by the gfx950 kernels.
 *
 * One spelling per primitive; the HIP build maps each to the CDNA4 instruction named in
 * its comment, the -DGS_EMU build (tests/emu, host fibers) restates it in scalar C++.
 */
#ifndef GS_PRIMS_H
#define GS_PRIMS_H

/* The compile-time experiment hooks (each `#ifndef X / #define X <default>` further down in the sources) may only be set in
 * builds that say so: a release library cannot be built with one of them flipped by accident. */
#if !defined(GS_EXPERIMENT) && (defined(GS_FUSED_SPARE) || defined(GS_FUSED_VGPR_ATTR) || defined(GS_EVENT_FLAGS) ||       \
                                defined(GS_ORDER_EVENT_FLAGS) || defined(GS_LOAD_AUX) || defined(GS_STORE_AUX) ||         \
                                defined(GS_LBP_PREFETCH) || defined(GS_MAD2_OPAQUE) || defined(GS_LBP_TILE_ODD_STRIDE) ||   \
                                defined(GS_LBP_SENS))
#error "experiment hook set without -DGS_EXPERIMENT (make variant / make experiment add it)"
#endif

#ifdef GS_EMU
#include "hip_emu.h"
#define GS_DYN_LDS(name) char *name = emu::S().dyn_lds
#else
#include <hip/hip_runtime.h>
#define GS_LAUNCH(kernel, grid, block, shmem, stream, ...) \
  hipLaunchKernelGGL(kernel, (grid), (block), (shmem), (stream), __VA_ARGS__)
#define GS_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) char name[]
#endif

#include <stdint.h>
#include <string.h>

#define GS_DEV __device__ __forceinline__
#define GS_HD __host__ __device__ __forceinline__
#ifndef GS_MAD2_OPAQUE
#define GS_MAD2_OPAQUE 1
#endif

namespace gs {

constexpr int kWave = 64; /* CDNA wavefront */

struct alignas(16) U4 { uint32_t x, y, z, w; };

/* Raw buffer access (buffer_load/store ... offen through a 128-bit V#): the per-lane byte offset
 * is range-checked by the hardware against the frame size, out-of-range lanes load 0 and drop
 * their stores.  That makes every strip-kernel memory op branch-free (no exec-mask diamonds, so
 * hipcc can keep loads in flight across the arithmetic with counted waits) and gives the zero
 * fill outside the image for free.  kOOB is an offset no frame (< 2 GiB) can contain. */
constexpr uint32_t kOOB = 0x80000000u;

#ifdef GS_EMU
/* ------------------------------------------------------------------ emulation */
GS_DEV unsigned lane_id() { return emu::lane_id(); }
/* set bits of `m` below this lane (v_mbcnt_lo/hi) */
GS_DEV unsigned mbcnt(uint64_t m) { return (unsigned)__builtin_popcountll(m & ((1ull << emu::lane_id()) - 1ull)); }
GS_DEV uint64_t ballot(bool p) {
  return emu::wave_exchange(p ? 1 : 0, [](const uint64_t *s, const bool *v) {
    uint64_t m = 0;
    for (int i = 0; i < 64; i++)
      if (v[i] && s[i]) m |= 1ull << i;
    return m;
  });
}
GS_DEV uint32_t wave_shr1(uint32_t x, uint32_t fill) {
  unsigned l = lane_id();
  return emu::wave_exchange(x, [=](const uint64_t *s, const bool *v) {
    return (l > 0 && v[l - 1]) ? (uint32_t)s[l - 1] : fill;
  });
}
GS_DEV uint32_t wave_shl1(uint32_t x, uint32_t fill) {
  unsigned l = lane_id();
  return emu::wave_exchange(x, [=](const uint64_t *s, const bool *v) {
    return (l < 63 && v[l + 1]) ? (uint32_t)s[l + 1] : fill;
  });
}
GS_DEV uint32_t shfl(uint32_t x, int src) {
  return emu::wave_exchange(x, [=](const uint64_t *s, const bool *v) {
    return v[src & 63] ? (uint32_t)s[src & 63] : 0u;
  });
}
GS_DEV uint32_t perm_b32(uint32_t hi, uint32_t lo, uint32_t sel) { /* v_perm_b32 */
  uint64_t src = ((uint64_t)hi << 32) | lo;
  uint32_t r = 0;
  for (int i = 0; i < 4; i++) {
    unsigned s = (sel >> (8 * i)) & 0xff;
    unsigned b = s <= 7 ? (unsigned)(src >> (8 * s)) & 0xff : (s == 0x0c ? 0u : (s >= 0x0d ? 0xffu : 0u));
    r |= b << (8 * i);
  }
  return r;
}
GS_DEV uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { /* v_alignbit_b32 */
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31));
}
struct BufRsrc { uint8_t *base; uint32_t n; };
GS_DEV BufRsrc make_buf(const void *base, size_t bytes) {
  return BufRsrc{(uint8_t *)base, (uint32_t)(bytes > 0x7fffffffu ? 0x7fffffffu : bytes)};
}
GS_DEV void emu_buf_check(const BufRsrc &b, uint32_t off, uint32_t sz) {
  if (off < b.n && off + sz > b.n) { fprintf(stderr, "emu: partially out-of-range buffer access\n"); abort(); }
}
/* multi-dword loads are range-checked per component (dword) like the hardware does: a component that lies wholly
 * behind the end reads 0, one that straddles the end is an error of the kernel */
GS_DEV U4 buf_load16(const BufRsrc &b, uint32_t off) {
  uint32_t d[4] = {0, 0, 0, 0};
  for (uint32_t k = 0; k < 4; k++) {
    const uint64_t o = (uint64_t)off + 4u * k;
    if (o >= b.n) continue;
    emu_buf_check(b, (uint32_t)o, 4);
    memcpy(&d[k], b.base + o, 4);
  }
  return U4{d[0], d[1], d[2], d[3]};
}
struct U2 { uint32_t x, y; };
GS_DEV U2 buf_load8(const BufRsrc &b, uint32_t off) {
  uint32_t d[2] = {0, 0};
  for (uint32_t k = 0; k < 2; k++) {
    const uint64_t o = (uint64_t)off + 4u * k;
    if (o >= b.n) continue;
    emu_buf_check(b, (uint32_t)o, 4);
    memcpy(&d[k], b.base + o, 4);
  }
  return U2{d[0], d[1]};
}
GS_DEV uint32_t buf_load1(const BufRsrc &b, uint32_t off) { return off < b.n ? b.base[off] : 0u; }
GS_DEV uint32_t buf_load4(const BufRsrc &b, uint32_t off) {
  uint32_t v = 0;
  emu_buf_check(b, off, 4);
  if (off < b.n) memcpy(&v, b.base + off, 4);
  return v;
}
template <bool NT> GS_DEV U4 buf_load16_pol(const BufRsrc &b, uint32_t off) { return buf_load16(b, off); }
template <bool NT> GS_DEV uint32_t buf_load4_pol(const BufRsrc &b, uint32_t off) { return buf_load4(b, off); }
GS_DEV void store_u32x4(void *p, const U4 &v) { memcpy(p, &v, 16); }
GS_DEV void store_u32x4_any(void *p, const U4 &v) { memcpy(p, &v, 16); }
GS_DEV U4 load_u32x4_any(const void *p) { U4 v; memcpy(&v, p, 16); return v; }
GS_DEV void store_u32x2_any(void *p, uint32_t lo, uint32_t hi) { memcpy(p, &lo, 4), memcpy((char *)p + 4, &hi, 4); }
GS_DEV void buf_store16(const BufRsrc &b, uint32_t off, const U4 &v) {
  emu_buf_check(b, off, 16);
  if (off < b.n) memcpy(b.base + off, &v, 16);
}
GS_DEV void buf_store16_wb(const BufRsrc &b, uint32_t off, const U4 &v) { buf_store16(b, off, v); }
GS_DEV void buf_store_n(const BufRsrc &b, uint32_t off, uint64_t v, uint32_t n) { /* the low n bytes of v */
  emu_buf_check(b, off, n);
  if (off < b.n) memcpy(b.base + off, &v, n);
}
GS_DEV void buf_store8(const BufRsrc &b, uint32_t off, uint32_t lo, uint32_t hi) { buf_store_n(b, off, lo | ((uint64_t)hi << 32), 8); }
GS_DEV void buf_store4(const BufRsrc &b, uint32_t off, uint32_t v) { buf_store_n(b, off, v, 4); }
GS_DEV void buf_store2(const BufRsrc &b, uint32_t off, uint32_t v) { buf_store_n(b, off, v, 2); }
GS_DEV void buf_store1(const BufRsrc &b, uint32_t off, uint32_t v) { buf_store_n(b, off, v, 1); }
GS_DEV uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { /* v_alignbyte_b32 */
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (sh & 3)));
}
/* dword gather: per-lane byte offset + wave-uniform byte offset (SGPR soffset on the GPU) */
GS_DEV uint32_t buf_gather4(const BufRsrc &b, uint32_t voff, uint32_t soff) { return buf_load4(b, voff + soff); }
GS_DEV uint32_t uniform(uint32_t x) { return x; } /* v_readfirstlane_b32 on the GPU */
#define GS_PK2(expr_lo, expr_hi) ((uint32_t)((expr_lo) & 0xffffu) | ((uint32_t)((expr_hi) & 0xffffu) << 16))
GS_DEV uint32_t pk_add_u16(uint32_t a, uint32_t b) { return GS_PK2((a & 0xffff) + (b & 0xffff), (a >> 16) + (b >> 16)); }
GS_DEV uint32_t pk_sub_u16(uint32_t a, uint32_t b) { return GS_PK2((a & 0xffff) - (b & 0xffff), (a >> 16) - (b >> 16)); }
GS_DEV uint32_t pk_mul_u16(uint32_t a, uint32_t b) { return GS_PK2((a & 0xffff) * (b & 0xffff), (a >> 16) * (b >> 16)); }
GS_DEV uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c) { return GS_PK2((a & 0xffff) * (b & 0xffff) + (c & 0xffff), (a >> 16) * (b >> 16) + (c >> 16)); }
GS_DEV uint32_t pk_mad2_u16(uint32_t a, uint32_t c) { return GS_PK2(2 * (a & 0xffff) + (c & 0xffff), 2 * (a >> 16) + (c >> 16)); }
GS_DEV uint32_t pk_mad_u16_s(uint32_t a, uint32_t b, uint32_t c) { return pk_mad_u16(a, b, c); }
GS_DEV uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) {
  for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 0xffu) * ((b >> (8 * i)) & 0xffu);
  return c;
}
GS_DEV uint32_t udot2_ones(uint32_t a, uint32_t c) { return (a & 0xffffu) + (a >> 16) + c; }
GS_DEV uint32_t load_u32_unaligned(const uint8_t *p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}
GS_DEV uint32_t mad_u32_u16_lo(uint32_t a, uint32_t b, uint32_t c) { return (a & 0xffffu) * (b & 0xffffu) + c; }
GS_DEV uint32_t mad_u32_u16_hi(uint32_t a, uint32_t b, uint32_t c) { return (a >> 16) * (b & 0xffffu) + c; }
GS_DEV void sched_fence() {}
#define GS_SCHED_GROUP(mask, n) ((void)0)
GS_DEV uint32_t opaque(uint32_t x) { return x; }
GS_DEV uint32_t push_ge_u32(uint32_t acc, uint32_t a, uint32_t b) { return acc * 2u + (a >= b ? 1u : 0u); }
GS_DEV uint32_t push_gt_u32(uint32_t acc, uint32_t a, uint32_t b) { return acc * 2u + (a > b ? 1u : 0u); }
GS_DEV void lds_add_through(unsigned *lds_base, uint32_t byte_off, uint32_t value, uint32_t &through) {
  (void)through;
  atomicAdd((unsigned *)((char *)lds_base + byte_off), value);
}
GS_DEV void lds_drain() {}
GS_DEV uint32_t pk_shl_u16(uint32_t a, unsigned s) { return GS_PK2((a & 0xffff) << s, (a >> 16) << s); }
GS_DEV uint32_t pk_shr_u16(uint32_t a, unsigned s) { return GS_PK2((a & 0xffff) >> s, (a >> 16) >> s); }
GS_DEV uint32_t pk_min_u16(uint32_t a, uint32_t b) {
  uint32_t al = a & 0xffff, bl = b & 0xffff, ah = a >> 16, bh = b >> 16;
  return GS_PK2(al < bl ? al : bl, ah < bh ? ah : bh);
}
GS_DEV uint32_t pk_subsat_u16(uint32_t a, uint32_t b) { /* max(a - b, 0) per half */
  uint32_t al = a & 0xffff, bl = b & 0xffff, ah = a >> 16, bh = b >> 16;
  return GS_PK2(al > bl ? al - bl : 0u, ah > bh ? ah - bh : 0u);
}
GS_DEV uint32_t pk_shl7_sat_u16(uint32_t a) { /* min(a * 128, 65535) per half */
  uint32_t al = (a & 0xffff) * 128u, ah = (a >> 16) * 128u;
  return GS_PK2(al > 0xffffu ? 0xffffu : al, ah > 0xffffu ? 0xffffu : ah);
}
GS_DEV uint32_t pk_max_u16(uint32_t a, uint32_t b) {
  uint32_t al = a & 0xffff, bl = b & 0xffff, ah = a >> 16, bh = b >> 16;
  return GS_PK2(al > bl ? al : bl, ah > bh ? ah : bh);
}
GS_DEV uint32_t pk_abs_i16(uint32_t a) {
  int al = (int16_t)(a & 0xffff), ah = (int16_t)(a >> 16);
  return GS_PK2((uint32_t)(al < 0 ? -al : al), (uint32_t)(ah < 0 ? -ah : ah));
}
GS_DEV uint32_t pk_max_i16(uint32_t a, uint32_t b) {
  int al = (int16_t)(a & 0xffff), ah = (int16_t)(a >> 16), bl = (int16_t)(b & 0xffff), bh = (int16_t)(b >> 16);
  return GS_PK2((uint32_t)(al > bl ? al : bl), (uint32_t)(ah > bh ? ah : bh));
}
GS_DEV uint32_t pk_sar_i16(uint32_t a, unsigned s) {
  int al = (int16_t)(a & 0xffff), ah = (int16_t)(a >> 16);
  return GS_PK2((uint32_t)(al >> s), (uint32_t)(ah >> s));
}
/* v_mfma_i32_32x32x32_i8: D = A (32 x 32, signed bytes) . B (32 x 32) + C for the whole wave.  Operand slots: lane l
 * holds, in its 16 bytes, A[i = l & 31][slot (l >> 5, byte)] resp. B[slot (l >> 5, byte)][j = l & 31]; the hardware
 * pairs slot s of A with slot s of B (which k a slot stands for never matters to a caller that fills both operands
 * by the same rule).  C / D: register r of lane l is row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31. */
GS_DEV void mfma_i32_32x32x32_i8(const U4 &a, const U4 &b, int32_t (&c)[16]) {
  const unsigned l = lane_id();
  uint64_t A[64][2], B[64][2];
  const uint64_t av[2] = {a.x | ((uint64_t)a.y << 32), a.z | ((uint64_t)a.w << 32)};
  const uint64_t bv[2] = {b.x | ((uint64_t)b.y << 32), b.z | ((uint64_t)b.w << 32)};
  for (int h = 0; h < 2; h++) {
    emu::wave_exchange(av[h], [&](const uint64_t *s, const bool *v) { for (int i = 0; i < 64; i++) A[i][h] = v[i] ? s[i] : 0; return 0; });
    emu::wave_exchange(bv[h], [&](const uint64_t *s, const bool *v) { for (int i = 0; i < 64; i++) B[i][h] = v[i] ? s[i] : 0; return 0; });
  }
  auto byte_of = [](const uint64_t (&q)[2], int k) { return (int)(int8_t)((q[k >> 3] >> (8 * (k & 7))) & 0xff); };
  for (int r = 0; r < 16; r++) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (int)(l >> 5), col = (int)(l & 31);
    int32_t acc = c[r];
    for (int g = 0; g < 2; g++)
      for (int k = 0; k < 16; k++) acc += byte_of(A[row + 32 * g], k) * byte_of(B[col + 32 * g], k);
    c[r] = acc;
  }
}
#else
/* ------------------------------------------------------------------ gfx950 */
GS_DEV unsigned lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
/* set bits of `m` below this lane */
GS_DEV unsigned mbcnt(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
GS_DEV uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
/* v_mov_b32_dpp wave_shr:1 -- lane i receives lane i-1; lane 0 keeps `fill` */
GS_DEV uint32_t wave_shr1(uint32_t x, uint32_t fill) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)x, 0x138, 0xf, 0xf, false);
}
/* v_mov_b32_dpp wave_shl:1 -- lane i receives lane i+1; lane 63 keeps `fill` */
GS_DEV uint32_t wave_shl1(uint32_t x, uint32_t fill) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)x, 0x130, 0xf, 0xf, false);
}
GS_DEV uint32_t shfl(uint32_t x, int src) { /* ds_bpermute_b32 */
  return (uint32_t)__builtin_amdgcn_ds_bpermute((src & 63) << 2, (int)x);
}
GS_DEV uint32_t perm_b32(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
GS_DEV uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }

/* cache-policy bits of the strip kernels' buffer ops (aux: 1 = sc0, 2 = nt, 16 = sc1) */
#ifndef GS_LOAD_AUX
#define GS_LOAD_AUX 0
#endif
#ifndef GS_STORE_AUX
#define GS_STORE_AUX 2 /* nt: results are streamed out, never re-read by this kernel; measured
                          -4 % (sobel) ... -9 % (blur) kernel time vs default policy on MI355X */
#endif
typedef unsigned int gs_u32x4 __attribute__((ext_vector_type(4)));
struct BufRsrc { __amdgpu_buffer_rsrc_t r; };
/* base/bytes must be wave-uniform (kernel arguments, blockIdx): the V# then lives in SGPRs */
GS_DEV BufRsrc make_buf(const void *base, size_t bytes) {
  return BufRsrc{__builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)(bytes > 0x7fffffffu ? 0x7fffffffu : bytes), 0x00020000)};
}
GS_DEV U4 buf_load16(const BufRsrc &b, uint32_t off) { /* buffer_load_dwordx4 offen */
  const gs_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b.r, (int)off, 0, GS_LOAD_AUX);
  return U4{v.x, v.y, v.z, v.w};
}
GS_DEV uint32_t buf_load4(const BufRsrc &b, uint32_t off) { /* buffer_load_dword offen */
  return __builtin_amdgcn_raw_buffer_load_b32(b.r, (int)off, 0, GS_LOAD_AUX);
}
/* NT = true: streaming (nt) policy for data that is read once and is too large to be found in the Infinity Cache again */
template <bool NT> GS_DEV U4 buf_load16_pol(const BufRsrc &b, uint32_t off) {
  const gs_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b.r, (int)off, 0, NT ? 2 : GS_LOAD_AUX);
  return U4{v.x, v.y, v.z, v.w};
}
template <bool NT> GS_DEV uint32_t buf_load4_pol(const BufRsrc &b, uint32_t off) {
  return __builtin_amdgcn_raw_buffer_load_b32(b.r, (int)off, 0, NT ? 2 : GS_LOAD_AUX);
}
struct U2 { uint32_t x, y; };
GS_DEV U2 buf_load8(const BufRsrc &b, uint32_t off) { /* buffer_load_dwordx2 offen */
  typedef unsigned int gs_u32x2_ __attribute__((ext_vector_type(2)));
  const gs_u32x2_ v = __builtin_amdgcn_raw_buffer_load_b64(b.r, (int)off, 0, GS_LOAD_AUX);
  return U2{v.x, v.y};
}
GS_DEV uint32_t buf_load1(const BufRsrc &b, uint32_t off) { /* buffer_load_ubyte offen */
  return (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(b.r, (int)off, 0, GS_LOAD_AUX);
}
/* buffer_load_dword v, voff, s[rsrc], soff offen: the wave-uniform part of the address rides in
 * an SGPR, so a gather whose lanes differ only by a fixed per-lane origin needs NO vector ALU
 * for addressing (cascade corner loads). Default cache policy. */
GS_DEV uint32_t buf_gather4(const BufRsrc &b, uint32_t voff, uint32_t soff) {
  return __builtin_amdgcn_raw_buffer_load_b32(b.r, (int)voff, (int)soff, 0);
}
GS_DEV uint32_t uniform(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
/* one global_store_dwordx4 at a 16-byte aligned address */
GS_DEV void store_u32x4(void *p, const U4 &v) { *(gs_u32x4 *)p = gs_u32x4{v.x, v.y, v.z, v.w}; }
/* the same at any address (global memory takes a dwordx4 / dwordx2 at any byte address) */
typedef gs_u32x4 gs_u32x4_a1 __attribute__((aligned(1)));
GS_DEV void store_u32x4_any(void *p, const U4 &v) { *(gs_u32x4_a1 *)p = gs_u32x4{v.x, v.y, v.z, v.w}; }
GS_DEV U4 load_u32x4_any(const void *p) {
  const gs_u32x4 v = *(const gs_u32x4_a1 *)p;
  return U4{v.x, v.y, v.z, v.w};
}
GS_DEV void store_u32x2_any(void *p, uint32_t lo, uint32_t hi) {
  typedef unsigned int u32x2_a1 __attribute__((ext_vector_type(2), aligned(1)));
  *(u32x2_a1 *)p = u32x2_a1{lo, hi};
}
GS_DEV void buf_store16(const BufRsrc &b, uint32_t off, const U4 &v) { /* buffer_store_dwordx4 offen */
  __builtin_amdgcn_raw_buffer_store_b128(gs_u32x4{v.x, v.y, v.z, v.w}, b.r, (int)off, 0, GS_STORE_AUX);
}
/* default (write-back) policy: for stores that fill a cache line piecewise (a lane's 64 B of an
 * integral-image row go out as four 16-B stores interleaved with its neighbours') -- nt there
 * would push partial lines to memory */
GS_DEV void buf_store16_wb(const BufRsrc &b, uint32_t off, const U4 &v) {
  __builtin_amdgcn_raw_buffer_store_b128(gs_u32x4{v.x, v.y, v.z, v.w}, b.r, (int)off, 0, 0);
}
/* narrower stores (default policy: they complete lines other lanes fill) for the ragged end of a row */
typedef unsigned int gs_u32x2 __attribute__((ext_vector_type(2)));
GS_DEV void buf_store8(const BufRsrc &b, uint32_t off, uint32_t lo, uint32_t hi) { /* buffer_store_dwordx2 offen */
  __builtin_amdgcn_raw_buffer_store_b64(gs_u32x2{lo, hi}, b.r, (int)off, 0, 0);
}
GS_DEV void buf_store4(const BufRsrc &b, uint32_t off, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b32(v, b.r, (int)off, 0, 0); }
GS_DEV void buf_store2(const BufRsrc &b, uint32_t off, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v, b.r, (int)off, 0, 0); }
GS_DEV void buf_store1(const BufRsrc &b, uint32_t off, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b8((unsigned char)v, b.r, (int)off, 0, 0); }
GS_DEV uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); } /* v_alignbyte_b32 */
typedef unsigned short gs_u16x2 __attribute__((ext_vector_type(2)));
typedef short gs_i16x2 __attribute__((ext_vector_type(2)));
#define GS_U2(x) __builtin_bit_cast(gs_u16x2, (uint32_t)(x))
#define GS_I2(x) __builtin_bit_cast(gs_i16x2, (uint32_t)(x))
#define GS_R(x) __builtin_bit_cast(uint32_t, (x))
GS_DEV uint32_t pk_add_u16(uint32_t a, uint32_t b) { return GS_R((gs_u16x2)(GS_U2(a) + GS_U2(b))); }  /* v_pk_add_u16 */
GS_DEV uint32_t pk_sub_u16(uint32_t a, uint32_t b) { return GS_R((gs_u16x2)(GS_U2(a) - GS_U2(b))); }  /* v_pk_sub_u16 */
GS_DEV uint32_t pk_mul_u16(uint32_t a, uint32_t b) { return GS_R((gs_u16x2)(GS_U2(a) * GS_U2(b))); }  /* v_pk_mul_lo_u16 */
GS_DEV uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c) { return GS_R((gs_u16x2)(GS_U2(a) * GS_U2(b) + GS_U2(c))); } /* v_pk_mad_u16 */
/* 2*a + c per half in ONE v_pk_mad_u16 (hipcc strength-reduces a*2+c into shift+add, so spell it;
 * pure VALU register op: no memory, no hazard beyond what hipcc pads around asm) */
GS_DEV uint32_t pk_mad2_u16(uint32_t a, uint32_t c) {
#if GS_MAD2_OPAQUE
  /* the multiplier pair (2, 2) behind an s_mov the optimiser cannot see through (hoisted out of
   * loops, one SGPR): the v_pk_mad_u16 itself is then a compiler-visible instruction, so the
   * scheduler and the hazard recogniser treat it like any other packed op */
  uint32_t two;
  asm("s_mov_b32 %0, 0x00020002" : "=s"(two));
  return pk_mad_u16(a, two, c);
#else
  uint32_t d;
  asm("v_pk_mad_u16 %0, %1, 2, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(c));
  return d;
#endif
}
/* the instruction scheduler moves nothing across this point (keeps a prefetch where it was put) */
GS_DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
/* ask the scheduler for `n` instructions of class `mask` next (0x008 MFMA, 0x002 VALU, 0x100 LDS read): a pipeline's stages in order */
#define GS_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
/* the value, with its history hidden from the optimiser (no instruction): keeps a packed value packed when the
 * compiler would rather keep an unpacked copy alive */
GS_DEV uint32_t opaque(uint32_t x) {
  asm volatile("" : "+v"(x));
  return x;
}
/* acc = 2 acc + (a >= b), unsigned: v_cmp_ge_u32 (VOPC) + v_addc_co_u32 (VOP2), both full-rate encodings -- hipcc's own
 * compare + v_cndmask_b32_e64 + v_or3_b32 per bit of a code are VOP3 forms, which issue at 0.58 of that rate on gfx950
 * (scripts/ubench_valu.cpp).  No hazard: a VALU write of VCC may feed the next instruction's carry-in. */
GS_DEV uint32_t push_ge_u32(uint32_t acc, uint32_t a, uint32_t b) {
  asm("v_cmp_ge_u32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
  return acc;
}
GS_DEV uint32_t push_gt_u32(uint32_t acc, uint32_t a, uint32_t b) { /* acc = 2 acc + (a > b) */
  asm("v_cmp_gt_u32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
  return acc;
}
/* LDS atomic add (no return value) at byte address `lds_byte` of the block's LDS, tied into the
 * dependency chain of `through`: the instruction is issued after `through` has been produced and
 * before anything consumes it.  hipcc gives an atomic nobody waits for the lowest priority and sinks
 * all of a row's atomics to the end of the scheduling region, where the burst fills the LDS queue
 * and stalls the wave; threading them through values of the surrounding arithmetic spreads them
 * at no instruction cost.  The statement is invisible to hipcc's lgkmcnt bookkeeping: the caller
 * drains with lds_drain() before the block's barrier. */
template <class T> GS_DEV uint32_t lds_address(T *p) { /* byte address inside the block's LDS */
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)p;
}
/* -DGS_PLAIN_LDS_ATOMICS: the same adds as compiler-visible atomicAdd (hipcc's own lgkmcnt bookkeeping, its own
 * placement) -- the fallback should a hipcc update stop tolerating LDS traffic it cannot see; `make variant
 * TAG=plain_lds EXTRA=-DGS_PLAIN_LDS_ATOMICS` is one of the A/B builds (scripts/gpu_variants.sh) and must
 * produce the same bytes.  Contract of the threaded form: between the first lds_add_through and lds_drain()
 * the caller performs NO compiler-visible access to the table (k_fused.h: the only LDS object is `lh`, cleared
 * before a __syncthreads() ahead of the row loop, read after lds_drain() + __syncthreads() behind it). */
#ifdef GS_PLAIN_LDS_ATOMICS
GS_DEV void lds_add_through(unsigned *lds_base, uint32_t byte_off, uint32_t value, uint32_t &through) {
  (void)through;
  atomicAdd((unsigned *)((char *)lds_base + byte_off), value);
}
GS_DEV void lds_drain() {}
#else
GS_DEV void lds_add_through(unsigned *lds_base, uint32_t byte_off, uint32_t value, uint32_t &through) {
  asm volatile("ds_add_u32 %1, %2" : "+v"(through) : "v"(lds_address(lds_base) + byte_off), "v"(value) : "memory");
}
GS_DEV void lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#endif
/* a*b + c per half with a wave-uniform multiplier pair b (SGPR): kept as one v_pk_mad_u16 even
 * when b is a power of two */
GS_DEV uint32_t pk_mad_u16_s(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c));
  return d;
}
/* sum of the four u8 x u8 products + c (v_dot4_u32_u8) */
GS_DEV uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }
/* lo16(a) + hi16(a) + c: one v_dot2_u32_u16 */
GS_DEV uint32_t udot2_ones(uint32_t a, uint32_t c) {
  typedef unsigned short gs_u16x2 __attribute__((ext_vector_type(2)));
  return __builtin_amdgcn_udot2(__builtin_bit_cast(gs_u16x2, a), gs_u16x2{1, 1}, c, false);
}
/* dword at any byte address (global memory handles unaligned dwords in hardware) */
GS_DEV uint32_t load_u32_unaligned(const uint8_t *p) {
  typedef uint32_t u32_unaligned __attribute__((aligned(1)));
  return *(const u32_unaligned *)p;
}
/* (low | high half of a) * (low half of the wave-uniform b) + c as a full 32-bit result: one
 * v_mad_u32_u16 with op_sel picking the half -- no separate extraction of the u16 */
GS_DEV uint32_t mad_u32_u16_lo(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[0,0,0,0]" : "=v"(d) : "v"(a), "s"(b), "v"(c));
  return d;
}
GS_DEV uint32_t mad_u32_u16_hi(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(a), "s"(b), "v"(c));
  return d;
}
GS_DEV uint32_t pk_shl_u16(uint32_t a, unsigned s) { return GS_R((gs_u16x2)(GS_U2(a) << (unsigned short)s)); } /* v_pk_lshlrev_b16 */
GS_DEV uint32_t pk_shr_u16(uint32_t a, unsigned s) { return GS_R((gs_u16x2)(GS_U2(a) >> (unsigned short)s)); } /* v_pk_lshrrev_b16 */
GS_DEV uint32_t pk_min_u16(uint32_t a, uint32_t b) { return GS_R(__builtin_elementwise_min(GS_U2(a), GS_U2(b))); } /* v_pk_min_u16 */
GS_DEV uint32_t pk_max_u16(uint32_t a, uint32_t b) { return GS_R(__builtin_elementwise_max(GS_U2(a), GS_U2(b))); } /* v_pk_max_u16 */
GS_DEV uint32_t pk_subsat_u16(uint32_t a, uint32_t b) { return GS_R(__builtin_elementwise_sub_sat(GS_U2(a), GS_U2(b))); } /* v_pk_sub_u16 clamp */
/* min(a * 128, 65535) per half: v_pk_mad_u16 ... clamp (unsigned saturation of the full a*b+c).
 * For a < 2048 the high byte of each half is min(a >> 1, 255) -- shift, clamp in one lane-op. */
GS_DEV uint32_t pk_shl7_sat_u16(uint32_t a) {
  uint32_t d;
  asm("v_pk_mad_u16 %0, %1, %2, 0 clamp" : "=v"(d) : "v"(a), "s"(0x00800080u));
  return d;
}
GS_DEV uint32_t pk_abs_i16(uint32_t a) { return GS_R(__builtin_elementwise_abs(GS_I2(a))); }           /* v_pk_sub_i16 + v_pk_max_i16 */
GS_DEV uint32_t pk_max_i16(uint32_t a, uint32_t b) { return GS_R(__builtin_elementwise_max(GS_I2(a), GS_I2(b))); } /* v_pk_max_i16 */
GS_DEV uint32_t pk_sar_i16(uint32_t a, unsigned s) { return GS_R((gs_i16x2)(GS_I2(a) >> (short)s)); }            /* v_pk_ashrrev_i16 */
/* v_mfma_i32_32x32x32_i8 (see the emulation above for the operand slots) */
GS_DEV void mfma_i32_32x32x32_i8(const U4 &a, const U4 &b, int32_t (&c)[16]) {
  typedef int gs_i32x4 __attribute__((ext_vector_type(4)));
  typedef int gs_i32x16 __attribute__((ext_vector_type(16)));
  gs_i32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = c[r];
  acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(gs_i32x4{(int)a.x, (int)a.y, (int)a.z, (int)a.w},
                                              gs_i32x4{(int)b.x, (int)b.y, (int)b.z, (int)b.w}, acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 16; r++) c[r] = acc[r];
}
#endif

/* ------------------------------------------------------------------ common */
/* Two u16 fields per dword whose sums cannot carry out of a field (unsigned pixel sums) are added
 * with PLAIN 32-bit adds: on gfx950 v_add_u32 / v_sub_u32 issue at the full rate (~900 G
 * wave-instructions/s chip-wide) and a three-input sum is one v_add3_u32, while every packed-16
 * instruction (v_pk_add_u16 ...) issues at ~577 G -- measured by scripts/ubench_valu.cpp,
 * profiles/r02a_ubench_valu.log.  sub2 needs every field of a >= the same field of b. */
#ifndef GS_PLAIN_ADDS
#define GS_PLAIN_ADDS 1 /* 0: the packed-16 spelling, kept for A/B builds (make variant EXTRA=-DGS_PLAIN_ADDS=0) */
#endif
#if GS_PLAIN_ADDS
GS_DEV uint32_t add2(uint32_t a, uint32_t b) { return a + b; }
GS_DEV uint32_t add2(uint32_t a, uint32_t b, uint32_t c) { return a + b + c; }
GS_DEV uint32_t sub2(uint32_t a, uint32_t b) { return a - b; }
#else
GS_DEV uint32_t add2(uint32_t a, uint32_t b) { return pk_add_u16(a, b); }
GS_DEV uint32_t add2(uint32_t a, uint32_t b, uint32_t c) { return pk_add_u16(pk_add_u16(a, b), c); }
GS_DEV uint32_t sub2(uint32_t a, uint32_t b) { return pk_sub_u16(a, b); }
#endif
/* |a - b| per field for unsigned fields: max - min (the difference cannot borrow) */
GS_DEV uint32_t absdiff2(uint32_t a, uint32_t b) { return sub2(pk_max_u16(a, b), pk_min_u16(a, b)); }
GS_DEV unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }
GS_DEV unsigned umax(unsigned a, unsigned b) { return a > b ? a : b; }
/* low 32 bits of the product of the operands' low 24 bits: v_mul_u32_u24, a full-rate instruction (v_mul_lo_u32 runs at a quarter) */
GS_DEV uint32_t mul_u24(uint32_t a, uint32_t b) {
#ifdef GS_EMU
  return (a & 0xffffffu) * (b & 0xffffffu);
#else
  return __ockl_mul24_u32(a, b); /* the intrinsic: a masked C product can be merged with a plain one and end up as v_mul_lo_u32 */
#endif
}
/* |a - b| for a, b < 65536 (v_sad_u16 with zero high halves on the GPU) */
GS_DEV unsigned absdiff_u16(unsigned a, unsigned b) {
#ifdef GS_EMU
  return a > b ? a - b : b - a;
#else
  return __builtin_amdgcn_sad_u16(a, b, 0u);
#endif
}
/* lane (4 * (lane / 4) + S[lane % 4])'s value: v_mov_b32_dpp quad_perm:[S0,S1,S2,S3] (no LDS crossbar).  Every lane
 * of a quad has to execute it together (on the GPU a disabled source lane yields 0; the emulator's exchange is a
 * rendezvous of the quad's live lanes). */
template <int S0, int S1, int S2, int S3>
GS_DEV uint32_t quad_perm(uint32_t x) {
#ifdef GS_EMU
  const int sel[4] = {S0, S1, S2, S3};
  return (uint32_t)emu::quad_exchange(x, (unsigned)sel[lane_id() & 3u]);
#else
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, S0 | (S1 << 2) | (S2 << 4) | (S3 << 6), 0xf, 0xf, false);
#endif
}
GS_DEV uint32_t readlane0(uint32_t x) { return shfl(x, 0); }

/* inclusive add-scan across the wave.  gfx950: six DPP adds -- row_shr 1/2/4/8 inside each row of
 * 16 lanes, then row_bcast15 / row_bcast31 carry the row totals across (no LDS crossbar);
 * emulator: Hillis-Steele over shuffles. */
GS_DEV uint32_t wave_incl_scan(uint32_t v) {
#ifdef GS_EMU
  unsigned l = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = shfl(v, (int)l - d);
    if ((int)l >= d) v += t;
  }
  return v;
#else
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); /* row_shr:1 */
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); /* row_shr:2 */
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); /* row_shr:4 */
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); /* row_shr:8 */
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); /* row_bcast:15 -> rows 1, 3 */
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); /* row_bcast:31 -> rows 2, 3 */
  return v;
#endif
}
/* lane k's value for a wave-uniform k (v_readlane_b32 with an SGPR lane select) */
GS_DEV uint32_t readlane_at(uint32_t x, unsigned k) {
#ifdef GS_EMU
  return shfl(x, (int)k);
#else
  return (uint32_t)__builtin_amdgcn_readlane((int)x, (int)k);
#endif
}
/* lane 63's value, wave-uniform (v_readlane_b32 -> SGPR on the GPU) */
GS_DEV uint32_t readlane_last(uint32_t x) {
#ifdef GS_EMU
  return shfl(x, 63);
#else
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
#endif
}
/* Orders this wave's LDS accesses across LANES: what any lane stored before it is visible to every lane's loads after
 * it.  The hardware executes a wave's DS instructions in order, so this only has to stop the compiler from moving
 * memory operations across it (s_waitcnt-free: v_nop-sized); the emulator's lanes are fibers and really meet here. */
GS_DEV void wave_sync() {
#ifdef GS_EMU
  (void)emu::wave_exchange(0, [](const uint64_t *, const bool *) { return 0; });
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}
GS_DEV uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += shfl(v, (int)(lane_id() ^ (unsigned)d));
  return v;
}
GS_DEV int wave_sum_i(int v) { return (int)wave_sum((uint32_t)v); }

/* 16 bytes shifted down by nb = 1..15 bytes, zeros entering at the top (nb wave-uniform: the dword part of the shift
 * is a scalar choice, the byte part one v_alignbyte_b32 per dword) */
GS_DEV U4 shift_down_bytes(U4 v, unsigned nb) {
  /* the dword part of the shift is a wave-uniform SWITCH (a chain of selects on nb / 4 ends up as a table in scratch memory
   * with hipcc, 64-bit shifts run at a quarter of the rate), the byte part one v_perm_b32 per dword with a selector that
   * reads bytes b .. b + 3 of the pair {hi, lo} */
  uint32_t e0, e1, e2, e3;
  switch (nb >> 2) {
    case 0: e0 = v.x, e1 = v.y, e2 = v.z, e3 = v.w; break;
    case 1: e0 = v.y, e1 = v.z, e2 = v.w, e3 = 0u; break;
    case 2: e0 = v.z, e1 = v.w, e2 = 0u, e3 = 0u; break;
    default: e0 = v.w, e1 = 0u, e2 = 0u, e3 = 0u; break;
  }
  const uint32_t sel = 0x03020100u + 0x01010101u * (nb & 3u);
  return U4{perm_b32(e1, e0, sel), perm_b32(e2, e1, sel), perm_b32(e3, e2, sel), perm_b32(0u, e3, sel)};
}
/* the first m = 1..15 bytes of o to buffer offset off (m wave-uniform): 8 + 4 + 2 + 1 byte stores as m's bits say.
 * Lanes that must not store pass off = kOOB. */
GS_DEV void buf_store_first(const BufRsrc &b, uint32_t off, uint32_t ox, uint32_t oy, uint32_t oz, uint32_t ow, unsigned m) {
  /* scalars by value: with a struct behind a reference hipcc turns "the low or the high half" into an indexed load from a
   * scratch copy */
  const bool on = off != kOOB;
  uint64_t rest = ox | ((uint64_t)oy << 32);
  if (m & 8u) buf_store8(b, off, ox, oy), rest = oz | ((uint64_t)ow << 32);
  if (m & 4u) buf_store4(b, on ? off + (m & 8u) : kOOB, (uint32_t)rest), rest >>= 32;
  if (m & 2u) buf_store2(b, on ? off + (m & 12u) : kOOB, (uint32_t)rest), rest >>= 16;
  if (m & 1u) buf_store1(b, on ? off + (m & 14u) : kOOB, (uint32_t)rest);
}

/* bytes {b0,b1,b2,b3} of a dword -> two dwords of u16 pairs: lo=(b0,b1) hi=(b2,b3) */
GS_DEV uint32_t unpack_lo(uint32_t d) { return perm_b32(0, d, 0x0c010c00u); }
GS_DEV uint32_t unpack_hi(uint32_t d) { return perm_b32(0, d, 0x0c030c02u); }
/* inverse: low bytes of the u16 pairs (lo=(p0,p1), hi=(p2,p3)) -> one dword */
GS_DEV uint32_t pack_lohi(uint32_t lo, uint32_t hi) { return perm_b32(hi, lo, 0x06040200u); }
/* same for the HIGH bytes of the u16 pairs */
GS_DEV uint32_t pack_lohi_b1(uint32_t lo, uint32_t hi) { return perm_b32(hi, lo, 0x07050301u); }

}  // namespace gs
#endif
