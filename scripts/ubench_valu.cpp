// VALU issue-rate microbenchmark for gfx950 (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench_valu.cpp -o build_variants/ubench_valu
// For each instruction: 8 independent dependency chains per wave, W waves per SIMD on every SIMD
// of the chip; reports wave-instructions per second (chip) and shader cycles per wave-instruction
// per SIMD (s_memtime inside the kernel, first wave of block 0).  The packed-16 / byte-permute
// numbers against the plain 32-bit ones decide how the fused blur+sobel kernel spells its sums
// (VERDICT r01 item 4).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

#define R8(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define STR(x) #x
#define XSTR(x) STR(x)

// dst = chain register %i; the other sources are %8 (b) and %9 (c)
#define DEFK(name, LINE)                                                                        \
  __global__ __launch_bounds__(256) void name(uint32_t *out, int iters, unsigned long long *cyc) { \
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,  \
             a6 = a0 + 6, a7 = a0 + 7;                                                          \
    uint32_t b = threadIdx.x * 3u + 1u, c = blockIdx.x + 5u;                                    \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                 \
    for (int i = 0; i < iters; i++) {                                                           \
      asm volatile(R8(LINE) R8(LINE) R8(LINE) R8(LINE)                                          \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),      \
                     "+v"(a7)                                                                   \
                   : "v"(b), "v"(c)                                                             \
                   : "vcc");                                                                    \
    }                                                                                           \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                 \
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;                                  \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;         \
  }

#define L2(ins, i) ins " %" #i ", %" #i ", %8\n"
#define L3(ins, i) ins " %" #i ", %" #i ", %8, %9\n"

#define K2(name, ins)            \
  DEFK_2(name, ins)
// helper macros: the line template needs the chain index, so spell each kernel's LINE macro
#define MK2(name, ins)                                   \
  static const char *name##_ins = ins;                   \
  DEFK(name, name##_LINE)

#define add_u32_LINE(i) L2("v_add_u32", i)
#define sub_u32_LINE(i) L2("v_sub_u32", i)
#define and_b32_LINE(i) L2("v_and_b32", i)
#define xor_b32_LINE(i) L2("v_xor_b32", i)
#define lshlrev_b32_LINE(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define add3_u32_LINE(i) L3("v_add3_u32", i)
#define lshl_add_u32_LINE(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %8\n"
#define add_lshl_u32_LINE(i) "v_add_lshl_u32 %" #i ", %" #i ", %8, 1\n"
#define and_or_b32_LINE(i) L3("v_and_or_b32", i)
#define or3_b32_LINE(i) L3("v_or3_b32", i)
#define lshl_or_b32_LINE(i) "v_lshl_or_b32 %" #i ", %" #i ", 8, %8\n"
#define xad_u32_LINE(i) L3("v_xad_u32", i)
#define bfe_u32_LINE(i) "v_bfe_u32 %" #i ", %" #i ", 8, 8\n"
#define perm_b32_LINE(i) L3("v_perm_b32", i)
#define alignbit_b32_LINE(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 16\n"
#define alignbyte_b32_LINE(i) "v_alignbyte_b32 %" #i ", %" #i ", %8, 2\n"
#define mul_u32_u24_LINE(i) L2("v_mul_u32_u24", i)
#define mad_u32_u24_LINE(i) L3("v_mad_u32_u24", i)
#define mul_lo_u32_LINE(i) L2("v_mul_lo_u32", i)
#define mul_hi_u32_LINE(i) L2("v_mul_hi_u32", i)
#define mad_u32_u16_LINE(i) L3("v_mad_u32_u16", i)
#define sad_u8_LINE(i) L3("v_sad_u8", i)
#define sad_u16_LINE(i) L3("v_sad_u16", i)
#define msad_u8_LINE(i) L3("v_msad_u8", i)
#define dot4_u32_u8_LINE(i) L3("v_dot4_u32_u8", i)
#define dot2_u32_u16_LINE(i) L3("v_dot2_u32_u16", i)
#define min3_u32_LINE(i) L3("v_min3_u32", i)
#define max3_u32_LINE(i) L3("v_max3_u32", i)
#define med3_u32_LINE(i) L3("v_med3_u32", i)
#define max_u32_LINE(i) L2("v_max_u32", i)
#define cndmask_b32_LINE(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define mov_dpp_shr1_LINE(i) "v_mov_b32_dpp %" #i ", %" #i " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define add_dpp_rowshr1_LINE(i) "v_add_u32_dpp %" #i ", %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define add_u32_sdwa_LINE(i) \
  "v_add_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define add_u16_LINE(i) L2("v_add_u16", i)
#define sub_u16_LINE(i) L2("v_sub_u16", i)
#define max_u16_LINE(i) L2("v_max_u16", i)
#define mad_u16_LINE(i) L3("v_mad_u16", i)
#define pk_add_u16_LINE(i) L2("v_pk_add_u16", i)
#define pk_sub_u16_LINE(i) L2("v_pk_sub_u16", i)
#define pk_sub_i16_LINE(i) L2("v_pk_sub_i16", i)
#define pk_max_i16_LINE(i) L2("v_pk_max_i16", i)
#define pk_min_u16_LINE(i) L2("v_pk_min_u16", i)
#define pk_mul_lo_u16_LINE(i) L2("v_pk_mul_lo_u16", i)
#define pk_mad_u16_LINE(i) L3("v_pk_mad_u16", i)
#define pk_mad_u16_clamp_LINE(i) "v_pk_mad_u16 %" #i ", %" #i ", %8, %9 clamp\n"
#define pk_lshrrev_b16_LINE(i) "v_pk_lshrrev_b16 %" #i ", 1, %" #i "\n"
#define pk_ashrrev_i16_LINE(i) "v_pk_ashrrev_i16 %" #i ", 1, %" #i "\n"
#define add_f32_LINE(i) L2("v_add_f32", i)
#define fma_f32_LINE(i) L3("v_fma_f32", i)
#define pk_add_f32_LINE(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n" /* 64-bit operands: see below */
#define mbcnt_LINE(i) "v_mbcnt_lo_u32_b32 %" #i ", %8, %" #i "\n"
#define cvt_pk_u8_f32_LINE(i) L3("v_cvt_pk_u8_f32", i)
#define lerp_u8_LINE(i) L3("v_lerp_u8", i)
#define bfi_b32_LINE(i) L3("v_bfi_b32", i)
#define sub_co_LINE(i) "v_sub_co_u32 %" #i ", vcc, %" #i ", %8\n"
#define cmp_cnd_LINE(i) "v_cmp_lt_u32 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"

/* round 2 (gs_fast / histogram questions): byte -> float conversion, float class flags, carries */
#define cvt_f32_ubyte0_LINE(i) "v_cvt_f32_ubyte0 %" #i ", %" #i "\n"
#define cvt_f32_ubyte2_LINE(i) "v_cvt_f32_ubyte2 %" #i ", %" #i "\n"
#define cvt_u32_f32_LINE(i) "v_cvt_u32_f32 %" #i ", %" #i "\n"
#define sub_f32_clamp_LINE(i) "v_sub_f32_e64 %" #i ", %" #i ", %8 clamp\n"
#define min_f32_LINE(i) L2("v_min_f32", i)
#define min_f32_abs_LINE(i) "v_min_f32_e64 %" #i ", %" #i ", |%8|\n"
#define fmac_f32_LINE(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define addc_co_LINE(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %" #i ", vcc\n"
#define cmp_only_LINE(i) "v_cmp_gt_u32 vcc, %" #i ", %8\n"
#define cmp_addc_LINE(i) "v_cmp_gt_u32 vcc, %8, %9\nv_addc_co_u32 %" #i ", vcc, %" #i ", %" #i ", vcc\n"
#define min_u16_LINE(i) L2("v_min_u16", i)
#define max_i16_LINE(i) L2("v_max_i16", i)
#define min_i32_LINE(i) L2("v_min_i32", i)
#define or_b32_LINE(i) L2("v_or_b32", i)
#define lshrrev_b32_LINE(i) "v_lshrrev_b32 %" #i ", 1, %" #i "\n"
#define sub_u32_sdwa_b0_LINE(i) \
  "v_sub_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n"
#define NEW(X)                                                                                   \
  X(cvt_f32_ubyte0) X(cvt_f32_ubyte2) X(cvt_u32_f32) X(sub_f32_clamp) X(min_f32) X(min_f32_abs)   \
  X(fmac_f32) X(addc_co) X(cmp_only) X(cmp_addc) X(min_u16) X(max_i16) X(min_i32) X(or_b32)       \
  X(lshrrev_b32) X(sub_u32_sdwa_b0)

#define ALL(X)                                                                                    \
  X(add_u32) X(sub_u32) X(and_b32) X(xor_b32) X(lshlrev_b32) X(add3_u32) X(lshl_add_u32)          \
  X(add_lshl_u32) X(and_or_b32) X(or3_b32) X(lshl_or_b32) X(xad_u32) X(bfe_u32) X(bfi_b32)        \
  X(perm_b32) X(alignbit_b32) X(alignbyte_b32) X(mul_u32_u24) X(mad_u32_u24) X(mul_lo_u32)        \
  X(mul_hi_u32) X(mad_u32_u16) X(sad_u8) X(sad_u16) X(msad_u8) X(dot4_u32_u8) X(dot2_u32_u16)     \
  X(min3_u32) X(max3_u32) X(med3_u32) X(max_u32) X(cndmask_b32) X(mov_dpp_shr1)                   \
  X(add_dpp_rowshr1) X(add_u32_sdwa) X(add_u16) X(sub_u16) X(max_u16) X(mad_u16) X(pk_add_u16)    \
  X(pk_sub_u16) X(pk_sub_i16) X(pk_max_i16) X(pk_min_u16) X(pk_mul_lo_u16) X(pk_mad_u16)          \
  X(pk_mad_u16_clamp) X(pk_lshrrev_b16) X(pk_ashrrev_i16) X(add_f32) X(fma_f32) X(mbcnt)          \
  X(cvt_pk_u8_f32) X(lerp_u8) X(sub_co) X(cmp_cnd)

#define X(n) DEFK(k_##n, n##_LINE)
ALL(X)
NEW(X)
#undef X

// v_cndmask_b32 with the mask in VCC written by a VALU compare before the loop / in an SGPR pair
// written by SALU / freshly written by a v_cmp each time (the cmp_cnd row above)
__global__ __launch_bounds__(256) void k_cnd_vcc_once(uint32_t *out, int iters, unsigned long long *cyc) {
  uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  uint32_t b = threadIdx.x * 3u + 1u, c = blockIdx.x + 5u;
  asm volatile("v_cmp_lt_u32 vcc, %0, %1\ns_nop 4\n" ::"v"(b), "v"(c) : "vcc");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    asm volatile(R8(cndmask_b32_LINE) R8(cndmask_b32_LINE) R8(cndmask_b32_LINE) R8(cndmask_b32_LINE)
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : "v"(b), "v"(c));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
#define cnd_sgpr_LINE(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, %10\n"
__global__ __launch_bounds__(256) void k_cnd_sgpr(uint32_t *out, int iters, unsigned long long *cyc) {
  uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  uint32_t b = threadIdx.x * 3u + 1u, c = blockIdx.x + 5u;
  unsigned long long m = 0x5555aaaa3333ccccull ^ (unsigned long long)blockIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    asm volatile(R8(cnd_sgpr_LINE) R8(cnd_sgpr_LINE) R8(cnd_sgpr_LINE) R8(cnd_sgpr_LINE)
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : "v"(b), "v"(c), "s"(m));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
// the same select spelled with bit operations on a per-lane all-ones / all-zeros mask register
#define bfi_sel_LINE(i) "v_bfi_b32 %" #i ", %9, %8, %" #i "\n"
DEFK(k_bfi_select, bfi_sel_LINE)

// LDS atomic rate (histogram): ds_add_u32 without return, address pattern = lane-private copies
__global__ __launch_bounds__(256) void k_ds_add(uint32_t *out, int iters, unsigned long long *cyc, int mode) {
  __shared__ unsigned lh[256 * 32];
  for (unsigned i = threadIdx.x; i < 256 * 32; i += 256) lh[i] = 0;
  __syncthreads();
  unsigned v = threadIdx.x * 2654435761u + blockIdx.x;
  const unsigned copy = threadIdx.x & 31u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 32; k++) {
      v = v * 1664525u + 1013904223u;
      unsigned bin = mode == 0 ? (v >> 24) : mode == 1 ? (k & 255u) : ((v >> 24) & 15u);
      atomicAdd(&lh[bin * 32u + copy], 1u);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = lh[threadIdx.x] + v;
}

// LDS atomic rate without any VALU in between: 16 precomputed addresses per lane (conflict-free copies),
// `active` lanes of every wave take part (the rest are masked off by EXEC)
__global__ __launch_bounds__(256) void k_ds_add_pure(uint32_t *out, int iters, unsigned long long *cyc, int active, int samebank) {
  __shared__ unsigned lh[256 * 32];
  for (unsigned i = threadIdx.x; i < 256 * 32; i += 256) lh[i] = 0;
  __syncthreads();
  unsigned v = threadIdx.x * 2654435761u + blockIdx.x;
  const unsigned copy = samebank ? 0u : (threadIdx.x & 31u);
  uint32_t ad[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    v = v * 1664525u + 1013904223u;
    ad[k] = (uint32_t)(size_t)&lh[(v >> 24) * 32u + copy];
  }
  const uint32_t one = 1;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if ((int)(threadIdx.x & 63u) < active) {
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int k = 0; k < 16; k++) asm volatile("ds_add_u32 %0, %1" ::"v"(ad[k]), "v"(one) : "memory");
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = lh[threadIdx.x] + v;
}

struct Entry {
  const char *name;
  void (*fn)(uint32_t *, int, unsigned long long *);
};

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000; /* ~3-9 ms per launch: long enough for the clock to settle */
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("# %s, %d CUs, clockRate %d kHz; 32 wave-instructions per loop iteration, %d iterations\n", prop.name, cus,
         prop.clockRate, iters);
  uint32_t *out;
  unsigned long long *cyc;
  CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
  CK(hipMalloc(&cyc, 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
#define X(n) {#n, k_##n},
  const Entry tab_all[] = {ALL(X){"cndmask_vcc_set_once", k_cnd_vcc_once}, {"cndmask_e64_sgprmask", k_cnd_sgpr}, {"bfi_select", k_bfi_select}};
  const Entry tab_new[] = {NEW(X)};
#undef X
  const bool only_new = argc > 2 && strcmp(argv[2], "new") == 0; /* ubench_valu <iters> new: the round-2 additions + LDS atomics */
  std::vector<Entry> tab;
  if (!only_new) tab.assign(tab_all, tab_all + sizeof(tab_all) / sizeof(tab_all[0]));
  tab.insert(tab.end(), tab_new, tab_new + sizeof(tab_new) / sizeof(tab_new[0]));
  printf("%-18s %6s %14s %14s %12s\n", "instruction", "w/SIMD", "Gwave-inst/s", "cyc/inst/SIMD", "ns/launch");
  for (const Entry &e : tab) {
    for (int wps : {1, 2, 4, 8}) {
      const int blocks = cus * wps; /* 256-thread block = 4 waves = one per SIMD */
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, iters, cyc); /* warm-up at full length */
      CK(hipDeviceSynchronize());
      float best = 1e30f;
      unsigned long long c = 0;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) {
          best = ms;
          CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        }
      }
      const double ninst = (double)blocks * 4 * iters * 32.0 * ((strcmp(e.name, "cmp_cnd") == 0 || strcmp(e.name, "cmp_addc") == 0) ? 2 : 1);
      /* cycles per wave-instruction per SIMD = wave cycles / (instructions of that wave * waves sharing the SIMD) */
      const double cpi = (double)c / (iters * 32.0 * ((strcmp(e.name, "cmp_cnd") == 0 || strcmp(e.name, "cmp_addc") == 0) ? 2 : 1)) / wps;
      printf("%-18s %6d %14.1f %14.2f %12.0f\n", e.name, wps, ninst / best / 1e6, cpi, best * 1e6);
    }
  }
  for (int mode = 0; mode < 3; mode++)
    for (int wps : {1, 2, 4, 8}) {
      const int blocks = cus * wps;
      const int it = iters / 8;
      hipLaunchKernelGGL(k_ds_add, dim3(blocks), dim3(256), 0, 0, out, 2, cyc, mode);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_ds_add, dim3(blocks), dim3(256), 0, 0, out, it, cyc, mode);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double ninst = (double)blocks * 4 * it * 32.0;
      printf("ds_add_u32 mode %d (%s) %d w/SIMD: %.1f Gwave-inst/s = %.2f Tlane-atomics/s (each iteration also 2 VALU for the LCG)\n",
             mode, mode == 0 ? "random bins" : mode == 1 ? "same bin per wave, own copy" : "16 bins", wps, ninst / ms / 1e6,
             ninst * 64 / ms / 1e9);
    }
  for (int samebank = 0; samebank < 2; samebank++)
    for (int active : {64, 32, 16, 8})
      for (int wps : {2, 4}) {
        const int blocks = cus * wps;
        const int it = iters / 8;
        hipLaunchKernelGGL(k_ds_add_pure, dim3(blocks), dim3(256), 0, 0, out, 2, cyc, active, samebank);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_ds_add_pure, dim3(blocks), dim3(256), 0, 0, out, it, cyc, active, samebank);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double ninst = (double)blocks * 4 * it * 32.0;
        printf("ds_add_u32 pure stream, %2d active lanes, %s, %d w/SIMD: %.1f Gwave-inst/s = %.2f Tlane-atomics/s = %.2f LDS cycles per wave-inst per CU\n",
               active, samebank ? "every lane in copy 0 (bank conflicts by value)" : "lane-private copies", wps, ninst / ms / 1e6,
               ninst * active / ms / 1e9, (double)cus * prop.clockRate * 1e3 / (ninst / (ms * 1e-3)));
      }
  return 0;
}
