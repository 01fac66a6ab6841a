// issue rate of v_mfma_i32_32x32x32_i8: NACC independent accumulator chains per wave, WAVES waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(int *out, int iters, v4i a0, v4i b0) {
  v16i acc[NACC];
#pragma unroll
  for (int n = 0; n < NACC; n++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[n][r] = 0;
  v4i a = a0, b = b0;
  a.x += threadIdx.x;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int n = 0; n < NACC; n++) acc[n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[n], 0, 0, 0);
  }
  int s = 0;
#pragma unroll
  for (int n = 0; n < NACC; n++)
#pragma unroll
    for (int r = 0; r < 16; r++) s += acc[n][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int blocks, const char *what) {
  int *out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  v4i a = {1, 2, 3, 4}, b = {5, 6, 7, 8};
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, 10, a, b);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, a, b);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // each block = 4 waves = 1 per SIMD of a CU; blocks per CU = blocks / 256
  const double mfma_per_simd = (double)iters * NACC * (blocks / 256.0);
  printf("%-28s chains %d, waves/SIMD %d: %.3f ms, %.1f ns per MFMA per SIMD = %.1f cycles at 2.4 GHz, %.0f TOPS\n", what, NACC, blocks / 256, ms,
         ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4, (double)iters * NACC * blocks * 4 * 65536.0 / (ms * 1e-3) / 1e12);
  hipFree(out);
}
int main() {
  run<1>(256, "1 wave/SIMD"); run<2>(256, "1 wave/SIMD"); run<4>(256, "1 wave/SIMD");
  run<1>(512, "2 waves/SIMD"); run<2>(512, "2 waves/SIMD"); run<4>(512, "2 waves/SIMD");
  run<2>(1024, "4 waves/SIMD");
  return 0;
}
