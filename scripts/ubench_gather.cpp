// Cost of one wave-wide global load from an L2-resident table on gfx950, by width and alignment (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench_gather.cpp -o build_variants/ubench_gather
// The LBP cascade's dense phase reads 16 integral-image corners per weak classifier with one dword per lane
// (64 consecutive windows = 256 contiguous bytes per wave-load).  Would 2 or 4 windows per lane (dwordx2 / dwordx4 at
// any dword alignment) move more bytes per texture-addresser cycle?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef u32x2 u32x2_u __attribute__((aligned(4)));
typedef u32x4 u32x4_u __attribute__((aligned(4)));

// W = dwords per lane (1, 2, 4); lane l reads W consecutive dwords at base + (l * W * stride + mis) dwords;
// 16 loads per iteration at 16 row offsets (like the 4 x 4 corner grid), rows of `pitch` dwords
template <int W>
__global__ __launch_bounds__(256) void k_gather(const unsigned *tab, unsigned ndw, unsigned pitch, unsigned mis, unsigned lane_stride,
                                                int iters, unsigned *out) {
  const unsigned lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
  unsigned acc = 0;
  const unsigned lanepart = lane * W * lane_stride + mis;
  unsigned base = (wave * 4099u) & (ndw / 2 - 1u);
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const unsigned off = (base + (unsigned)k * pitch + lanepart) & (ndw - 1u); /* ndw is a power of two; the table has 8 spare dwords */
      if (W == 1) acc += tab[off];
      else if (W == 2) { const u32x2 v = *(const u32x2_u *)(tab + off); acc += v.x ^ v.y; }
      else { const u32x4 v = *(const u32x4_u *)(tab + off); acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    base = (base + 977u * 64u) & (ndw / 2 - 1u);
  }
  out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int W> void run(const char *what, const unsigned *tab, unsigned ndw, unsigned mis, unsigned lane_stride, unsigned *out, int cus, int mhz) {
  const int blocks = cus * 8, iters = 2000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_gather<W>, dim3(blocks), dim3(256), 0, 0, tab, ndw, 1921u, mis, lane_stride, 50, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_gather<W>, dim3(blocks), dim3(256), 0, 0, tab, ndw, 1921u, mis, lane_stride, iters, out);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double loads = (double)blocks * 4 * iters * 16;
  const double cyc = (double)cus * mhz * 1e6 * (ms * 1e-3) / loads; /* CU-cycles per wave-load */
  printf("%-58s %8.3f ms  %6.1f G wave-loads/s  %6.2f cycles per wave-load per CU  %6.1f B/clk/CU  (%.2f cycles per 64 dwords)\n", what, ms,
         loads / ms / 1e6, cyc, 256.0 * W / cyc, cyc / W);
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, mhz = prop.clockRate / 1000;
  const unsigned ndw = 2u << 20; /* 8 MB: L2 / MALL resident */
  unsigned *tab, *out;
  CK(hipMalloc(&tab, (size_t)ndw * 4 + 64)); CK(hipMemset(tab, 1, (size_t)ndw * 4 + 64));
  CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
  printf("# %s, %d CUs @ %d MHz; 16 loads per iteration at row offsets k * 1921 dwords, 8 blocks of 4 waves per CU\n", prop.name, cus, mhz);
  run<1>("dword, consecutive lanes (256 B per wave-load)", tab, ndw, 0, 1, out, cus, mhz);
  run<1>("dword, consecutive lanes, base + 1 dword", tab, ndw, 1, 1, out, cus, mhz);
  run<2>("dwordx2, consecutive lanes (512 B), 8 B aligned", tab, ndw, 0, 1, out, cus, mhz);
  run<2>("dwordx2, consecutive lanes, base + 1 dword", tab, ndw, 1, 1, out, cus, mhz);
  run<4>("dwordx4, consecutive lanes (1 KB), 16 B aligned", tab, ndw, 0, 1, out, cus, mhz);
  run<4>("dwordx4, consecutive lanes, base + 1 dword", tab, ndw, 1, 1, out, cus, mhz);
  run<4>("dwordx4, consecutive lanes, base + 2 dwords", tab, ndw, 2, 1, out, cus, mhz);
  run<4>("dwordx4, consecutive lanes, base + 3 dwords", tab, ndw, 3, 1, out, cus, mhz);
  run<1>("dword, lane stride 2 dwords (windows at step 2)", tab, ndw, 0, 2, out, cus, mhz);
  run<1>("dword, lane stride 3 dwords", tab, ndw, 0, 3, out, cus, mhz);
  return 0;
}
