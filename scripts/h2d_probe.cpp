// H2D / D2H rate of a 512 MiB buffer vs how the host side was allocated (MI355X box probe)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t n = 512ull << 20;
  void *d; hipMalloc(&d, n);
  struct { const char *name; unsigned flags; int kind; } v[] = {
    {"malloc (pageable)", 0, 0}, {"hipHostMalloc default", 0, 1}, {"hipHostMalloc NonCoherent", hipHostMallocNonCoherent, 1},
    {"hipHostMalloc WriteCombined", hipHostMallocWriteCombined, 1}, {"hipHostMalloc Coherent", hipHostMallocCoherent, 1},
    {"malloc + hipHostRegister", 0, 2}};
  for (auto &x : v) {
    void *h = nullptr; double t0 = now();
    if (x.kind == 1) { if (hipHostMalloc(&h, n, x.flags) != hipSuccess) { printf("%-30s alloc failed\n", x.name); continue; } }
    else { h = malloc(n); if (x.kind == 2) hipHostRegister(h, n, 0); }
    double ta = now() - t0; t0 = now(); memset(h, 7, n); double tm = now() - t0;
    hipMemcpy(d, h, n, hipMemcpyHostToDevice); hipDeviceSynchronize();
    t0 = now(); for (int i = 0; i < 3; i++) hipMemcpy(d, h, n, hipMemcpyHostToDevice); hipDeviceSynchronize(); double up = (now() - t0) / 3;
    t0 = now(); for (int i = 0; i < 3; i++) hipMemcpy(h, d, n, hipMemcpyDeviceToHost); hipDeviceSynchronize(); double dn = (now() - t0) / 3;
    printf("%-30s alloc %.1f ms, first touch %.1f ms, H2D %.1f GB/s, D2H %.1f GB/s\n", x.name, ta * 1e3, tm * 1e3, n / up / 1e9, n / dn / 1e9);
    if (x.kind == 1) hipHostFree(h); else { if (x.kind == 2) hipHostUnregister(h); free(h); }
  }
  return 0;
}
