/*
 * gs_ctx.cpp -- the per-thread context of libgrayskull_hip.so and the device / stream / memory / tuning entry points of
 * the C ABI (include/grayskull_hip.h).  There is no CPU fallback: without a HIP device every entry point aborts.
 */
#include "gs_internal.h"

namespace gsi {
Ctx &ctx() {
  static thread_local Ctx c;
  return c;
}
TuneTable g_tune;
}  // namespace gsi

extern "C" {

const char *gsh_version(void) {
#ifdef GS_EMU
  return "grayskull_hip 0.1 (kernel-logic emulator build -- test tool, not a product)";
#else
  return "grayskull_hip 0.1 gfx950";
#endif
}
int gsh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
void gsh_set_device(int ordinal) {
  Ctx &c = ctx();
  if (c.device_set && c.device != ordinal) c.release();
  c.device = ordinal;
  c.device_set = false;
  c.ensure_device();
}
void gsh_set_stream(void *s) {
  Ctx &c = ctx();
  if (c.own_stream) {
    c.sync();
    (void)hipStreamDestroy(c.stream);
    c.own_stream = false;
  }
  c.stream = (hipStream_t)s;
  c.user_stream = s != nullptr;
}
void *gsh_get_stream(void) { return (void *)ctx().s(); }
void gsh_set_async(int on) { ctx().async = on != 0; }
void gsh_profile(int on) {
#ifndef GS_EMU
  Ctx &c = ctx();
  c.ensure_device();
  c.prof_on = on != 0;
  c.prof_n = 0;
  /* on > 1: create that many event pairs now, so that none is created inside a timed region */
  for (int i = 0; on > 1 && i < 2 * std::min(on, (int)Ctx::kProfPairs); i++)
    if (!c.prof_ev[i]) GS_HIP(hipEventCreateWithFlags(&c.prof_ev[i], sync_event_flags())); /* timing on, no system fence */
#else
  (void)on;
#endif
}
unsigned gsh_profile_read(double *total_ms) {
  unsigned n = 0;
  double sum = 0;
#ifndef GS_EMU
  Ctx &c = ctx();
  c.sync();
  for (unsigned i = 0; i < c.prof_n; i++) {
    float ms = 0;
    GS_HIP(hipEventSynchronize(c.prof_ev[2 * i + 1]));
    GS_HIP(hipEventElapsedTime(&ms, c.prof_ev[2 * i], c.prof_ev[2 * i + 1]));
    sum += ms;
  }
  n = c.prof_n;
  c.prof_n = 0;
#endif
  if (total_ms) *total_ms = sum;
  return n;
}
void gsh_tune(int key, int value) {
#ifndef GS_EXPERIMENT
  if ((key == 16 || key == 22 || key == 23) && value != 0) { /* the keys that change results exist in experiment builds only */
    fprintf(stderr, "grayskull_hip: gsh_tune(%d, %d) ignored: that probe needs a -DGS_EXPERIMENT build (make experiment)\n", key, value);
    return;
  }
#endif
  if (key >= 0 && key < 32) g_tune.set(key, value);
}
void gsh_sync(void) { ctx().sync(); }
void gsh_shutdown(void) { ctx().release(); }
void *gsh_malloc(size_t bytes) {
  ctx().ensure_device();
  void *p = nullptr;
  GS_HIP(hipMalloc(&p, bytes ? bytes : 1));
  return p;
}
void gsh_free(void *p) {
  if (p) GS_HIP(hipFree(p));
}
void *gsh_host_alloc(size_t bytes) {
  ctx().ensure_device();
  void *p = nullptr;
  GS_HIP(hipHostMalloc(&p, bytes ? bytes : 1, 0));
  return p;
}
void gsh_host_free(void *p) {
  if (p) GS_HIP(hipHostFree(p));
}
void gsh_memset(void *dev, int byte, size_t bytes) {
  GS_HIP(hipMemsetAsync(dev, byte, bytes, ctx().s()));
}
void gsh_upload(void *dev, const void *host, size_t bytes) {
  GS_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, ctx().s()));
  ctx().sync();
}
void gsh_download(void *host, const void *dev, size_t bytes) {
  GS_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx().s()));
  ctx().sync();
}
int gsh_is_device_ptr(const void *p) { return is_dev(p) ? 1 : 0; }

}  /* extern "C" */
