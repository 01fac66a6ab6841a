/*
 * gs_comm.cpp -- RCCL control plane for host programs that drive several GPUs from ONE process (one host thread per
 * device, gsh_set_device): SURVEY.md 8(e)'s collectives -- the broadcast of the flattened cascade blob, the all-gather
 * of per-file result counts and output checksums, the all-reduce that closes the timing -- all KB-scale; pixel planes
 * never cross GPUs.  The C99 batch driver (grayskull_amd/host/gsbatch.c --gpus N) is the caller; bench.py reaches the
 * same RCCL through torch.distributed, one process per GPU.
 *
 * librccl.so is dlopen()ed on first use, so the library itself does not depend on it.  Three backends behind one interface:
 *   RCCL   ncclCommInitAll over the worker devices (several GPUs; one GPU only when GS_COMM_RCCL=1 asks for it -- a CLI run of
 *          milliseconds should not pay a communicator for a world of one);
 *   LOCAL  a world of one: the "collectives" are stream-ordered local copies;
 *   HOST   several workers without a usable librccl (or GS_COMM_BACKEND=host), and the kernel-logic emulator build (GS_EMU, a
 *          test tool, whose "devices" are host threads): the KB-scale payloads bounce through host memory and the workers'
 *          threads meet at a rendezvous -- slower than RCCL by a host round trip per collective, which is nothing next to
 *          refusing to run.
 */
#include <condition_variable>
#include <mutex>

#include "gs_internal.h"

#ifndef GS_EMU
#include <dlfcn.h>
#include <rccl/rccl.h> /* types and enums only: every function is looked up at run time */
#endif

namespace {

/* rendezvous of `world` host threads with a pointer table (emulator; world-1 fallback never waits) */
struct Meet {
  int world;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  unsigned long long gen = 0;
  const void *ptr[64] = {};
  explicit Meet(int w) : world(w) {}
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    const unsigned long long g = gen;
    if (++arrived == world) {
      arrived = 0, gen++;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return gen != g; });
    }
  }
};

#ifndef GS_EMU
struct Rccl {
  void *so = nullptr;
  int version = 0;
  ncclResult_t (*GetVersion)(int *) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  bool ok = false;
  std::string why;
};
Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *names[] = {getenv("GS_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
      if (!n || !*n) continue;
      if ((r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
      r.why = dlerror();
    }
    if (!r.so) return;
#define GS_SYM(field, name)                                         \
  if (!(r.field = (decltype(r.field))dlsym(r.so, name))) {          \
    r.why = std::string("librccl lacks ") + name;                   \
    return;                                                         \
  }
    GS_SYM(GetVersion, "ncclGetVersion") GS_SYM(CommInitAll, "ncclCommInitAll") GS_SYM(CommDestroy, "ncclCommDestroy")
    GS_SYM(GetErrorString, "ncclGetErrorString") GS_SYM(Broadcast, "ncclBroadcast") GS_SYM(AllGather, "ncclAllGather")
    GS_SYM(AllReduce, "ncclAllReduce")
#undef GS_SYM
    (void)r.GetVersion(&r.version);
    r.ok = true;
  });
  return r;
}
#define GS_NCCL(call)                                                                                          \
  do {                                                                                                         \
    ncclResult_t r_ = (call);                                                                                  \
    if (r_ != ncclSuccess) {                                                                                   \
      fprintf(stderr, "grayskull_hip: %s failed: %s (%s:%d)\n", #call, rccl().GetErrorString(r_), __FILE__, __LINE__); \
      abort();                                                                                                 \
    }                                                                                                          \
  } while (0)
#endif

}  // namespace

struct gsh_comm {
  int rank = 0, world = 1, device = 0;
  enum Kind { LOCAL, RCCL, HOST } kind = LOCAL;
  Meet *meet = nullptr; /* shared by the world's handles (HOST; LOCAL has a world of one) */
#ifndef GS_EMU
  ncclComm_t nc = nullptr;
#endif
  char backend[96] = "local copies (world of 1, librccl not needed)";
};

namespace {
/* the thread that drives a communicator must be the thread whose context sits on its device */
void check_thread(const gsh_comm *c) {
  GS_ASSERT(c != nullptr);
#ifndef GS_EMU
  if (ctx().device_set && ctx().device != c->device) {
    fprintf(stderr, "grayskull_hip: communicator of device %d driven from a thread on device %d\n", c->device, ctx().device);
    abort();
  }
#endif
}
/* world-1 path: a stream-ordered local copy */
void local_copy(void *dst, const void *src, size_t bytes) {
  if (!bytes || dst == src) return;
  GS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx().s()));
}
/* HOST backend: every worker's `bytes` from its device buffer, in rank order, in host memory of every worker */
void host_collect(gsh_comm *c, const void *dev_src, size_t bytes, std::vector<char> &all) {
  std::vector<char> mine(bytes);
  GS_HIP(hipMemcpyAsync(mine.data(), dev_src, bytes, hipMemcpyDeviceToHost, ctx().s()));
  ctx().sync();
  Meet &m = *c->meet;
  m.ptr[c->rank] = mine.data();
  m.wait();
  all.resize((size_t)c->world * bytes);
  for (int r = 0; r < c->world; r++) memcpy(all.data() + (size_t)r * bytes, m.ptr[r], bytes);
  m.wait(); /* everybody has copied: `mine` may go */
}
void host_put(void *dev_dst, const void *host_src, size_t bytes) { /* the source is a local: done before it goes */
  GS_HIP(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx().s()));
  ctx().sync();
}
template <class T> void host_reduce(gsh_comm *c, T *buf, size_t n, int op) {
  if (c->world == 1) return;
  std::vector<char> all;
  host_collect(c, buf, n * sizeof(T), all);
  std::vector<T> acc(n);
  for (size_t i = 0; i < n; i++) {
    T a;
    memcpy(&a, all.data() + i * sizeof(T), sizeof(T));
    for (int r = 1; r < c->world; r++) {
      T b;
      memcpy(&b, all.data() + ((size_t)r * n + i) * sizeof(T), sizeof(T));
      a = op == 0 ? (T)(a + b) : (b > a ? b : a);
    }
    acc[i] = a;
  }
  host_put(buf, acc.data(), n * sizeof(T));
}
}  // namespace

extern "C" {

int gsh_comm_init_all(gsh_comm **comms, int ndev, const int *devices) {
  GS_ASSERT(comms && ndev >= 1 && ndev <= 64);
  for (int i = 0; i < ndev; i++) {
    comms[i] = new gsh_comm();
    comms[i]->rank = i, comms[i]->world = ndev, comms[i]->device = devices ? devices[i] : i;
  }
  auto host_backend = [&](const char *why) {
    Meet *m = new Meet(ndev);
    for (int i = 0; i < ndev; i++) {
      comms[i]->kind = gsh_comm::HOST, comms[i]->meet = m;
      snprintf(comms[i]->backend, sizeof comms[i]->backend, "host rendezvous of %d worker threads (%s)", ndev, why);
    }
    return 0;
  };
#ifdef GS_EMU
  return ndev == 1 ? 0 : host_backend("emulator build, a test tool");
#else
  const char *force = getenv("GS_COMM_BACKEND");
  if (ndev == 1 && !(getenv("GS_COMM_RCCL") && getenv("GS_COMM_RCCL")[0] == '1')) return 0; /* LOCAL: a world of one needs no RCCL */
  if (force && !strcmp(force, "host")) return ndev == 1 ? 0 : host_backend("GS_COMM_BACKEND=host");
  Rccl &r = rccl();
  if (!r.ok) {
    if (ndev == 1) return 0;
    fprintf(stderr, "grayskull_hip: librccl.so cannot be used (%s): the %d workers exchange their results through host memory\n",
            r.why.c_str(), ndev);
    return host_backend("librccl unavailable");
  }
  int before = 0;
  (void)hipGetDevice(&before);
  std::vector<ncclComm_t> nc((size_t)ndev);
  std::vector<int> devs((size_t)ndev);
  for (int i = 0; i < ndev; i++) devs[(size_t)i] = comms[i]->device;
  GS_NCCL(r.CommInitAll(nc.data(), ndev, devs.data()));
  (void)hipSetDevice(before); /* ncclCommInitAll visits every device */
  for (int i = 0; i < ndev; i++) {
    comms[i]->kind = gsh_comm::RCCL, comms[i]->nc = nc[(size_t)i];
    snprintf(comms[i]->backend, sizeof comms[i]->backend, "rccl %d.%d.%d (ncclCommInitAll, %d rank%s)", r.version / 10000,
             (r.version / 100) % 100, r.version % 100, ndev, ndev == 1 ? "" : "s");
  }
  return 0;
#endif
}

void gsh_comm_destroy_all(gsh_comm **comms, int ndev) {
  if (!comms) return;
  Meet *m = ndev > 0 && comms[0] ? comms[0]->meet : nullptr;
  for (int i = 0; i < ndev; i++) {
    if (!comms[i]) continue;
#ifndef GS_EMU
    if (comms[i]->kind == gsh_comm::RCCL && comms[i]->nc) (void)rccl().CommDestroy(comms[i]->nc);
#endif
    delete comms[i];
    comms[i] = nullptr;
  }
  delete m;
}

int gsh_comm_rank(const gsh_comm *c) { return c ? c->rank : 0; }
int gsh_comm_world(const gsh_comm *c) { return c ? c->world : 1; }
const char *gsh_comm_backend(const gsh_comm *c) { return c ? c->backend : "none"; }

/* all of them: device buffers, enqueued on the calling thread's stream (gsh_sync() before the host reads a result) */
void gsh_comm_broadcast(gsh_comm *c, void *buf, size_t bytes, int root) {
  check_thread(c);
  GS_ASSERT(root >= 0 && root < c->world && (buf || bytes == 0));
  if (bytes == 0) return;
#ifndef GS_EMU
  if (c->kind == gsh_comm::RCCL) {
    GS_NCCL(rccl().Broadcast(buf, buf, bytes, ncclUint8, root, c->nc, ctx().s()));
    return;
  }
#endif
  if (c->world == 1) return;
  std::vector<char> all;
  host_collect(c, buf, bytes, all);
  if (c->rank != root) host_put(buf, all.data() + (size_t)root * bytes, bytes);
}

void gsh_comm_all_gather(gsh_comm *c, const void *send, void *recv, size_t bytes_per_rank) {
  check_thread(c);
  GS_ASSERT((send && recv) || bytes_per_rank == 0);
  if (bytes_per_rank == 0) return;
#ifndef GS_EMU
  if (c->kind == gsh_comm::RCCL) {
    GS_NCCL(rccl().AllGather(send, recv, bytes_per_rank, ncclUint8, c->nc, ctx().s()));
    return;
  }
#endif
  if (c->world == 1) {
    local_copy(recv, send, bytes_per_rank);
    return;
  }
  std::vector<char> all;
  host_collect(c, send, bytes_per_rank, all);
  host_put(recv, all.data(), all.size());
}

/* op: 0 = sum, 1 = max; in place */
void gsh_comm_all_reduce_u64(gsh_comm *c, unsigned long long *buf, size_t n, int op) {
  check_thread(c);
  GS_ASSERT((buf || n == 0) && (op == 0 || op == 1));
  if (n == 0) return;
#ifndef GS_EMU
  if (c->kind == gsh_comm::RCCL) {
    GS_NCCL(rccl().AllReduce(buf, buf, n, ncclUint64, op == 0 ? ncclSum : ncclMax, c->nc, ctx().s()));
    return;
  }
#endif
  host_reduce(c, buf, n, op);
}
void gsh_comm_all_reduce_f64(gsh_comm *c, double *buf, size_t n, int op) {
  check_thread(c);
  GS_ASSERT((buf || n == 0) && (op == 0 || op == 1));
  if (n == 0) return;
#ifndef GS_EMU
  if (c->kind == gsh_comm::RCCL) {
    GS_NCCL(rccl().AllReduce(buf, buf, n, ncclFloat64, op == 0 ? ncclSum : ncclMax, c->nc, ctx().s()));
    return;
  }
#endif
  host_reduce(c, buf, n, op);
}

}  /* extern "C" */
