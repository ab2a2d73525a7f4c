// sba_common.hpp -- host-side plumbing shared by the C ABI (sba_api.hip) and the per-camera-model engine translation
// units (sba_engine_ncp11.hip / sba_engine_ncp13.hip): HIP error handling, the per-handle device arena, and the abstract
// engine interface the C ABI dispatches through.
#pragma once
#include "../../include/sba_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>     // types and enum values only: the library itself is bound with dlopen (Rccl below), never linked

namespace sba_host {

inline thread_local std::string g_last_error;

struct HipError { hipError_t e; const char* what; const char* file; int line; };
#define HIPCHK(expr)                                                 \
  do {                                                               \
    hipError_t _e = (expr);                                          \
    if (_e != hipSuccess) throw HipError{_e, #expr, __FILE__, __LINE__}; \
  } while (0)

// Device memory of one handle comes from a few large slabs (bump allocation, released together when the handle dies):
// a handle owns ~40 buffers, and at ~80 us per hipMalloc / hipFree they used to cost more wall time than the solve.
// Slabs released by a dying handle are parked in a small per-process pool instead of going back to the driver: a drop-in
// call creates and destroys a handle around every solve (the instance must stay picklable), and at ~0.3 ms per hipFree /
// hipMalloc of a large block the teardown used to cost more than the solve.  The pool keeps at most POOL_MAX_BYTES.
struct SlabPool {
  struct Slab { char* base; size_t size; int device; };
  static constexpr size_t POOL_MAX_BYTES = (size_t)2 << 30;
  std::mutex mu;
  std::vector<Slab> free_slabs;
  size_t bytes = 0;
  static SlabPool& get() { static SlabPool* p = new SlabPool(); return *p; }     // never destroyed: no HIP calls at process exit
  char* take(size_t min_size, int device, size_t* got) {
    std::lock_guard<std::mutex> lk(mu);
    int best = -1;
    for (int i = 0; i < (int)free_slabs.size(); ++i)
      if (free_slabs[i].device == device && free_slabs[i].size >= min_size && (best < 0 || free_slabs[i].size < free_slabs[best].size)) best = i;
    if (best < 0) return nullptr;
    Slab sl = free_slabs[best];
    free_slabs.erase(free_slabs.begin() + best);
    bytes -= sl.size;
    *got = sl.size;
    return sl.base;
  }
  bool give(char* base, size_t size, int device) {
    std::lock_guard<std::mutex> lk(mu);
    if (bytes + size > POOL_MAX_BYTES) return false;
    free_slabs.push_back({base, size, device});
    bytes += size;
    return true;
  }
};

struct Arena {
  struct Slab { char* base; size_t size, used; };
  std::vector<Slab> slabs;
  int device = 0;
  // slab sizes double from 8 MB to 64 MB: few hipMalloc calls (each large one costs milliseconds on this driver, whatever
  // its size) without grabbing much more than the handle needs
  size_t next_size = (size_t)8 << 20;
  void* take(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    for (auto& sl : slabs)
      if (sl.size - sl.used >= bytes) { void* p = sl.base + sl.used; sl.used += bytes; return p; }
    const size_t want = std::max(bytes, next_size);
    size_t sz = 0;
    char* base = SlabPool::get().take(bytes, device, &sz);
    if (base) {       // a recycled slab looks like a fresh allocation (tens of microseconds).  hipMemset on device memory may return
      HIPCHK(hipMemsetAsync(base, 0, sz, nullptr));     // before it has run, and the engine's stream does not wait for the null
      HIPCHK(hipStreamSynchronize(nullptr));            // stream: make sure the zeros are down before anything is uploaded
    }
    if (!base) {
      sz = want;
      void* b = nullptr;
      HIPCHK(hipMalloc(&b, sz));
      base = static_cast<char*>(b);
    }
    next_size = std::min(next_size * 2, (size_t)64 << 20);
    slabs.push_back({base, sz, bytes});
    return base;
  }
  ~Arena() { for (auto& sl : slabs) if (!SlabPool::get().give(sl.base, sl.size, device)) (void)hipFree(sl.base); }
};
// The per-handle stream, events and pinned state record are recycled the same way (creating and destroying them costs
// about as much as a small solve).  Entries are only taken back after the stream has drained.
struct HostRes {
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;   // for a large read-back that may run beside kernels of the main stream
  hipEvent_t ev[2 + 12] = {};
  hipEvent_t ev_wait = nullptr;        // busy-poll event of Engine::sync_spin (no timing)
  void* pinned = nullptr;            // PINNED_BYTES of pinned host memory: 1 KB LM state mirror + a landing area for small device-to-host copies
  static constexpr size_t PINNED_BYTES = 64 * 1024, PINNED_STATE_BYTES = 1024;
  int device = 0;
};
struct HostResPool {
  std::mutex mu;
  std::vector<HostRes> free_res;
  static HostResPool& get() { static HostResPool* p = new HostResPool(); return *p; }
  bool take(int device, HostRes* out) {
    std::lock_guard<std::mutex> lk(mu);
    for (size_t i = 0; i < free_res.size(); ++i)
      if (free_res[i].device == device) { *out = free_res[i]; free_res.erase(free_res.begin() + i); return true; }
    return false;
  }
  void give(const HostRes& r) {
    std::lock_guard<std::mutex> lk(mu);
    if (free_res.size() < 8) { free_res.push_back(r); return; }
    for (auto e : r.ev) if (e) (void)hipEventDestroy(e);
    if (r.ev_wait) (void)hipEventDestroy(r.ev_wait);
    if (r.stream) (void)hipStreamDestroy(r.stream);
    if (r.copy_stream) (void)hipStreamDestroy(r.copy_stream);
    if (r.pinned) (void)hipHostFree(r.pinned);
  }
};

// Exchange areas exported by handles of THIS process (sba_ipc_export): hipIpcOpenMemHandle refuses a handle of the calling
// process, so ranks that share a process (one handle per thread, or per device of a single-process multi-GPU host) look their
// peers' areas up here and only open what the table does not know.
struct IpcLocalAreas {
  struct Entry { uint8_t key[SBA_IPC_HANDLE_BYTES]; double* p; };
  std::mutex mu;
  std::vector<Entry> entries;
  static IpcLocalAreas& get() { static IpcLocalAreas* t = new IpcLocalAreas(); return *t; }
  void add(const uint8_t* key, double* p) {
    std::lock_guard<std::mutex> lk(mu);
    Entry e; memcpy(e.key, key, sizeof e.key); e.p = p;
    entries.push_back(e);
  }
  void remove(const uint8_t* key) {
    std::lock_guard<std::mutex> lk(mu);
    for (size_t i = 0; i < entries.size(); ++i)
      if (!memcmp(entries[i].key, key, sizeof entries[i].key)) { entries.erase(entries.begin() + i); return; }
  }
  double* find(const uint8_t* key) {
    std::lock_guard<std::mutex> lk(mu);
    for (const Entry& e : entries) if (!memcmp(e.key, key, sizeof e.key)) return e.p;
    return nullptr;
  }
};

inline thread_local Arena* tl_arena = nullptr;      // set for the duration of a call on a handle (ArenaScope)
struct ArenaScope {
  Arena* prev;
  explicit ArenaScope(Arena* a) : prev(tl_arena) { tl_arena = a; }
  ~ArenaScope() { tl_arena = prev; }
};

template <typename U>
struct DevBuf {
  U* p = nullptr;
  size_t n = 0;
  bool pooled = false;
  void alloc(size_t count) {
    free();
    n = count;
    if (!count) return;
    if (tl_arena) { p = static_cast<U*>(tl_arena->take(count * sizeof(U))); pooled = true; }
    else HIPCHK(hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(U)));
  }
  void free() { if (p && !pooled) (void)hipFree(p); p = nullptr; pooled = false; n = 0; }
  ~DevBuf() { free(); }
  void upload(const std::vector<U>& h, hipStream_t s) {
    if (h.size() != n) alloc(h.size());
    if (n) HIPCHK(hipMemcpyAsync(p, h.data(), n * sizeof(U), hipMemcpyHostToDevice, s));
  }
  void zero(hipStream_t s) { if (n) HIPCHK(hipMemsetAsync(p, 0, n * sizeof(U), s)); }
};

// host-side layout loops over the observation list: a handful of threads once the list is long enough to pay for them
template <typename F>
void par_for(int64_t n, F&& f /* (lo, hi, thread) */) {
  const int nt = n < 200000 ? 1 : 4;
  if (nt == 1) { f((int64_t)0, n, 0); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t) th.emplace_back([&f, n, nt, t] { f(n * t / nt, n * (t + 1) / nt, t); });
  for (auto& x : th) x.join();
}

// RCCL, bound at run time (dlopen) so that single-GPU users never load the 0.5 GB library.  The multi-rank LM loop calls
// these on the engine's own stream: one ncclAllReduce (reduced camera system, upper triangle) and one ncclAllGather
// (8 scalars per rank) per LM trial -- SURVEY 8(e); torch.distributed is not involved in a step.
struct Rccl {
  static constexpr int ID_BYTES = NCCL_UNIQUE_ID_BYTES;
  using UniqueId = ncclUniqueId;
  using comm_t = void*;
  // enum values from the header this library was compiled against; load() checks that the library found at run time is of the
  // same major version, i.e. that they mean the same thing there
  enum { kFloat64 = (int)ncclFloat64 };
  enum { kSum = (int)ncclSum, kMax = (int)ncclMax };
  int (*get_unique_id)(UniqueId*) = nullptr;
  int (*comm_init_rank)(comm_t*, int, UniqueId, int) = nullptr;
  int (*comm_destroy)(comm_t) = nullptr;
  int (*all_reduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  int (*all_gather)(const void*, void*, size_t, int, comm_t, hipStream_t) = nullptr;
  int (*get_version)(int*) = nullptr;
  const char* (*error_string)(int) = nullptr;
  bool ok = false;
  int version = 0;
  std::string why;
  static Rccl& get() {
    static Rccl r = load();        // C++11 magic static: initialised once, other threads wait for it
    return r;
  }
  static Rccl load() {
    Rccl r;
    // a copy the process already holds (PyTorch-ROCm bundles its own librccl.so) is reused; otherwise the system library
    void* lib = nullptr;
    for (const char* name : {"librccl.so", "librccl.so.1"}) { lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD); if (lib) break; }
    if (!lib) for (const char* name : {"librccl.so.1", "librccl.so"}) { lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
    if (!lib) {
      const char* e = dlerror();                   // one call: dlerror() clears the message it returns
      r.why = std::string("cannot load librccl.so: ") + (e ? e : "unknown error");
      return r;
    }
    r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(dlsym(lib, "ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(dlsym(lib, "ncclCommInitRank"));
    r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(lib, "ncclCommDestroy"));
    r.all_reduce = reinterpret_cast<decltype(r.all_reduce)>(dlsym(lib, "ncclAllReduce"));
    r.all_gather = reinterpret_cast<decltype(r.all_gather)>(dlsym(lib, "ncclAllGather"));
    r.get_version = reinterpret_cast<decltype(r.get_version)>(dlsym(lib, "ncclGetVersion"));
    r.error_string = reinterpret_cast<decltype(r.error_string)>(dlsym(lib, "ncclGetErrorString"));
    r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_reduce && r.all_gather && r.error_string && r.get_version;
    if (!r.ok) { r.why = "librccl.so lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce / ncclAllGather / ncclGetVersion"; return r; }
    if (r.get_version(&r.version) != 0 || r.version / 10000 != NCCL_MAJOR) {
      r.ok = false;
      r.why = "librccl.so reports version " + std::to_string(r.version) + ", this library was built against the NCCL " +
              std::to_string(NCCL_MAJOR) + ".x API (rccl.h " + std::to_string(NCCL_VERSION_CODE) + ")";
    }
    return r;
  }
};
struct RcclError { int code; const char* what; int line; };
#define RCCLCHK(expr)                                                          \
  do {                                                                         \
    int _r = (expr);                                                           \
    if (_r != 0) throw sba_host::RcclError{_r, #expr, __LINE__};               \
  } while (0)

// One problem handle = one engine.  The concrete class is Engine<T> of the camera model's namespace (sba_engine.hpp);
// the C ABI only sees this interface.
struct EngineBase {
  virtual ~EngineBase() {}
  std::string err;
  Arena arena;       // declared in the base: outlives every DevBuf member of the engine
  virtual void init(const sba_problem_desc& d) = 0;
  virtual int upload(const double* cams, const double* pts, const double* uv, const int64_t* ci, const int64_t* pi, const double* w) = 0;
  virtual int set_params_x(const double* x) = 0;
  virtual int get_params(double* cams_out, double* pts_out) = 0;
  virtual int get_gradient(double* gc_out, double* gp_out) = 0;
  virtual int get_transform(double* theta12) = 0;
  virtual int get_step(double* delta_c_out) = 0;
  virtual int residual(const double* x, double* r_out, double* cost_out) = 0;
  virtual int residual_jacobian(const double* x, double* r_out, double* Jc_out, double* Jp_out) = 0;
  virtual int solve(const sba_lm_opts* o, double* cams_out, double* pts_out, sba_lm_report* rep, sba_lm_iter_log* lg, int cap, int32_t* rows) = 0;
  virtual int64_t exchange_size() const = 0;
  virtual int lm_begin(const sba_lm_opts* o) = 0;
  virtual int lm_linearize() = 0;
  virtual int lm_form_reduced(double* E) = 0;
  virtual int lm_solve_trial(double* E, double* scal) = 0;
  virtual int lm_decide(const double* scal_all, int n_ranks, int32_t* status_out, int32_t* accepted_out, sba_lm_iter_log* row) = 0;
  virtual int lm_decide_async(const double* scal_all, int n_ranks) = 0;
  virtual int lm_poll(int32_t* status_out, int32_t* iterations_out) = 0;
  virtual int lm_run(int32_t* status_out, int32_t* iterations_out) = 0;
  virtual int lm_finish(double* cams_out, double* pts_out, sba_lm_report* rep) = 0;
  virtual int get_log(sba_lm_iter_log* log, int32_t cap, int32_t* rows) = 0;
  virtual int get_kernel_profile(double* total_us, int64_t* count) = 0;
  virtual int time_kernel(const char* name, int reps, double* mean_us) = 0;
  virtual int comm_init(const uint8_t* id, int rank, int n_ranks) = 0;
  virtual int ipc_export(int n_ranks, uint8_t* handle_out) = 0;
  virtual int ipc_attach(int rank, int n_ranks, const uint8_t* handles) = 0;
  virtual int set_fixed_points(const uint8_t* mask) = 0;
  virtual int set_robust_loss(int loss, double f_scale) = 0;
};

}  // namespace sba_host
