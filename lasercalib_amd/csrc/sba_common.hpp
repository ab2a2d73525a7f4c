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
#include <string>
#include <thread>
#include <vector>

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
struct Arena {
  struct Slab { char* base; size_t size, used; };
  std::vector<Slab> slabs;
  // slab sizes double from 8 MB to 64 MB: few hipMalloc calls (each large one costs milliseconds on this driver, whatever
  // its size) without grabbing much more than the handle needs
  size_t next_size = (size_t)8 << 20;
  void* take(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    for (auto& sl : slabs)
      if (sl.size - sl.used >= bytes) { void* p = sl.base + sl.used; sl.used += bytes; return p; }
    const size_t sz = std::max(bytes, next_size);
    void* base = nullptr;
    HIPCHK(hipMalloc(&base, sz));
    next_size = std::min(next_size * 2, (size_t)64 << 20);
    slabs.push_back({static_cast<char*>(base), sz, bytes});
    return base;
  }
  ~Arena() { for (auto& sl : slabs) (void)hipFree(sl.base); }
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
  virtual int lm_finish(double* cams_out, double* pts_out, sba_lm_report* rep) = 0;
  virtual int get_log(sba_lm_iter_log* log, int32_t cap, int32_t* rows) = 0;
  virtual int get_kernel_profile(double* total_us, int64_t* count) = 0;
  virtual int time_kernel(const char* name, int reps, double* mean_us) = 0;
};

}  // namespace sba_host
