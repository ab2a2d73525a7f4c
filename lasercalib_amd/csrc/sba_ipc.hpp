// sba_ipc.hpp -- one-shot exchange between the ranks of a sharded solve through peer-mapped device buffers (no RCCL launch).
//
// Every rank owns one "area" of device memory (uncached, exported with hipIpcGetMemHandle, mapped by every peer with
// hipIpcOpenMemHandle: other processes on the same GPU, or peer GPUs over xGMI).  Per LM trial:
//     k_build_exchange writes the rank's packed reduced system straight into its own area      (slot = exchange count & 1)
//     k_ipc_publish    release-stores the exchange count into the rank's flag
//     k_ipc_gate       ONE wave waits until every peer's flag has reached the count (bounded: 5 s, then the solve fails)
//     k_ipc_sum_system every rank adds the n_ranks copies in rank order -> the same bits everywhere, no broadcast
// and the 8 trial scalars per rank the same way (k_trial_scalars writes them into the area, the gate kernel copies all ranks'
// rows into the local array k_decide reads).  A trial has two exchange points (the decision needs the trial cost of the step the
// system was solved for), each costing two or three ~2.5 us launches instead of an RCCL collective of tens of microseconds on a
// 120 us iteration.  Why a slot may be overwritten two exchanges later: a rank publishes exchange s + 1 only after its own
// k_ipc_sum_system of exchange s (stream order), and nobody passes the gate of s + 1 before every rank has published it.
// The wait sits in a kernel of ONE wave, so that the peers' kernels always find free CUs (two ranks may share a GPU: tests);
// every wave of every kernel here leaves after a bounded time whatever the peers do (tools/micro/ipc_probe.hip).
#pragma once
#include "sba_lm_kernels.hpp"

namespace SBA_NS {

constexpr int IPC_KINDS = 3;                 // 0: reduced system, 1: trial scalars, 2: small host-side vectors (begin / finish)
constexpr int IPC_FLAG_STRIDE = 16;          // doubles (128 bytes) between flags
constexpr long long IPC_TIMEOUT_TICKS = 500000000LL;      // 5 s of the 100 MHz wall clock: the default bound of a gate's wait for a peer (SBA_IPC_TIMEOUT_S)
struct IpcLayout {                           // offsets in doubles from the start of an area
  size_t flag, sys, scal, vec, total;
  int nvec;
  __host__ __device__ static IpcLayout make(int n) {
    IpcLayout L;
    L.nvec = n + 8;
    L.flag = 0;
    L.sys = (size_t)IPC_KINDS * 2 * IPC_FLAG_STRIDE;
    L.scal = L.sys + 2 * ((exch_packed_size(n) + 15) & ~(size_t)15);
    L.vec = L.scal + 2 * 16;
    L.total = L.vec + 2 * (((size_t)L.nvec + 15) & ~(size_t)15);
    return L;
  }
  __host__ __device__ size_t sys_slot(int n, int s) const { return sys + (size_t)s * ((exch_packed_size(n) + 15) & ~(size_t)15); }
  __host__ __device__ size_t scal_slot(int s) const { return scal + (size_t)s * 16; }
  __host__ __device__ size_t vec_slot(int s) const { return vec + (size_t)s * (((size_t)nvec + 15) & ~(size_t)15); }
  __host__ __device__ size_t flag_of(int kind, int s) const { return flag + ((size_t)kind * 2 + s) * IPC_FLAG_STRIDE; }
};

__global__ void k_ipc_publish(double* __restrict__ mine, size_t flag_off, unsigned long long value, const LMState* __restrict__ st) {
  if (st && st->status >= 0) return;           // a finished solve exchanges nothing any more -- on every rank alike
  __threadfence_system();
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(mine + flag_off), value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One wave.  Lane r waits for rank r's flag; afterwards (optional) rows of `ncopy` doubles are copied from every rank's area into
// dst[r * ncopy + i].  On a timeout the solve is stopped: status 0 with the failure flag LMState::comm_fail set.
__global__ __launch_bounds__(64) void k_ipc_gate(double* const* __restrict__ areas, int n_ranks, size_t flag_off, unsigned long long value,
                                                 LMState* __restrict__ st, int* __restrict__ fail /* outside the LM loop (st == NULL) */,
                                                 size_t copy_off, int ncopy, double* __restrict__ dst, long long timeout_ticks) {
  if (st && st->status >= 0) return;
  const int lane = threadIdx.x;
  bool late = false;
  for (int r = lane; r < n_ranks; r += 64) {
    const unsigned long long* f = reinterpret_cast<const unsigned long long*>(areas[r] + flag_off);
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < value) {
      if (wall_clock64() - t0 > timeout_ticks) { late = true; break; }
      __builtin_amdgcn_s_sleep(4);
    }
  }
  const bool any_late = __any(late);
  if (any_late) {
    if (lane == 0 && st) { st->comm_fail = 1; st->status = 0; }
    if (lane == 0 && fail) *fail = 1;
    return;
  }
  for (int i = lane; i < n_ranks * ncopy; i += 64) {
    const int r = i / ncopy, k = i - r * ncopy;
    dst[i] = __builtin_nontemporal_load(areas[r] + copy_off + k);
  }
}

// E (full symmetric layout) = sum over the ranks, in rank order, of their packed systems
__global__ void k_ipc_sum_system(double* const* __restrict__ areas, int n_ranks, size_t slot_off, int n, int free_cams,
                                 double* __restrict__ E, const LMState* __restrict__ st) {
  if (st->status >= 0) return;
  const size_t nn = (size_t)n * n;
  const size_t tail = exch_packed_index(n, n - 1, n - 1) + 1;
  const size_t total = free_cams ? nn + 3 * (size_t)n : 0;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    size_t src;
    if (idx < nn) {
      const int i = (int)(idx / n), j = (int)(idx - (size_t)i * n);
      src = exch_packed_index(n, i < j ? i : j, i < j ? j : i);
    } else {
      src = tail + (idx - nn);
    }
    double s = 0;
    for (int r = 0; r < n_ranks; ++r) s += __builtin_nontemporal_load(areas[r] + slot_off + src);
    E[idx] = s;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double s = 0;
    for (int r = 0; r < n_ranks; ++r) s += __builtin_nontemporal_load(areas[r] + slot_off + exch_packed_size(n) - 1);
    E[nn + 3 * (size_t)n] = s;
  }
}

// out[i] = sum / max over the ranks of their small vectors (lm_begin / lm_finish)
__global__ void k_ipc_reduce_vec(double* const* __restrict__ areas, int n_ranks, size_t slot_off, int count, int take_max, double* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    double s = __builtin_nontemporal_load(areas[0] + slot_off + i);
    for (int r = 1; r < n_ranks; ++r) {
      const double v = __builtin_nontemporal_load(areas[r] + slot_off + i);
      s = take_max ? fmax(s, v) : s + v;
    }
    out[i] = s;
  }
}

}  // namespace SBA_NS
