// ipc_probe.hip -- can two PROCESSES on one MI355X map each other's device buffers (hipIpcGetMemHandle / hipIpcOpenMemHandle) and
// hand data over with a flag, a bounded device-side wait on the reader's side and no host synchronisation in between?
// (the mechanism of the engine's one-shot exchange, sba_ipc_*: DESIGN.md 6).  Forks BEFORE any HIP call; the two processes swap
// their 64-byte handles through pipes.  Prints per process: allocation kind, rounds, wait cycles, checksum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/wait.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[%d] %s failed: %s\n", getpid(), #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int NDATA = 16384;       // doubles per slot (128 KB: the size of a 176-row packed system)
struct Area { double data[2][NDATA]; unsigned long long flag[2][16]; };

__global__ void k_produce(Area* mine, int round, double seed) {
  const int slot = round & 1;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < NDATA; i += gridDim.x * blockDim.x) mine->data[slot][i] = seed + i;
}
__global__ void k_publish(Area* mine, int round) {
  __threadfence_system();
  __hip_atomic_store(&mine->flag[round & 1][0], (unsigned long long)(round + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_gate(const Area* peer, int round, long long* waited, int* timed_out) {
  const long long t0 = wall_clock64();
  long long t = t0;
  while (__hip_atomic_load(&peer->flag[round & 1][0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)(round + 1)) {
    t = wall_clock64();
    if (t - t0 > 200000000LL) { *timed_out = 1; break; }          // 2 s at 100 MHz: every wave leaves, whatever the peer does
    __builtin_amdgcn_s_sleep(8);
  }
  *waited += wall_clock64() - t0;
}
__global__ void k_consume(const Area* mine, const Area* peer, int round, double* out) {
  const int slot = round & 1;
  double s = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < NDATA; i += gridDim.x * blockDim.x) s += mine->data[slot][i] + peer->data[slot][i];
  atomicAdd(out, s);
}

int main() {
  int ab[2], ba[2];
  if (pipe(ab) || pipe(ba)) return 1;
  const pid_t child = fork();
  const int rank = child == 0 ? 1 : 0;
  const int rfd = rank == 0 ? ba[0] : ab[0], wfd = rank == 0 ? ab[1] : ba[1];
  CHK(hipSetDevice(0));
  Area* mine = nullptr;
  const char* kind = "uncached";
  if (hipExtMallocWithFlags(reinterpret_cast<void**>(&mine), sizeof(Area), hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    kind = "plain";
    CHK(hipMalloc(reinterpret_cast<void**>(&mine), sizeof(Area)));
  }
  CHK(hipMemset(mine, 0, sizeof(Area)));
  CHK(hipDeviceSynchronize());
  hipIpcMemHandle_t hm, hp;
  CHK(hipIpcGetMemHandle(&hm, mine));
  if (write(wfd, &hm, sizeof hm) != (ssize_t)sizeof hm || read(rfd, &hp, sizeof hp) != (ssize_t)sizeof hp) return 3;
  Area* peer = nullptr;
  CHK(hipIpcOpenMemHandle(reinterpret_cast<void**>(&peer), hp, hipIpcMemLazyEnablePeerAccess));
  long long* waited; int* tout; double* out;
  CHK(hipMalloc(&waited, 8)); CHK(hipMalloc(&tout, 4)); CHK(hipMalloc(&out, 8));
  CHK(hipMemset(waited, 0, 8)); CHK(hipMemset(tout, 0, 4)); CHK(hipMemset(out, 0, 8));
  hipStream_t st;
  CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const int rounds = 200;
  for (int r = 0; r < rounds; ++r) {          // enqueued back to back: no host synchronisation between the rounds
    hipLaunchKernelGGL(k_produce, dim3(64), dim3(256), 0, st, mine, r, (double)(rank * 1000 + r));
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, st, mine, r);
    hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, st, peer, r, waited, tout);
    hipLaunchKernelGGL(k_consume, dim3(64), dim3(256), 0, st, mine, peer, r, out);
  }
  CHK(hipStreamSynchronize(st));
  long long hw; int ht; double ho;
  CHK(hipMemcpy(&hw, waited, 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&ht, tout, 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&ho, out, 8, hipMemcpyDeviceToHost));
  double expect = 0;
  for (int r = 0; r < rounds; ++r) expect += (double)NDATA * (1000.0 + 2.0 * r) + 2.0 * ((double)NDATA * (NDATA - 1) / 2);
  printf("[rank %d] %s memory, %d rounds, waited %.1f us per round, timed out %d, checksum %s (%.6g vs %.6g)\n", rank, kind, rounds,
         hw / 100.0 / rounds, ht, ho == expect ? "OK" : "WRONG", ho, expect);
  CHK(hipIpcCloseMemHandle(peer));
  if (rank == 0) { int stt = 0; waitpid(child, &stt, 0); return ho == expect && !ht && WIFEXITED(stt) && WEXITSTATUS(stt) == 0 ? 0 : 1; }
  return ho == expect && !ht ? 0 : 1;
}
