// handoff_bench.hip -- how long does it take one wave to see what another wave of the same launch wrote?
// Two waves (workgroups a and b of one launch; with `stride` between them so that they land on different XCDs / CUs) play ping-pong
// through device memory with agent-scope (sc1) stores and loads, the mechanism of the chained / queued ILU(0) solves:
//   flag only            : a stores k, b polls until it reads k, stores k to the reply word, a polls ...
//   payload + flag       : a stores 512 B (one 8-byte value per lane, sc1), waits for the acknowledgement (vmcnt(0)), stores the flag;
//                          b polls the flag, then loads the 512 B with sc1 loads and checks them
//   payload as the flag  : a stores the 512 B only; b polls the payload itself until no lane holds the previous round's value
// Reported: microseconds per one-way hand-off.  Build: hipcc --offload-arch=gfx950 -O2 tools/handoff_bench.hip -o tools/_build/handoff_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::printf("%s failed: %s\n", #e, hipGetErrorString(_e)); return 2; } } while (0)

__device__ __forceinline__ unsigned ld(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ldd(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void std_(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int MODE, int SLEEP>
__global__ __launch_bounds__(64) void k_pingpong(int wa, int wb, int rounds, unsigned *flags, double *payload, unsigned *bad) {
  const int me = blockIdx.x == wa ? 0 : (blockIdx.x == wb ? 1 : -1);
  if (me < 0) return;
  const int lane = threadIdx.x;
  unsigned *mine = flags + 64 * me, *other = flags + 64 * (1 - me);     // (different cache lines)
  double *pmine = payload + 64 * me, *pother = payload + 64 * (1 - me);
  for (int k = 1; k <= rounds; ++k) {
    if (ld(bad) >= 1000000u) return;   // a wait gave up: end the game (never hang the device)
    if (me == 0) {   // send k, wait for the reply k
      if (MODE >= 1) std_(pmine + lane, (double)k);
      if (MODE == 1) __builtin_amdgcn_s_waitcnt(0x0F70);
      if (MODE <= 1 && lane == 0) st(mine, (unsigned)k);
    }
    // wait for round k of the other side (side 1 waits for the message, side 0 for the reply)
    if (MODE <= 1) {
      if (lane == 0) { unsigned spins = 0; while (ld(other) != (unsigned)k) { if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP); if (++spins > 4000000u) { atomicAdd(bad, 1000000u); break; } } }
      __builtin_amdgcn_wave_barrier();
      if (MODE == 1) { const double v = ldd(pother + lane); if (v != (double)k) atomicAdd(bad, 1u); }
    } else {
      for (unsigned spins = 0;; ++spins) { const double v = ldd(pother + lane); if (__all(v == (double)k)) break; if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
                                           if (spins > 4000000u) { if (lane == 0) atomicAdd(bad, 1000000u); break; } }
    }
    if (me == 1) {   // reply
      if (MODE >= 1) std_(pmine + lane, (double)k);
      if (MODE == 1) __builtin_amdgcn_s_waitcnt(0x0F70);
      if (MODE <= 1 && lane == 0) st(mine, (unsigned)k);
    }
  }
}

int main() {
  unsigned *flags, *bad;
  double *payload;
  CK(hipMalloc(&flags, 4096)); CK(hipMalloc(&payload, 4096)); CK(hipMalloc(&bad, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rounds = 2000;
  std::printf("{\"what\": \"one-way hand-off between two waves of one launch, microseconds\", \"rows\": [\n");
  const int pairs[][2] = {{0, 1}, {0, 8}, {0, 9}, {0, 4}, {0, 256}, {3, 1027}};
  for (auto &pr : pairs)
    for (int mode = 0; mode < 3; ++mode)
      for (int sl = 0; sl < 2; ++sl) {
        CK(hipMemset(flags, 0, 4096)); CK(hipMemset(payload, 0, 4096)); CK(hipMemset(bad, 0, 4));
        const int grid = pr[1] + 1;
        CK(hipEventRecord(e0, 0));
#define RUN(M, S) hipLaunchKernelGGL((k_pingpong<M, S>), dim3(grid), dim3(64), 0, 0, pr[0], pr[1], rounds, flags, payload, bad)
        if (mode == 0) { if (sl) RUN(0, 4); else RUN(0, 0); }
        else if (mode == 1) { if (sl) RUN(1, 4); else RUN(1, 0); }
        else { if (sl) RUN(2, 4); else RUN(2, 0); }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned nb = 0;
        CK(hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost));
        std::printf("  {\"workgroups\": [%d, %d], \"mode\": \"%s\", \"s_sleep\": %d, \"one_way_us\": %.3f, \"stale_payload_reads\": %u},\n", pr[0], pr[1],
                    mode == 0 ? "flag only" : mode == 1 ? "payload, ack, flag" : "payload is the flag", sl ? 4 : 0, ms * 1e3 / (2.0 * rounds), nb);
      }
  std::printf("  {}]}\n");
  return 0;
}
