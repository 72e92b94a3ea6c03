// syncbench.hip -- how long the host waits for a handful of device scalars: (a) hipMemcpyAsync to pinned memory +
// hipStreamSynchronize (what fetch_scalars does), (b) a one-block kernel that writes them and then a sequence word straight into
// pinned host memory while the host polls that word.  Each behind a short kernel, as in the solver.
//   hipcc --offload-arch=gfx950 -O3 -o syncbench tools/syncbench.hip && ./syncbench
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_work(double *sc, int n) {
  if (threadIdx.x == 0 && blockIdx.x == 0) for (int i = 0; i < n; ++i) sc[i] += 1.0;
}
__global__ void k_publish(const double *sc, double *host, volatile unsigned long long *seq, unsigned long long v, int n) {
  if (threadIdx.x < n) host[threadIdx.x] = sc[threadIdx.x];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) { *seq = v; __threadfence_system(); }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const int n = 64, reps = 2000;
  double *sc, *h;
  unsigned long long *seq;
  hipStream_t s;
  CK(hipStreamCreate(&s));
  CK(hipMalloc(&sc, n * sizeof(double)));
  CK(hipMemset(sc, 0, n * sizeof(double)));
  CK(hipHostMalloc(&h, n * sizeof(double), hipHostMallocDefault));
  CK(hipHostMalloc(&seq, sizeof(*seq), hipHostMallocDefault));
  *seq = 0;
  for (int mode = 0; mode < 2; ++mode)
    for (int pass = 0; pass < 2; ++pass) {   // pass 0 warms up
      const double t0 = now();
      for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s, sc, n);
        if (mode == 0) {
          CK(hipMemcpyAsync(h, sc, n * sizeof(double), hipMemcpyDeviceToHost, s));
          CK(hipStreamSynchronize(s));
        } else {
          const unsigned long long want = (unsigned long long)pass * reps + r + 1;
          hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s, sc, h, seq, want, n);
          while (*(volatile unsigned long long *)seq != want) __builtin_ia32_pause();
        }
      }
      const double dt = now() - t0;
      if (pass == 1) std::printf("%s: %.2f us per round trip (kernel + read-back), last value %.0f\n", mode == 0 ? "memcpy + stream synchronize " : "kernel publish + host polling", 1e6 * dt / reps, h[0]);
    }
  CK(hipStreamSynchronize(s));
  return 0;
}
