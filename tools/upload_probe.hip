// upload_probe.hip -- what does hipMemcpyAsync(H2D) from PAGEABLE host memory promise on this runtime?
//
// DevBuf::upload (mpp_amd/csrc/runtime.hpp) copies from std::vector temporaries that die, or are overwritten, right after the
// call returns.  That is only correct if the source has been consumed when hipMemcpyAsync returns.  For every size this probe
//   1. fills a pageable buffer with pattern A and issues hipMemcpyAsync on a non-blocking stream,
//   2. IMMEDIATELY overwrites the source with pattern B (and, in the "free" variant, frees it and lets malloc reuse it),
//   3. synchronises and reads the device buffer back through a pinned buffer,
// and reports how many elements on the device are not pattern A, plus the time the call took against the time the following
// hipStreamSynchronize took (a call that returns before the data has left the source shows as a short call and a long wait).
// Build: hipcc --offload-arch=gfx950 -O2 tools/upload_probe.hip -o tools/_build/upload_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::printf("%s failed: %s\n", #e, hipGetErrorString(_e)); return 2; } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const size_t sizes[] = {8, 1 << 10, 8 << 10, 64 << 10, 256 << 10, 1 << 20, 4 << 20, 16 << 20, 64 << 20, 256 << 20, 1024u << 20};
  int bad_total = 0;
  std::printf("{\"probe\": \"hipMemcpyAsync H2D from pageable memory, source overwritten right after the call\", \"rows\": [\n");
  for (int variant = 0; variant < 2; ++variant)
    for (size_t bytes : sizes) {
      const size_t n = bytes / 8;
      double *dev = nullptr, *pin = nullptr;
      CK(hipMalloc(&dev, bytes));
      CK(hipHostMalloc(&pin, bytes, hipHostMallocDefault));
      CK(hipMemsetAsync(dev, 0, bytes, s));
      CK(hipStreamSynchronize(s));
      double *src = static_cast<double *>(std::malloc(bytes));
      for (size_t i = 0; i < n; ++i) src[i] = 1.0 + (double)i;          // pattern A
      const double t0 = now();
      CK(hipMemcpyAsync(dev, src, bytes, hipMemcpyHostToDevice, s));
      const double t1 = now();
      if (variant == 0) {
        for (size_t i = 0; i < n; ++i) src[i] = -7.0;                   // pattern B, at once
      } else {                                                          // the temporary dies and its memory is handed out again
        std::free(src);
        src = static_cast<double *>(std::malloc(bytes));
        std::memset(src, 0xFF, bytes);
      }
      const double t2 = now();
      CK(hipStreamSynchronize(s));
      const double t3 = now();
      CK(hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
      size_t bad = 0;
      for (size_t i = 0; i < n; ++i) bad += (pin[i] != 1.0 + (double)i);
      bad_total += bad != 0;
      std::printf("  {\"variant\": \"%s\", \"bytes\": %zu, \"call_ms\": %.3f, \"sync_after_ms\": %.3f, \"not_pattern_A\": %zu},\n",
                  variant == 0 ? "overwrite" : "free+reuse", bytes, (t1 - t0) * 1e3, (t3 - t2) * 1e3, bad);
      std::free(src);
      CK(hipFree(dev));
      CK(hipHostFree(pin));
    }
  std::printf("  {}], \"sizes_with_corruption\": %d}\n", bad_total);
  return 0;
}
