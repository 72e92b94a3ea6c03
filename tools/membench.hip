// membench.hip -- streaming-pattern micro-benchmarks on MI355X: which copy/read/write shapes reach the HBM ceiling.
// build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/_build/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int B = 256;
typedef double2 d2;

__global__ __launch_bounds__(B) void copy_gs(size_t n2, const d2 *__restrict__ x, d2 *__restrict__ y) {
  for (size_t q = (size_t)blockIdx.x * B + threadIdx.x; q < n2; q += (size_t)gridDim.x * B) y[q] = x[q];
}
__global__ __launch_bounds__(B) void copy_gs_nt(size_t n2, const d2 *__restrict__ x, d2 *__restrict__ y) {
  for (size_t q = (size_t)blockIdx.x * B + threadIdx.x; q < n2; q += (size_t)gridDim.x * B) {
    d2 v; v.x = __builtin_nontemporal_load(&x[q].x); v.y = __builtin_nontemporal_load(&x[q].y);
    __builtin_nontemporal_store(v.x, &y[q].x); __builtin_nontemporal_store(v.y, &y[q].y);
  }
}
__global__ __launch_bounds__(B) void copy_gs_u4(size_t n2, const d2 *__restrict__ x, d2 *__restrict__ y) {
  const size_t stride = (size_t)gridDim.x * B;
  size_t q = (size_t)blockIdx.x * B + threadIdx.x;
  for (; q + 3 * stride < n2; q += 4 * stride) {
    d2 a = x[q], b = x[q + stride], c = x[q + 2 * stride], d = x[q + 3 * stride];
    y[q] = a; y[q + stride] = b; y[q + 2 * stride] = c; y[q + 3 * stride] = d;
  }
  for (; q < n2; q += stride) y[q] = x[q];
}
// block-contiguous: block b owns one contiguous chunk
__global__ __launch_bounds__(B) void copy_chunk(size_t n2, const d2 *__restrict__ x, d2 *__restrict__ y) {
  const size_t chunk = (n2 + gridDim.x - 1) / gridDim.x;
  const size_t lo = chunk * blockIdx.x, hi = lo + chunk < n2 ? lo + chunk : n2;
  for (size_t q = lo + threadIdx.x; q < hi; q += B) y[q] = x[q];
}
__global__ __launch_bounds__(B) void copy_chunk_u4(size_t n2, const d2 *__restrict__ x, d2 *__restrict__ y) {
  const size_t chunk = (n2 + gridDim.x - 1) / gridDim.x;
  const size_t lo = chunk * blockIdx.x, hi = lo + chunk < n2 ? lo + chunk : n2;
  size_t q = lo + threadIdx.x;
  for (; q + 3 * B < hi; q += 4 * B) {
    d2 a = x[q], b = x[q + B], c = x[q + 2 * B], d = x[q + 3 * B];
    y[q] = a; y[q + B] = b; y[q + 2 * B] = c; y[q + 3 * B] = d;
  }
  for (; q < hi; q += B) y[q] = x[q];
}
__global__ __launch_bounds__(B) void copy_1pt(size_t n2, const d2 *__restrict__ x, d2 *__restrict__ y) {
  const size_t q = (size_t)blockIdx.x * B + threadIdx.x;
  if (q < n2) y[q] = x[q];
}
__global__ __launch_bounds__(B) void read_gs(size_t n2, const d2 *__restrict__ x, double *out) {
  double acc = 0;
  for (size_t q = (size_t)blockIdx.x * B + threadIdx.x; q < n2; q += (size_t)gridDim.x * B) { d2 v = x[q]; acc += v.x + v.y; }
  if (acc == 1.2345e300) out[0] = acc;
}
__global__ __launch_bounds__(B) void read_chunk(size_t n2, const d2 *__restrict__ x, double *out) {
  const size_t chunk = (n2 + gridDim.x - 1) / gridDim.x;
  const size_t lo = chunk * blockIdx.x, hi = lo + chunk < n2 ? lo + chunk : n2;
  double acc = 0;
  for (size_t q = lo + threadIdx.x; q < hi; q += B) { d2 v = x[q]; acc += v.x + v.y; }
  if (acc == 1.2345e300) out[0] = acc;
}
__global__ __launch_bounds__(B) void write_gs(size_t n2, d2 *__restrict__ y) {
  for (size_t q = (size_t)blockIdx.x * B + threadIdx.x; q < n2; q += (size_t)gridDim.x * B) y[q] = make_double2(1.0, 2.0);
}
// 8-byte-per-lane forms (what a lane-per-cell kernel issues)
__global__ __launch_bounds__(B) void copy8_1pt(size_t n, const double *__restrict__ x, double *__restrict__ y) {
  const size_t i = (size_t)blockIdx.x * B + threadIdx.x;
  if (i < n) y[i] = x[i];
}
__global__ __launch_bounds__(B) void triad8_1pt(size_t n, const double *__restrict__ a, const double *__restrict__ b,
                                                const double *__restrict__ c, double *__restrict__ y) {
  const size_t i = (size_t)blockIdx.x * B + threadIdx.x;
  if (i < n) y[i] = a[i] + 0.5 * b[i] + c[i];
}

// ceiling of the CSR-stream access mix without gather or LDS: per 64-row tile 3 coalesced val loads (8 B/lane),
// 3 coalesced col loads (4 B/lane), rowptr (4 B/lane), x once (8 B/lane), y store (8 B/lane) = 56 B/row
__global__ __launch_bounds__(B) void csr_mix(size_t n, const double *__restrict__ val, const int *__restrict__ col,
                                             const int *__restrict__ rp, const double *__restrict__ x, double *__restrict__ y) {
  const size_t i = (size_t)blockIdx.x * B + threadIdx.x;
  if (i >= n) return;
  const size_t t0 = (i >> 6) * 192, lane = i & 63;
  double s = x[i] + (double)rp[i];
#pragma unroll
  for (int j = 0; j < 3; ++j) s += val[t0 + lane + 64 * j] * (double)col[t0 + lane + 64 * j];
  y[i] = s;
}
__global__ __launch_bounds__(B) void csr_mix_nt(size_t n, const double *__restrict__ val, const int *__restrict__ col,
                                                const int *__restrict__ rp, const double *__restrict__ x, double *__restrict__ y) {
  const size_t i = (size_t)blockIdx.x * B + threadIdx.x;
  if (i >= n) return;
  const size_t t0 = (i >> 6) * 192, lane = i & 63;
  double s = x[i] + (double)__builtin_nontemporal_load(rp + i);
#pragma unroll
  for (int j = 0; j < 3; ++j)
    s += __builtin_nontemporal_load(val + t0 + lane + 64 * j) * (double)__builtin_nontemporal_load(col + t0 + lane + 64 * j);
  __builtin_nontemporal_store(s, y + i);
}

// csr_mix_nt + the pieces a real CSR SpMV adds, one at a time
template <int GATHER3, int LDSRT, int RP2>
__global__ __launch_bounds__(B) void csr_mix_x(size_t n, const double *__restrict__ val, const int *__restrict__ col,
                                               const int *__restrict__ rp, const double *__restrict__ x, double *__restrict__ y) {
  __shared__ double lds[B / 64][192];
  const size_t i = (size_t)blockIdx.x * B + threadIdx.x;
  if (i >= n) return;
  const size_t t0 = (i >> 6) * 192, lane = i & 63;
  const int w = threadIdx.x >> 6;
  double s = (double)__builtin_nontemporal_load(rp + i);
  if (RP2) s += (double)__builtin_nontemporal_load(rp + i + 1);
  double p[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const size_t k = t0 + lane + 64 * j;
    const double v = __builtin_nontemporal_load(val + k);
    const int c = __builtin_nontemporal_load(col + k);
    if (GATHER3) {
      // entry k of the tile belongs to row k/3 and column offset (k%3 - 1)*64: same addresses a 3-point SpMV gathers
      const size_t kk = lane + 64 * j;
      long long xi = (long long)((i >> 6) << 6) + (long long)(kk / 3) + ((long long)(kk % 3) - 1) * 64;
      if (xi < 0) xi = 0;
      if (xi >= (long long)n) xi = n - 1;
      p[j] = v * x[xi] + (double)c;
    } else {
      p[j] = v * (double)c;
    }
  }
  if (!GATHER3) s += x[i];
  if (LDSRT) {
#pragma unroll
    for (int j = 0; j < 3; ++j) lds[w][lane + 64 * j] = p[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < 3; ++j) s += lds[w][3 * lane + j];
  } else {
    s += p[0] + p[1] + p[2];
  }
  __builtin_nontemporal_store(s, y + i);
}

template <class F>
double timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char **argv) {
  size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 104857600ull;
  const int reps = 20;
  double *x, *y, *z, *w, *out;
  CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8)); CK(hipMalloc(&z, n * 8)); CK(hipMalloc(&w, n * 8)); CK(hipMalloc(&out, 64));
  CK(hipMemset(x, 0, n * 8)); CK(hipMemset(y, 0, n * 8)); CK(hipMemset(z, 0, n * 8)); CK(hipMemset(w, 0, n * 8));
  const size_t n2 = n / 2;
  auto rep = [&](const char *name, double ms, double bytes) { printf("%-28s %8.4f ms %8.1f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
  for (int grid : {1024, 2048, 4096, 8192, 16384, 65536}) {
    char nm[64];
    snprintf(nm, 64, "copy_gs g=%d", grid); rep(nm, timeit([&] { copy_gs<<<grid, B>>>(n2, (d2 *)x, (d2 *)y); }, reps), 16.0 * n);
    snprintf(nm, 64, "copy_chunk g=%d", grid); rep(nm, timeit([&] { copy_chunk<<<grid, B>>>(n2, (d2 *)x, (d2 *)y); }, reps), 16.0 * n);
  }
  rep("copy_gs_nt g=4096", timeit([&] { copy_gs_nt<<<4096, B>>>(n2, (d2 *)x, (d2 *)y); }, reps), 16.0 * n);
  rep("copy_gs_u4 g=2048", timeit([&] { copy_gs_u4<<<2048, B>>>(n2, (d2 *)x, (d2 *)y); }, reps), 16.0 * n);
  rep("copy_chunk_u4 g=2048", timeit([&] { copy_chunk_u4<<<2048, B>>>(n2, (d2 *)x, (d2 *)y); }, reps), 16.0 * n);
  rep("copy_chunk_u4 g=8192", timeit([&] { copy_chunk_u4<<<8192, B>>>(n2, (d2 *)x, (d2 *)y); }, reps), 16.0 * n);
  rep("copy_1pt (16B/lane)", timeit([&] { copy_1pt<<<(unsigned)((n2 + B - 1) / B), B>>>(n2, (d2 *)x, (d2 *)y); }, reps), 16.0 * n);
  rep("copy8_1pt (8B/lane)", timeit([&] { copy8_1pt<<<(unsigned)((n + B - 1) / B), B>>>(n, x, y); }, reps), 16.0 * n);
  rep("triad8_1pt 3r1w (8B/lane)", timeit([&] { triad8_1pt<<<(unsigned)((n + B - 1) / B), B>>>(n, x, z, w, y); }, reps), 32.0 * n);
  rep("read_gs g=4096", timeit([&] { read_gs<<<4096, B>>>(n2, (d2 *)x, out); }, reps), 8.0 * n);
  rep("read_gs g=16384", timeit([&] { read_gs<<<16384, B>>>(n2, (d2 *)x, out); }, reps), 8.0 * n);
  rep("read_chunk g=4096", timeit([&] { read_chunk<<<4096, B>>>(n2, (d2 *)x, out); }, reps), 8.0 * n);
  rep("write_gs g=4096", timeit([&] { write_gs<<<4096, B>>>(n2, (d2 *)y); }, reps), 8.0 * n);
  {
    double *val; int *col, *rp;
    CK(hipMalloc(&val, 3 * n * 8)); CK(hipMalloc(&col, 3 * n * 4)); CK(hipMalloc(&rp, (n + 64) * 4));
    CK(hipMemset(val, 0, 3 * n * 8)); CK(hipMemset(col, 0, 3 * n * 4)); CK(hipMemset(rp, 0, (n + 64) * 4));
    rep("csr_mix 56B/row", timeit([&] { csr_mix<<<(unsigned)((n + B - 1) / B), B>>>(n, val, col, rp, x, y); }, reps), 56.0 * n);
    rep("csr_mix_nt 56B/row", timeit([&] { csr_mix_nt<<<(unsigned)((n + B - 1) / B), B>>>(n, val, col, rp, x, y); }, reps), 56.0 * n);
    const unsigned g1 = (unsigned)((n + B - 1) / B);
    rep("mix_nt base(templ)", timeit([&] { csr_mix_x<0, 0, 0><<<g1, B>>>(n, val, col, rp, x, y); }, reps), 56.0 * n);
    rep("mix_nt +rp2", timeit([&] { csr_mix_x<0, 0, 1><<<g1, B>>>(n, val, col, rp, x, y); }, reps), 56.0 * n);
    rep("mix_nt +lds", timeit([&] { csr_mix_x<0, 1, 0><<<g1, B>>>(n, val, col, rp, x, y); }, reps), 56.0 * n);
    rep("mix_nt +gather3", timeit([&] { csr_mix_x<1, 0, 0><<<g1, B>>>(n, val, col, rp, x, y); }, reps), 56.0 * n);
    rep("mix_nt +gather3+lds+rp2", timeit([&] { csr_mix_x<1, 1, 1><<<g1, B>>>(n, val, col, rp, x, y); }, reps), 56.0 * n);
  }
  rep("hipMemcpyDtoD", timeit([&] { CK(hipMemcpyAsync(y, x, n * 8, hipMemcpyDeviceToDevice, 0)); }, reps), 16.0 * n);
  return 0;
}
