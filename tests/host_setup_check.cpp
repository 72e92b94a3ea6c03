// Host-side helpers of the set-up passes (mpp_amd/csrc/runtime.{hpp,cpp}) checked without a GPU: the parallel running sums against
// a serial loop, the task pool, and the vector type that leaves its elements untouched on resize.  Links the library's runtime.o
// (no HIP call is made).  Prints "ok <cases>" or the first mismatch.
#include "../mpp_amd/csrc/runtime.hpp"

#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>

int main() {
  using namespace mpp;
  std::mt19937 rng(12345);
  int cases = 0;
  for (long long n : {0LL, 1LL, 2LL, 63LL, 4096LL, 65535LL, 65536LL, 65537LL, 300001LL, 5000000LL}) {
    std::vector<int> a((size_t)n + 1), ref;
    a[0] = (int)(rng() % 5);
    int mx = 0;
    for (long long i = 0; i < n; ++i) { a[(size_t)i + 1] = (int)(rng() % 9); mx = std::max(mx, a[(size_t)i + 1]); }
    ref = a;
    for (long long i = 0; i < n; ++i) ref[(size_t)i + 1] += ref[(size_t)i];
    const int got = host_prefix_sums(a.data(), n);
    if (a != ref || got != mx) { std::printf("prefix sums differ at n = %lld (max %d vs %d)\n", n, got, mx); return 1; }
    ++cases;
  }
  {   // more than 2^31 - 1 in all: an error, not a wrapped pointer array
    const long long n = 3000;
    std::vector<int> a((size_t)n + 1, 1000000);
    bool threw = false;
    try { host_prefix_sums(a.data(), n); } catch (const Error &) { threw = true; }
    if (!threw) { std::printf("no overflow error\n"); return 1; }
    ++cases;
  }
  {   // every task exactly once, exceptions reach the caller
    std::vector<int> hit(1000, 0);
    host_parallel_tasks((int)hit.size(), [&](int q) { ++hit[(size_t)q]; });
    if (std::accumulate(hit.begin(), hit.end(), 0) != 1000 || *std::min_element(hit.begin(), hit.end()) != 1) { std::printf("tasks\n"); return 1; }
    bool threw = false;
    try { host_parallel_tasks(64, [&](int q) { if (q == 17) throw Error("seventeen"); }); } catch (const Error &e) { threw = std::string(e.what()) == "seventeen"; }
    if (!threw) { std::printf("task exception lost\n"); return 1; }
    ++cases;
  }
  {   // BigVec: resize keeps what was there and does not write the new tail; copies compare equal
    BigVec<int> v;
    v.resize(4);
    for (int i = 0; i < 4; ++i) v[(size_t)i] = i + 1;
    v.resize(1 << 20);
    if (v[0] != 1 || v[3] != 4) { std::printf("BigVec lost its head\n"); return 1; }
    host_parallel_for((long long)v.size(), [&](long long i) { v[(size_t)i] = (int)(i & 1023); });
    const BigVec<int> w = v;
    if (!(w == v) || w[(size_t)123456] != (123456 & 1023)) { std::printf("BigVec copy\n"); return 1; }
    ++cases;
  }
  std::printf("ok %d\n", cases);
  return 0;
}
