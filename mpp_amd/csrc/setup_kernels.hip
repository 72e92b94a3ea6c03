// Set-up passes that run on the device (once per mesh, mppgpu_setup), from the connection list as the caller gave it:
//  * column detection: the vertical chains of the internal connection set, the tiled level-major cell numbering and the numbering
//    of the connections (Solver::detect_columns is the host route);
//  * the per-row incident-connection lists of the general assembly kernels: a count, a scan, a fill in any order and one short sort
//    per row by the caller's (connection, side);
//  * the CSR pattern derived from them: the diagonal plus the distinct other ends of a row, sorted.
// On the host these were passes of atomic counters over 8 x 10^8 incidences with a sort and binary searches per row: 15 of the
// 25.6 s the 512^3 thermal mesh's mppgpu_setup took on the 16 host threads a GPU box grants; here they take 0.9 s
// (profiles/r04/setup_phase_timing_device.log).  Every array comes out exactly as the host builds it -- nothing depends on the
// order the atomics ran in -- and option setup_verify = 1 makes the library build both and compare (tests/test_gpu_setup.py).
#include "kernels.hpp"
#include "runtime.hpp"

#include <vector>

namespace mpp {
namespace {

constexpr int kBlock = 256;
constexpr int kScanItems = 8;                       // consecutive items per thread of a scan block
constexpr int kScanTile = kBlock * kScanItems;

inline dim3 nblk(long long n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

__global__ __launch_bounds__(kBlock) void k_inc_count(int nc, const int *__restrict__ ext2int, const int *__restrict__ cup,
                                                      const int *__restrict__ cdn, int *__restrict__ cnt) {
  const size_t c = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (c >= (size_t)nc) return;
  atomicAdd(&cnt[(size_t)ext2int[cup[c]] + 1], 1);
  atomicAdd(&cnt[(size_t)ext2int[cdn[c]] + 1], 1);
}

// a[i] <- a[0] + ... + a[i] within tiles of kScanTile items; the tile's total goes to tile_sum
__global__ __launch_bounds__(kBlock) void k_scan_tiles(long long n, int *__restrict__ a, int *__restrict__ tile_sum) {
  __shared__ int part[kBlock];
  const long long base = (long long)blockIdx.x * kScanTile + (long long)threadIdx.x * kScanItems;
  int v[kScanItems], run = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    v[j] = base + j < n ? a[base + j] : 0;
    run += v[j];
    v[j] = run;
  }
  part[threadIdx.x] = run;
  __syncthreads();
  for (int off = 1; off < kBlock; off <<= 1) {   // inclusive scan of the threads' sums
    const int add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  const int before = part[threadIdx.x] - run;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j)
    if (base + j < n) a[base + j] = v[j] + before;
  if (threadIdx.x == kBlock - 1) tile_sum[blockIdx.x] = part[kBlock - 1];
}
__global__ __launch_bounds__(kBlock) void k_scan_add(long long n, int *__restrict__ a, const int *__restrict__ tile_off) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) a[i] += tile_off[i / kScanTile];
}

__global__ __launch_bounds__(kBlock) void k_inc_fill(int nc, const int *__restrict__ ext2int, const int *__restrict__ cup,
                                                     const int *__restrict__ cdn, const int *__restrict__ ptr,
                                                     int *__restrict__ cursor, int *__restrict__ rconn) {
  const size_t c = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (c >= (size_t)nc) return;
  const int u = ext2int[cup[c]], d = ext2int[cdn[c]];
  rconn[ptr[u] + atomicAdd(&cursor[u], 1)] = (int)(c << 1);
  rconn[ptr[d] + atomicAdd(&cursor[d], 1)] = (int)(c << 1) | 1;
}

// one row per thread: its list in the caller's (connection, side) order and the other end of every entry (internal cell)
__global__ __launch_bounds__(kBlock) void k_inc_sort(IncidenceArgs a) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (size_t)a.n) return;
  const int b = a.ptr[i], e = a.ptr[i + 1];
  for (int p = b + 1; p < e; ++p) {   // insertion sort: a handful of entries
    const int v = a.rconn[p];
    int q = p - 1;
    while (q >= b && a.rconn[q] > v) { a.rconn[q + 1] = a.rconn[q]; --q; }
    a.rconn[q + 1] = v;
  }
  for (int p = b; p < e; ++p) {
    const int v = a.rconn[p], c = v >> 1;
    a.roth[p] = (v & 1) ? a.ext2int[a.cup[c]] : a.ext2int[a.cdn[c]];
  }
}
// CSR pattern = the diagonal + the distinct other ends of a row's incidences (GoveqnRichardsODEPressureType.F90:1689, 2051-2069):
// the row's length into cnt[i + 1] ...
__global__ __launch_bounds__(kBlock) void k_csr_count(IncidenceArgs a, int *__restrict__ cnt) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (size_t)a.n) return;
  const int b = a.ptr[i], e = a.ptr[i + 1];
  int m = 1;
  for (int p = b; p < e; ++p) {
    const int o = a.roth[p];
    bool seen = o == (int)i;
    for (int q = b; q < p && !seen; ++q) seen = a.roth[q] == o;
    m += !seen;
  }
  cnt[i + 1] = m;
}
// ... then, with the row pointers known, its sorted columns and the place of the diagonal
__global__ __launch_bounds__(kBlock) void k_csr_write(IncidenceArgs a, const int *__restrict__ rowptr, int *__restrict__ col,
                                                      int *__restrict__ diag, int *__restrict__ max_row) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (size_t)a.n) return;
  const int b = a.ptr[i], e = a.ptr[i + 1], r0 = rowptr[i];
  int m = 0;
  auto insert = [&](int v) {   // into the sorted col[r0 .. r0 + m) unless it is there
    int q = m - 1;
    while (q >= 0 && col[r0 + q] > v) --q;
    if (q >= 0 && col[r0 + q] == v) return;
    for (int t = m - 1; t > q; --t) col[r0 + t + 1] = col[r0 + t];
    col[r0 + q + 1] = v;
    ++m;
  };
  insert((int)i);
  for (int p = b; p < e; ++p) insert(a.roth[p]);
  for (int t = 0; t < m; ++t) if (col[r0 + t] == (int)i) diag[i] = r0 + t;
  atomicMax(max_row, m);
}
// ... and the incidence entries' matrix slots, their final encoding, the longest list, rows that meet a slot twice
__global__ __launch_bounds__(kBlock) void k_inc_slots(IncidenceArgs a) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (size_t)a.n) return;
  const int b = a.ptr[i], e = a.ptr[i + 1];
  const int rb = a.rowptr[i], re = a.rowptr[i + 1];
  bool dup = false;
  for (int p = b; p < e; ++p) {
    const int v = a.rconn[p], c = v >> 1, side = v & 1, oth = a.roth[p];
    int lo = rb, hi = re;   // lower bound of oth in the row's sorted columns
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.col[mid] < oth) lo = mid + 1; else hi = mid; }
    const int slot = (lo < re && a.col[lo] == oth) ? lo : -1;
    for (int q = b; q < p; ++q) dup |= a.rslot[q] == slot;
    a.rconn[p] = (a.c2i[c] << 1) | side;
    a.rslot[p] = slot;
  }
  atomicMax(&a.flags[0], e - b);
  if (dup) a.flags[1] = 1;
}

__global__ __launch_bounds__(kBlock) void k_inc_ell(int n, const int *__restrict__ ptr, const int *__restrict__ rconn,
                                                    const int *__restrict__ rslot, const int *__restrict__ roth, int *__restrict__ enc,
                                                    int *__restrict__ oth, int *__restrict__ slt) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (size_t)n) return;
  for (int e = ptr[i], k = 0; e < ptr[i + 1]; ++e, ++k) {
    const size_t q = (size_t)k * n + i;
    enc[q] = rconn[e]; slt[q] = rslot[e]; oth[q] = roth[e];
  }
}

// ---- column detection (Solver::detect_columns): the vertical chains of the internal connection set, any cell numbering ----
__global__ __launch_bounds__(kBlock) void k_col_links(ColumnArgs a) {
  const size_t c = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (c >= (size_t)a.nc || !a.is_vert[c]) return;
  const int u = a.cup[c], d = a.cdn[c];
  a.nxt[u] = d; a.below[d] = u;
  // a chain has at most ONE link above and one below every cell: counted (two links between the same pair store the same values)
  atomicAdd(&a.deg_up[u], 1); atomicAdd(&a.deg_dn[d], 1);
  atomicAdd(&a.flags[0], 1);
}
__global__ __launch_bounds__(kBlock) void k_col_starts(ColumnArgs a, int *__restrict__ start) {   // start[i + 1] = 1 for the first cell of a chain
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (size_t)a.n) return;
  if (a.deg_up[i] > 1 || a.deg_dn[i] > 1) a.flags[1] = 1;
  start[i + 1] = a.below[i] == -1;
}
__global__ __launch_bounds__(kBlock) void k_col_walk(ColumnArgs a, const int *__restrict__ start, int len, int *__restrict__ lev,
                                                     int *__restrict__ colof) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (size_t)a.n || start[i + 1] == start[i]) return;   // (after the scan: the rank of a start is start[i + 1] - 1)
  const int ch = start[i + 1] - 1;
  int cur = (int)i, k = 0;
  while (cur != -1 && k <= len) { lev[cur] = k; colof[cur] = ch; cur = a.nxt[cur]; ++k; }
  if (k != len) a.flags[2] = 1;
}
__device__ __forceinline__ int cell_index_dev(int lev, int col, int ncols, int nz) {   // Solver::cell_index: tiled level-major
  const int t = col >> 6, tw = min(64, ncols - (t << 6));
  return t * 64 * nz + lev * tw + (col & 63);
}
__global__ __launch_bounds__(kBlock) void k_col_number(ColumnArgs a, int ncols, int nz, const int *__restrict__ lev,
                                                       const int *__restrict__ colof, int *__restrict__ ext2int) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (size_t)a.n) return;
  if (lev[i] < 0) { a.flags[3] = 1; ext2int[i] = 0; return; }
  ext2int[i] = cell_index_dev(lev[i], colof[i], ncols, nz);
}
__global__ __launch_bounds__(kBlock) void k_col_identity(int n, int *__restrict__ ext2int) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < (size_t)n) ext2int[i] = (int)i;
}
__global__ __launch_bounds__(kBlock) void k_col_inverse(ColumnArgs a, const int *__restrict__ ext2int, int *__restrict__ int2ext) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (size_t)a.n) return;
  int2ext[ext2int[i]] = (int)i;
  if (ext2int[i] != (int)i) a.flags[4] = 1;
}
// internal connection numbering: chain links at the index of their lower cell; a lateral connection counts towards its upwind cell
__global__ __launch_bounds__(kBlock) void k_conn_vertical(ColumnArgs a, const int *__restrict__ ext2int, int *__restrict__ c2i,
                                                          int *__restrict__ cnt) {
  const size_t c = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (c >= (size_t)a.nc) return;
  const int u = ext2int[a.cup[c]];
  if (a.is_vert[c]) c2i[c] = u;
  else atomicAdd(&cnt[(size_t)u + 1], 1);
}
__global__ __launch_bounds__(kBlock) void k_conn_lateral_fill(ColumnArgs a, const int *__restrict__ ext2int, const int *__restrict__ ptr,
                                                              int *__restrict__ cursor, int *__restrict__ list) {
  const size_t c = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (c >= (size_t)a.nc || a.is_vert[c]) return;
  const int u = ext2int[a.cup[c]];
  list[ptr[u] + atomicAdd(&cursor[u], 1)] = (int)c;
}
// the j-th lateral connection (in the caller's order) whose upwind cell is u sits at n + j n + u
__global__ __launch_bounds__(kBlock) void k_conn_lateral_number(ColumnArgs a, const int *__restrict__ ptr, int *__restrict__ list,
                                                                int *__restrict__ c2i) {
  const size_t u = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (u >= (size_t)a.n) return;
  const int b = ptr[u], e = ptr[u + 1];
  for (int p = b + 1; p < e; ++p) {
    const int v = list[p];
    int q = p - 1;
    while (q >= b && list[q] > v) { list[q + 1] = list[q]; --q; }
    list[q + 1] = v;
  }
  for (int p = b, j = 0; p < e; ++p, ++j) c2i[list[p]] = a.n + j * a.n + (int)u;
  atomicMax(&a.flags[5], e - b);
}
__global__ __launch_bounds__(kBlock) void k_conn_identity(int nc, int *__restrict__ c2i) {
  const size_t c = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (c < (size_t)nc) c2i[c] = (int)c;
}

__global__ __launch_bounds__(kBlock) void k_fill_int(long long n, int v, int *__restrict__ a) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) a[i] = v;
}

}  // namespace

#define SLAUNCH(kern, grid, ...) hipLaunchKernelGGL(kern, grid, dim3(kBlock), 0, stream, __VA_ARGS__)

// a[i] <- a[0] + ... + a[i] for i in [0, n) on the device; returns the total (a[n - 1] afterwards).  Throws past 2^31 - 1.
long long device_inclusive_scan(int *a, long long n, hipStream_t stream) {
  if (n <= 0) return 0;
  const long long ntile = (n + kScanTile - 1) / kScanTile;
  DevBuf<int> tile;
  tile.alloc((size_t)ntile);
  SLAUNCH(k_scan_tiles, dim3((unsigned)ntile), n, a, tile.p);
  std::vector<int> h((size_t)ntile);
  HIP_CHECK(hipMemcpyAsync(h.data(), tile.p, (size_t)ntile * sizeof(int), hipMemcpyDeviceToHost, stream));
  HIP_CHECK(hipStreamSynchronize(stream));
  long long run = 0;
  for (long long t = 0; t < ntile; ++t) { const long long s = h[(size_t)t]; h[(size_t)t] = (int)run; run += s; if (run > 0x7fffffffLL) throw Error("more than 2^31 - 1 entries: the 32-bit pointer arrays cannot hold the pattern"); }
  HIP_CHECK(hipMemcpyAsync(tile.p, h.data(), (size_t)ntile * sizeof(int), hipMemcpyHostToDevice, stream));
  SLAUNCH(k_scan_add, nblk(n), n, a, tile.p);
  HIP_CHECK(hipStreamSynchronize(stream));   // (h and tile go out of scope)
  return run;
}

void device_fill_int(int *a, long long n, int v, hipStream_t stream) {
  if (n > 0) SLAUNCH(k_fill_int, nblk(n), n, v, a);
}

// a.ptr: n + 1 ints, zeroed by the caller; on return the row pointers.  Returns the number of incidences (= 2 nc); the caller then
// allocates rconn / rslot / roth of that size and calls device_incidence_fill, device_csr_* and device_incidence_slots.
long long device_incidence_pointers(const IncidenceArgs &a, hipStream_t stream) {
  if (a.nc > 0) SLAUNCH(k_inc_count, nblk(a.nc), a.nc, a.ext2int, a.cup, a.cdn, a.ptr);
  return device_inclusive_scan(a.ptr + 1, a.n, stream);
}
// cursor: n ints, zeroed by the caller (scratch).  Leaves rconn as (caller's connection << 1 | side), sorted per row, and roth.
void device_incidence_fill(const IncidenceArgs &a, int *cursor, hipStream_t stream) {
  if (a.nc > 0) SLAUNCH(k_inc_fill, nblk(a.nc), a.nc, a.ext2int, a.cup, a.cdn, a.ptr, cursor, a.rconn);
  if (a.n > 0) SLAUNCH(k_inc_sort, nblk(a.n), a);
}
// the CSR pattern from the incidence lists: rowptr (n + 1 ints, the caller zeroes [0]) gets the row pointers, the return value is nnz;
// then device_csr_columns fills col (nnz), diag (n) and *max_row (zeroed by the caller)
long long device_csr_pointers(const IncidenceArgs &a, int *rowptr, hipStream_t stream) {
  if (a.n > 0) SLAUNCH(k_csr_count, nblk(a.n), a, rowptr);
  return device_inclusive_scan(rowptr + 1, a.n, stream);
}
void device_csr_columns(const IncidenceArgs &a, const int *rowptr, int *col, int *diag, int *max_row, hipStream_t stream) {
  if (a.n > 0) SLAUNCH(k_csr_write, nblk(a.n), a, rowptr, col, diag, max_row);
}
// flags: 2 ints, zeroed: [0] <- the longest list, [1] <- 1 if a row meets a matrix slot twice; a.rowptr / a.col = the CSR pattern
void device_incidence_slots(const IncidenceArgs &a, hipStream_t stream) {
  if (a.n > 0) SLAUNCH(k_inc_slots, nblk(a.n), a);
}
void device_incidence_ell(const IncidenceArgs &a, int *enc, int *oth, int *slt, hipStream_t stream) {
  if (a.n > 0) SLAUNCH(k_inc_ell, nblk(a.n), a.n, a.ptr, a.rconn, a.rslot, a.roth, enc, oth, slt);
}

// Column detection and the two numberings on the device.  In: the caller's connection ends (0-based) and the vertical flags.
// Out: ext2int, int2ext (n ints), c2i (nc ints) on the device, the layout in `out`.  ghost cells (identity layout) are the caller's
// business: pass allow_columns = false.
void device_detect_columns(const ColumnArgs &a, bool allow_columns, int *ext2int, int *int2ext, int *c2i, ColumnLayout &out,
                           hipStream_t stream) {
  const int n = a.n, nc = a.nc;
  device_fill_int(a.nxt, n, -1, stream); device_fill_int(a.below, n, -1, stream);
  HIP_CHECK(hipMemsetAsync(a.deg_up, 0, (size_t)n * sizeof(int), stream));
  HIP_CHECK(hipMemsetAsync(a.deg_dn, 0, (size_t)n * sizeof(int), stream));
  HIP_CHECK(hipMemsetAsync(a.flags, 0, 8 * sizeof(int), stream));
  if (nc > 0) SLAUNCH(k_col_links, nblk(nc), a);
  DevBuf<int> start, lev, colof;
  start.alloc((size_t)n + 1); start.zero(stream);
  SLAUNCH(k_col_starts, nblk(n), a, start.p);
  const int nchains = (int)device_inclusive_scan(start.p + 1, n, stream);
  int hf[8];
  auto flags = [&] {
    HIP_CHECK(hipMemcpyAsync(hf, a.flags, sizeof hf, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
  };
  flags();
  const long long nvert = hf[0];
  bool columns = false;
  out.ncols = n; out.nz = 1;
  if (allow_columns && hf[1] == 0 && nchains > 0 && n % nchains == 0) {
    const int len = n / nchains;
    lev.alloc((size_t)n); colof.alloc((size_t)n);
    device_fill_int(lev.p, n, -1, stream); device_fill_int(colof.p, n, -1, stream);
    SLAUNCH(k_col_walk, nblk(n), a, start.p, len, lev.p, colof.p);
    flags();
    if (hf[2] == 0) {
      SLAUNCH(k_col_number, nblk(n), a, nchains, len, lev.p, colof.p, ext2int);
      flags();
      if (hf[3] == 0) { columns = true; out.ncols = nchains; out.nz = len; }
    }
  }
  if (!columns) SLAUNCH(k_col_identity, nblk(n), n, ext2int);
  SLAUNCH(k_col_inverse, nblk(n), a, ext2int, int2ext);
  out.columns = columns;
  if (columns && nvert == nc) {
    if (nc > 0) SLAUNCH(k_conn_vertical, nblk(nc), a, ext2int, c2i, start.p);   // (no lateral connection: the counters stay untouched)
    out.nc_int = n; out.vertical_only = true;
  } else if (columns) {
    start.zero(stream);   // reused: lateral connections per upwind cell
    SLAUNCH(k_conn_vertical, nblk(nc), a, ext2int, c2i, start.p);
    const long long nlat = device_inclusive_scan(start.p + 1, n, stream);
    int jmax = 0;
    if (nlat > 0) {
      DevBuf<int> cursor, list;
      cursor.alloc((size_t)n); cursor.zero(stream);
      list.alloc((size_t)nlat);
      SLAUNCH(k_conn_lateral_fill, nblk(nc), a, ext2int, start.p, cursor.p, list.p);
      SLAUNCH(k_conn_lateral_number, nblk(n), a, start.p, list.p, c2i);
      flags();
      jmax = hf[5];
      if ((long long)n * (jmax + 1) > 0x3fffffffLL) throw Error("too many lateral connections per cell for 32-bit connection ids");
    }
    out.nc_int = n + jmax * n; out.vertical_only = nlat == 0;
  } else {
    if (nc > 0) SLAUNCH(k_conn_identity, nblk(nc), nc, c2i);
    out.nc_int = nc; out.vertical_only = false;
  }
  flags();
  out.identity_perm = hf[4] == 0;
}

}  // namespace mpp
