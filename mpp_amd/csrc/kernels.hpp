// kernels.hpp -- launch interface of the HIP kernels (gfx950).  Every launcher enqueues on `stream` and returns
// immediately; none allocates or synchronises (so a caller may capture them in a hipGraph).
#pragma once

#include <hip/hip_runtime.h>

#include "physics.hpp"

namespace mpp {

// device-resident scalar block shared by the Krylov/Newton kernels (doubles)
enum ScalarSlot : int {
  S_RHO = 0, S_RHOOLD, S_ALPHA, S_OMEGA, S_OMEGAOLD, S_BETA, S_D1, S_D2, S_DP, S_FLAG, S_ST, S_DP0, S_RRP, S_ERR,
  S_FJP, S_FJS,   // F . (J p), F . (J s) of a BiCGStab iteration whose fused product kernels reduced them (option ls_fold)
  S_SUM0 = 16,  // reduction results land in S_SUM0 .. S_SUM0+63
  S_COUNT = 96
};

constexpr int kRedPairs = 1024;          // 16-byte pairs per reduction block (small problems)
constexpr int kMaxPartialBlocks = 4096;  // reduction grid cap = partial sums per vector
constexpr int kMaxMultiDot = 32;

// physical parameters every assembly kernel receives by value
struct CellArrays {
  const double *P;                                   // pressure the kernel evaluates at
  const double *por, *sat_res, *alpha, *m, *vol;     // soil (internal ordering)
  const double *sbc_pu, *sbc_ps, *sbc_b2, *sbc_b3;   // smoothed Brooks-Corey only (else nullptr)
  const signed char *active;                         // nullptr = every cell active
  const signed char *satf;                           // per-cell saturation function (SatFunc ids) when the mesh mixes them, else nullptr
  const double *por_pref, *por_slope;                // POROSITY_LINEAR: pressure_reference, lin_mod_slope per cell; nullptr = constant model
  int sat_func;                                      // the mesh's saturation function when satf == nullptr
  int all_vg_half;                                   // every cell is van Genuchten with m == 1/2 (host-verified)
  double frac_liq;                                   // frac_liq_sat of boundary auxvars (1.0) and of every cell when ...
  const double *frac_cell;                           // ... no per-cell values were set (VAR_FRAC_LIQ_SAT), else per cell
  EosParams eos;
  const double *powtab;                              // the tables of fpow (fastpow.hpp) in device memory
};

struct ConnArrays {   // internal connections, precomputed geometry
  const double *Dq, *upw, *dgrav, *area;
};

// ---- column-structured (vertical-only) fused assembly: one thread marches one soil column segment ----
// Second stage of a reduction folded into the kernel that produces the partial sums (option fold_finalize; single-pass sizes only:
// nb <= kFoldMaxPartials partial sums per vector): every block takes a ticket after its partial sums are acknowledged, the block
// whose ticket comes last sums them in k_finalize's order (the same bits), runs the scalar step and publishes -- one launch less
// per reduction.  counter == nullptr: not folded (launch_finalize follows).
constexpr int kFoldMaxPartials = 4096;
struct Fold {
  unsigned *counter = nullptr;   // zero between launches
  double *sums = nullptr;        // results at sums[S_SUM0 + j]
  double *sc = nullptr;
  int step = -1, nvec = 0;
  double *pub_host = nullptr; unsigned long long *pub_seq = nullptr; unsigned long long pub_value = 0;
};
struct AuxOut { double *kr, *dkr, *den, *dden; };
struct ColsArgs {
  CellArrays cell;
  ConnArrays conn;       // vertical connection (k,c)->(k+1,c) stored at the index of cell (k,c)
  const double *DqA, *porvol;       // conn.Dq * conn.area and cell.por * cell.vol, streamed instead of the four arrays
  // byte savers of structured inputs (both optional, nullptr = the per-cell / per-connection arrays are streamed):
  //  * soil classes: when the mesh holds at most 256 distinct (sat_res, alpha, m, smoothing coefficients) tuples, one byte
  //    per cell selects a row of cls_table ([7][ncls]: sat_res, alpha, m, pu, ps, b2, b3), staged in LDS by every block
  //  * per-level geometry: when upwind weight and gravity distance of the connection below level k are the same in every
  //    column (extruded grids), lev_upw[k] / lev_dgrav[k] (k = 1 .. nz-1) replace the two per-connection streams
  const unsigned char *soil_cls = nullptr;
  const double *cls_table = nullptr;
  int ncls = 0;
  const double *lev_upw = nullptr, *lev_dgrav = nullptr;
  const double *lev_porvol = nullptr;   // likewise porosity x volume per level when every column agrees (constant-porosity soils)
  // smoothed Brooks-Corey with per-cell soils, one smoothing kind in the whole mesh (bz2: b2 = 0, bz3: b3 = 0): the march streams
  // p_u and the one non-zero coefficient (sbc_b) and forms p_s = -0.9 / alpha itself: 16 instead of 32 bytes per cell
  const double *sbc_b = nullptr;
  int sbc_b_is_b3 = 0;
  const double *accum_prev, *src;   // src: dense mass-rate source [kmol/s], nullptr if none
  double *F;             // residual out (MODE 0,2)
  double *val;           // CSR values out (MODE 1,2), level-major analytic layout
  const int *rowptr;     // CSR row pointers (level-major rows)
  int ncols, nz, nseg;
  double dtInv;
  // complete-row variant (col_bc != nullptr; nseg is taken as 1): side-array entry of the boundary cell at level 0 /
  // level nz-1 of every column (col_bc[c], col_bc[ncols + c]; -1 none), the boundary flux and diagonal contributions
  // k_bc left there, the column LU factors to write (tl may be nullptr) and the partial sums of F.F, P.P (mode 0, 2)
  const int *col_bc = nullptr;
  const double *side_f = nullptr, *side_d = nullptr;
  // ... or (bc_ptr != nullptr) the boundary entries themselves: the march evaluates the fluxes of its column's end cells in its
  // prologue (k_bc's loop and arithmetic) and no k_bc launch precedes it.  Entry e of condition cell j: bc_ptr[j] <= e < bc_ptr[j+1]
  const int *bc_ptr = nullptr, *bc_type = nullptr;
  const double *bc_Dq = nullptr, *bc_dgrav = nullptr, *bc_area = nullptr, *bc_value = nullptr;
  double *tl = nullptr, *tinvb = nullptr, *tc = nullptr;
  bool store_l = true;   // false: the elimination factors l and the reciprocal pivots are not written (the fused solves redo them from the
                         // rows they read and take g = up / pivot alone; launch_tri_factor rebuilds all three planes on demand)
  double *fwd = nullptr;    // with tl: L^{-1} F as well (F = the residual of this pass, or the one in `F` in mode 1)
  double *partials = nullptr;
  // optional trial point (complete-row variant): the state is trial_x + trial_a * trial_y, written to trial_w
  const double *trial_x = nullptr, *trial_y = nullptr;
  double trial_a = 0.0;
  double *trial_w = nullptr;
  // 3-D meshes of complete soil columns (round 5; lateral connections on top of the vertical chains): the march does the accumulation
  // term and the vertical fluxes of a column as on a vertical-only grid, writes the column's tridiagonal at the CSR slots of the
  // 7-point rows (slot_lo / slot_di / slot_up, -1 where a row has no such entry) and leaves the auxvars its lateral pass needs
  // (aux: kr, d kr, density, d density per cell); the lateral connections are added row by row afterwards
  // (launch_assemble_cols_lateral).  Never together with col_bc.
  const int *slot_lo = nullptr, *slot_di = nullptr, *slot_up = nullptr;
  AuxOut aux = {nullptr, nullptr, nullptr, nullptr};
  // ... or (tri3 != nullptr) the tridiagonal goes, coalesced, into three arrays of n entries -- lower at [i], diagonal at [n + i], upper
  // at [2 n + i] -- and the lateral pass, which holds the row's other four entries, writes the WHOLE CSR row: one kernel fills every
  // line of the values instead of two kernels each filling part of it
  double *tri3 = nullptr;
};
// mode 0 residual, 1 Jacobian, 2 both
void launch_assemble_cols(int mode, const ColsArgs &a, hipStream_t stream);
// accumulation term only (PreSolve): out[i] = por*den*sat*vol/dt
void launch_accum(const CellArrays &cell, int n, double dtInv, double *out, hipStream_t stream);

// ---- general (any connection list) assembly: auxvar pass + row-gather pass ----
struct GenArgs {
  CellArrays cell;
  ConnArrays conn;
  const int *cup, *cdn;                 // connection end cells
  const int *rc_ptr, *rc_conn, *rc_slot; // per-row incident connections ((conn<<1)|side), off-diagonal slot
  // the same lists in ELL form, [k*n + row] (enc = (conn<<1)|side or -1, the other end cell, the off-diagonal slot);
  // nullptr when a row has more than 8 incident connections or two connections share a matrix entry
  const int *ell_enc = nullptr, *ell_o = nullptr, *ell_slot = nullptr;
  int ell_k = 0;
  const int *diag_slot, *rowptr;
  const double *accum_prev, *src;
  AuxOut aux;
  double *F, *val;
  int n;
  double dtInv;
  // lateral pass of a 3-D mesh of complete columns (launch_assemble_cols_lateral) with the march's tridiagonal in three arrays
  // (ColsArgs::tri3): the row's lower / upper chain slots (-1: none); the pass writes the whole row
  const double *tri3 = nullptr;
  const int *slot_lo = nullptr, *slot_up = nullptr;
};
void launch_assemble_general(int mode, const GenArgs &a, hipStream_t stream);
// the lateral half of the assembly of a 3-D mesh of complete soil columns: after launch_assemble_cols with ColsArgs::slot_di / aux set,
// every row adds the fluxes of its lateral connections (the ELL incidence lists, entries of chain links skipped: their connection
// index is below n) to F and to its diagonal, writes its off-column entries, and subtracts the source term
void launch_assemble_cols_lateral(int mode, const GenArgs &a, hipStream_t stream);
// set-up: the lateral entries of the K-wide ELL lists compacted into KL-wide ones ([j*n + row], -1 padded); *overflow = 1 if a row has more
void launch_ell_lateral(int n, int K, const int *enc, const int *oth, const int *slot, int KL, int *lenc, int *loth, int *lslot, int *overflow,
                        hipStream_t stream);

// ---- boundary conditions: one thread per cell that carries at least one BC connection ----
struct BcArgs {
  CellArrays cell;
  int ncond_cells;
  const int *cc_cell, *cc_ptr;          // cells with BCs; their BC entries cc_ptr[j]..cc_ptr[j+1]
  const double *Dq, *dgrav, *area, *value;
  const int *type;                      // cond_type | (swap_order << 16)
  const int *diag_slot;
  double *F, *val;
  double *flux_out;                     // optional per-BC-connection flux [kg/s] (mode 4)
  double *side_f = nullptr, *side_d = nullptr;   // when set: per-condition-cell sums go here instead of into F / val
  const double *trial_x = nullptr, *trial_y = nullptr;   // when set: the state is trial_x + trial_a * trial_y, not cell.P
  double trial_a = 0.0;
};
void launch_bc(int mode, const BcArgs &a, hipStream_t stream);

// ---- down-regulated potential mass sinks (COND_DOWNREG_MASS_RATE_CAMPBELL / _FETCH2, GoveqnRichardsODEPressureType.F90
// :1900-1927 residual, :2158-2188 Jacobian): one thread per cell that carries such sinks ----
struct SinkArgs {
  const double *P;                       // pressure the sinks are evaluated at
  const signed char *active;
  int ncells;
  const int *cell, *ptr;                 // cells with sinks; their entries ptr[j]..ptr[j+1]
  const int *type;                       // 512 Campbell | 513 FETCH2
  const double *value, *pc, *expo;       // potential rate [kg/s], pot_mass_sink_pressure, pot_mass_sink_exponent
  const int *diag_slot;
  double *F, *val;
  double *flux_out;                      // mode 4: actual rate [kg/s] per entry
};
void launch_sinks(int mode, const SinkArgs &a, hipStream_t stream);   // mode 0 residual, 1 Jacobian, 2 both, 4 flux output

// internal-connection mass flux [kg/s] at the pressure in cell.P (VAR_MASS_FLUX, AUXVAR_CONN_INTERNAL)
void launch_conn_flux(const CellArrays &cell, const ConnArrays &conn, const int *cup, const int *cdn, int nc,
                      double *flux_out, hipStream_t stream);
// per-cell outputs: var = 608 sat, 610 mass, 611 soil matrix potential
void launch_cell_var(const CellArrays &cell, int n, int var_type, double *out, hipStream_t stream);

// ---- sparse linear algebra ----
// the 32-bit CSR kernels: variant 0 lane-per-row (any row length), 1 block CSR-stream (small, cache-resident matrices),
// 17 software-pipelined wave CSR-stream (rows of <= 8 entries, else variant 0); the solver's default at scale is the compact
// index below (20 / 21) or its windowed form (22 / 23).  Every variant sums a row in entry order.
void launch_spmv(int variant, int n, int max_row_nnz, const int *rowptr, const int *col, const double *val,
                 const double *x, double *y, hipStream_t stream);
// the pipelined kernel on the compact index (16-bit in-tile row offsets + 16-bit column offsets from the tile's first
// row, int32 tile pointers): needs max_row_nnz <= 8; `tpw` tiles per wave
// split > 0 (field-blocked systems): entries are (offset << 1) | block, offsets from the tile's first cell in that block
void launch_spmv_compact(int tpw, int n, int max_row_nnz, int split, const int *tileptr, const unsigned short *roff,
                         const short *dcol, const double *val, const double *x, double *y, hipStream_t stream);
// the compact index with up to 8 window bases per tile (tbase[8*tile + w]) and entries (window << 13) | offset: for patterns
// whose columns lie further than 2^15 rows from their tile (3-D grids)
void launch_spmv_multibase(int tpw, int n, int max_row_nnz, const int *tileptr, const unsigned short *roff,
                           const unsigned short *mcol, const int *tbase, const double *val, const double *x, double *y,
                           hipStream_t stream);
// x.x, f.(A x) and optionally xx.xx without writing A x: partial sums in partials[j*nb + b], nb = the return value
// (= spmv_compact_dots_blocks, which sizes the buffer: 3*nb doubles); finish with launch_finalize
int spmv_compact_dots_blocks(int tpw, int n);
// the same sums on a vertical-only column grid from the values at their computed places (no index): partials[j * blocks + b] = x.x,
// f.(A x), q.q (q may be null: two sums); returns the blocks (= spmv_cols_dots_blocks(n))
int spmv_cols_dots_blocks(int n);
int launch_spmv_cols_dots(int n, int ncols, int nz, const double *val, const double *x, const double *f, const double *q, double *partials,
                          hipStream_t stream);
int launch_spmv_compact_dots(int tpw, int n, int max_row_nnz, int split, const int *tileptr, const unsigned short *roff,
                             const short *dcol, const double *val, const double *x, const double *f, const double *xx,
                             double *partials, hipStream_t stream);
// per-column tridiagonal (block-Jacobi): factor from CSR slots, then apply
struct TriArgs {
  int ncols, nz;
  const int *tri_lo, *tri_di, *tri_up;  // CSR slots of the (k-1), k, (k+1) entries of row (k,c); -1 when absent
  const double *val;
  double *l, *invb, *c;                 // factors, level-major: l_k = lo_k / pivot_{k-1}, 1 / pivot_k, g_k = up_k / pivot_k
};
void launch_tri_factor(const TriArgs &a, hipStream_t stream);
// The same per-column LU on a mesh that is NOT in the tiled column layout -- a rank's piece of a split 3-D mesh: its ghost cells stay
// last, so its cells keep the caller's order -- with the columns given as linked chains of owned cells: start[c] = first (lowest) cell
// of chain c, nxt[i] / prv[i] = the cell above / below cell i in its chain (-1 at the ends).  One thread per chain; the factors go where
// TriArgs puts them (per cell).  A cell without a vertical neighbour is a chain of one (its block is its diagonal entry): ghost cells
// (identity rows) are such chains.  Round 5: this is what keeps `pc_type bjacobi` the SAME operator on every rank count.
struct ColChainArgs {
  int nchains;
  const int *start, *nxt, *prv;
  TriArgs t;
};
void launch_chain_tri_factor(const ColChainArgs &a, hipStream_t stream);
void launch_chain_tri_solve(const ColChainArgs &a, const double *r, double *z, hipStream_t stream);
void launch_tri_solve(const TriArgs &a, const double *r, double *z, hipStream_t stream);
// the same solve with fused dot products of the result: z.z (w == nullptr) or z.w and z.z; returns the number of partial
// blocks (<= kMaxPartialBlocks is the caller's responsibility); finish with launch_finalize
int launch_tri_solve_dots(const TriArgs &a, const double *r, double *z, const double *w, double *partials, hipStream_t stream);
// the same with L^{-1} r parked in LDS between the sweeps (one wave per 64-column tile, nz <= kTriLdsMaxLevels; `group` levels
// are kept in flight per stream; scratch64: 64 doubles the idle lanes of a ragged last tile may write); one partial sum per
// tile: tri_solve_lds_blocks(ncols)
constexpr int kTriLdsMaxLevels = 120;   // 120 x 512 B = 60 KB of LDS per wave
int tri_solve_lds_blocks(int ncols);
int launch_tri_solve_lds(const TriArgs &a, const double *r, double *z, const double *w, double *partials, double *scratch64,
                         int group, hipStream_t stream);
// z = M^{-1} (A x) in one pass on a vertical-only grid (A = the CSR values of the column tridiagonals at their computed slots, a.val;
// row sums as the CSR SpMV kernels form them: same bits as launch_spmv_compact followed by launch_tri_solve_lds), with the same
// partial sums; nz >= 2
int launch_spmv_tri_lds(const TriArgs &a, const double *x, double *z, const double *w, double *partials, double *scratch64, hipStream_t stream,
                        const Fold &fold = Fold());
// the same with x = U^{-1} d produced first (d: a right-hand side the assembly pass forward-eliminated; x is written): the start of
// a BiCGStab solve in one kernel.  Partial sums, one per 64-column tile: (z,x), (z,z), (x,x) in rows 0, 1, 2 of `partials`.
// fdot != nullptr: also the partial sums of fdot . (A x) in row 3 of `partials` (the line search's F . (J p), see k_tri_solve_lds FDOT)
int launch_back_spmv_tri_lds(const TriArgs &a, const double *d, double *x, double *z, double *partials, double *scratch64, hipStream_t stream,
                             const Fold &fold = Fold(), const double *fdot = nullptr);
// ... and with s = r - sc[S_ALPHA] v formed on the way (written to `s`): z = M^{-1} A s with (z,s), (z,z) -- BiCGStab's VecWAXPY, second
// product and omega's two dot products in one kernel
// fdot != nullptr: also the partial sums of fdot . (A s) in row 2 of `partials`
int launch_waxpy_spmv_tri_lds(const TriArgs &a, const double *r, const double *v, const double *sc, double *s, double *z, double *partials,
                              double *scratch64, hipStream_t stream, const Fold &fold = Fold(), const double *fdot = nullptr);
// z = U^{-1} d for a right-hand side the assembly pass already forward-eliminated, with the partial sums of z.z
int launch_tri_back_dots(const TriArgs &a, const double *d, double *z, double *partials, hipStream_t stream);
void launch_jacobi_setup(int n, const int *diag_slot, const double *val, double *invd, hipStream_t stream);

// ---- ILU(0), level-scheduled (PETSc's default PC for SeqAIJ; elimination order = the caller's natural ordering) ----
// Level-ordered triangular solves.  The factors are repacked (after every numeric factorisation) into two ELL blocks
// whose rows are sorted by dependency level -- `n` positions, entry k of position p at [k*n + p], columns are
// positions of the same ordering, padding entries have value 0 and point at the row itself -- so a level is one
// contiguous, fully coalesced range and its dependencies sit in the ranges just before it.  In place on y / x.
// numeric ILU(0) of one level from a precomputed elimination program (level-ordered): for position p the words
// prog[ptr[p] .. ptr[p+1]) are, per L entry in elimination order, {slot(i,k), k, m, (slot(k,j), slot(i,j)) x m}
void launch_ilu_factor_program(const long long *ptr, const int *prog, const int *rows, const int *diag_slot, int lo, int hi,
                               double *lu, double *idiag, hipStream_t stream);
void launch_ilu_repack(long long total, const int *slot, const double *lu, double *out, hipStream_t stream);  // out[e] = slot[e] >= 0 ? lu[slot[e]] : 0
// The same solves as ONE launch per triangle: a tile is <= 256 consecutive rows of one level; its workgroup waits (spinning
// on per-tile flags) only for the tiles that hold its rows' dependencies, so the tail of a level overlaps the head of the
// next and no per-level launch gap is paid.  With `ticket` set, workgroups take their tiles by ticket (see k_ell_chain), so
// the lowest unfinished tile always belongs to a running workgroup whatever the dispatch order; with ticket == nullptr
// (option ilu_chain_tickets = 0) tile = blockIdx.x and progress relies on the hardware dispatching workgroups in index
// order -- observed, not promised by HIP; the whole mode is therefore opt-in (option ilu_chain).  A wait that exceeds its
// spin budget sets sc[S_ERR]; the host checks it behind every chained application and redoes that application per level.
struct ChainArgs {
  int n, K;                        // rows; ELL width
  const int *col;                  // [K][n] positions (in this triangle's level order) of the dependencies; self = padding
  const double *val;               // [K][n]
  const double *idiag;             // upper triangle: 1/u_pp; nullptr for the (unit) lower triangle
  const int *tile_lo, *tile_hi;    // rows [tile_lo[t], tile_hi[t]) of tile t
  const int *dep;                  // [dmax][ntiles] tiles to wait for (-1 none)
  int ntiles, dmax;
  unsigned *flags;                 // flags[t] == epoch: tile t is done in this application
  unsigned epoch;
  double *v;                       // solved in place
  double *sc;                      // sc[S_ERR] receives 1.0 on a timed-out wait
  unsigned max_spins;              // polls (~0.4 us apart) before a wait gives up
  unsigned *ticket;                // monotonic ticket counter of this chain (never reset)
  unsigned ticket_base;            // tickets handed out by earlier launches of this chain (mod 2^32)
  int tiles_per_block;             // consecutive tiles a workgroup owns per ticket
};
void launch_ell_chain(const ChainArgs &a, hipStream_t stream);
void launch_ell_lower_level(int n, int K, const int *col, const double *val, int lo, int hi, double *y, hipStream_t stream);
void launch_ell_upper_level(int n, int K, const int *col, const double *val, const double *idiag, int lo, int hi, double *x,
                            hipStream_t stream);

// ---- the same ILU(0) solves when the pattern is a structured 7-point (5-, 3-point) grid in the caller's natural ordering ----
// The grid (n1 fastest) is cut into boxes of B1 x B2 x B3 cells, B2 * B3 = 64.  A box is solved by one wave: lane (j', k') owns
// the box's line in i' and at step h works on cell i' = h - j' - k', so the three dependencies of a cell are the lane's own
// previous value and two values from neighbour lanes one step earlier -- registers and cross-lane moves instead of memory and
// a kernel boundary per dependency level.  A box depends on its three lower (upper) neighbour boxes only: one launch per box
// level I + J + K, NB1 + NB2 + NB3 - 2 of them instead of n1 + n2 + n3 - 2 cell levels.
// Layout of everything the solves stream through (factor entries by direction, the reciprocal diagonal, the vector): step row
// h = i' + j' + k' of a box, lane l = j' + B2 * k'; rows go in pairs so that a lane's operands of two consecutive steps are one
// 16-byte load: slot = (box * S + (h & ~1)) * 64 + 2 * l + (h & 1), box = I + NB1 * (J + NB2 * K) (box_slot).  S = B1 + B2 + B3 - 2
// rounded up to an even number; slots without a cell are zero.
constexpr int kBoxRing = 12, kBoxRingUpper = 10;   // row pairs of operands a wave keeps in flight: 5 (upper solve: 6) memory
                                                   // instructions each, and the counter holds 63
constexpr int kBoxMaxLen = 128;   // B1 at most (the faces of a box live in LDS: about (B1 + 2 B2) * B2 doubles)
struct BoxGrid {
  int n1, n2, n3;             // extents: caller's id = i + n1 * (j + n2 * k)
  int B1, B2, B3, lg2;        // box extents (B2 = 1 << lg2, B2 * B3 = 64)
  int NB1, NB2, NB3;          // boxes per direction
  int S;                      // step rows per box
};
__host__ __device__ inline size_t box_slot(const BoxGrid &g, long long box, int row, int lane) {
  return ((size_t)box * g.S + (size_t)(row & ~1)) * 64 + 2 * lane + (row & 1);
}
// L = {down, south, west}, U = {east, north, up} entries (ascending column order within a row), Di = 1 / u_ii
// (nown: the cells this rank owns come first; columns from nown on are ghost cells and are skipped)
void launch_box_fill(long long nslots, const BoxGrid &g, const int *cellmap, const int *int2ext, const int *rowptr, const int *col, int nown,
                     const double *lu, const double *idiag, double *const L[3], double *const U[3], double *Di, hipStream_t stream);
// The numeric factorisation by boxes, when every extent of the grid differs from 2 (box_factor_applies): launch_box_load puts the
// matrix into the box layout (L, U, Di = the diagonal, T = the upper entries of the lower neighbours that point back at a cell),
// launch_box_factor then turns L into the multipliers and Di into the reciprocal pivots, one launch per box level in ascending
// order -- the arithmetic and order of the elimination program of the general kernel.
constexpr int kBoxRingFactor = 5;   // row pairs in flight there (11 memory instructions each)
inline bool box_factor_applies(const BoxGrid &g) { return g.n1 != 2 && g.n2 != 2 && g.n3 != 2; }
void launch_box_load(long long nslots, const BoxGrid &g, const int *cellmap, const int *int2ext, const int *rowptr, const int *col, int nown,
                     const double *val, double *const L[3], double *const U[3], double *Di, double *const T[3], hipStream_t stream);
void launch_box_factor(const BoxGrid &g, const int *boxes, int nboxes, double *const L[3], const double *const T[3], double *Di, double *scratch,
                       hipStream_t stream);
void launch_gather_or_zero(long long n, const int *idx, const double *in, double *out, hipStream_t stream);   // out[i] = idx[i] >= 0 ? in[idx[i]] : 0
// r (internal row order, ext2int: the caller's id -> internal row) into the box layout v, padding zero; one workgroup per box with an
// LDS image of the box (box_gather_in_fits: S * 64 doubles within 48 KB; otherwise launch_gather_or_zero through the slot map)
bool box_gather_in_fits(const BoxGrid &g);
void launch_box_gather_in(const BoxGrid &g, const int *ext2int, const double *r, double *v, hipStream_t stream);
void launch_box_gather_out(const BoxGrid &g, const int *ext2int, const double *v, double *z, hipStream_t stream);
// the boxes of one box level (ids in boxes[0 .. nboxes)), in place on the vector in box layout
// scratch: 128 doubles anybody may overwrite
void launch_box_lower(const BoxGrid &g, const int *boxes, int nboxes, const double *const L[3], double *y, double *scratch, hipStream_t stream);
void launch_box_upper(const BoxGrid &g, const int *boxes, int nboxes, const double *const U[3], const double *Di, double *x, double *scratch,
                      hipStream_t stream);
// Both solves with ONE launch per triangle: boxes handed out by ticket, in level order, to `nworkers` resident waves; a box waits for the
// flags of its three predecessor boxes and raises its own (see k_box_lower_queue).  flags: one word per box and triangle, epoch: this
// application's stamp (never 0); ticket: one monotonic counter per triangle, ticket_base = what earlier launches took from it (every
// launch takes nboxes + nworkers tickets).  A wait that exhausts max_spins polls sets sc[S_ERR] = 1: the caller redoes the application
// with one launch per level.
struct BoxQueueArgs {
  const int *order; int nboxes, nworkers;
  unsigned *flags; unsigned epoch; unsigned *ticket; unsigned ticket_base, max_spins;
  double *sc;
};
// Both solves as streams (k_box_lower_stream): a wave owns a pencil (the NB1 boxes of one (J, K)), pencils handed out by ticket in the
// order of J + K, faces handed over 8 steps at a time through face buffers whose initial value (all bits set) is the "not yet" mark.
// faces: 4 x box_stream_face_doubles(g) doubles, every byte 0xFF before the first application (each application leaves them so);
// ticket: two counters (lower, upper), ticket_base what earlier launches took (npencils + nworkers per launch).
bool box_stream_applies(const BoxGrid &g);   // B2 = B3 = 8, S a multiple of 8
size_t box_stream_face_doubles(const BoxGrid &g);
struct BoxStreamArgs {
  const int *order_lower, *order_upper; int npencils, nworkers;
  unsigned *ticket; unsigned ticket_base[2]; unsigned max_spins;
  double *sc, *faces;
};
// r != nullptr: the lower solve takes its right-hand side from r (internal row order) through `cellmap` (internal row of every slot, -1
// padding) instead of from v; z != nullptr: the upper solve writes the solution to z through the same map as well as to v
void launch_box_solve_stream(const BoxGrid &g, const BoxStreamArgs &a, const double *const L[3], const double *const U[3], const double *Di,
                             double *v, double *scratch, hipStream_t stream, const double *r = nullptr, double *z = nullptr,
                             const int *cellmap = nullptr);
void launch_box_solve_queue(const BoxGrid &g, const BoxQueueArgs &lower, const BoxQueueArgs &upper, const double *const L[3],
                            const double *const U[3], const double *Di, double *v, double *scratch, hipStream_t stream);

// ---- soil heat diffusion (thermal KSP path) ----
struct ThermalArgs {
  int n;
  const double *T;                                        // temperature the coefficients are evaluated at (soln_prev)
  const double *liq, *ice, *snow_water, *tuning, *dz, *vol;
  const double *por, *tkmg, *tkdry, *csol;
  const signed char *itype, *shallow, *nsnow_zero, *active;   // nsnow_zero: 1 when num_snow_layer == 0
  double *therm_cond, *heat_cap;                          // auxvar outputs
  // rows
  const int *cup, *cdn, *rc_ptr, *rc_conn, *rc_slot, *diag_slot, *rowptr;
  const double *cdu, *cdd, *carea;
  const double *src;                                      // dense heat-rate source [W] or nullptr
  double *rhs, *val;
  double dt, cnfac;
};
void launch_thermal_aux(const ThermalArgs &a, hipStream_t stream);
void launch_thermal_rows(int mode, const ThermalArgs &a, hipStream_t stream);   // mode 0 rhs, 1 operator, 2 both
struct ThermalBcArgs {
  ThermalArgs t;
  int ncond_cells;
  const int *cc_cell, *cc_ptr, *type;            // cond_type per entry
  const double *du, *dd, *area, *value, *frac, *dhsdT;
  const double *area_dirichlet_rhs = nullptr;    // thermal_reference_defects: the reference's stale `area` per COND_DIRICHLET entry
  bool reference_defects = false;
  const signed char *bc_active;
};
void launch_thermal_bc(int mode, const ThermalBcArgs &a, hipStream_t stream);

// ---- coupled thermal-hydrology: [P(1..N); T(1..N)] unknowns, 2N x 2N CSR, see th_kernels.hip ----
struct ThArgs {
  int n;                                   // cells
  const double *x;                         // state the kernels evaluate at: [P; T], internal cell order
  const double *por, *sat_res, *alpha, *m, *vol;           // soil shared by both equations
  const double *sbc_pu, *sbc_ps, *sbc_b2, *sbc_b3;         // smoothed Brooks-Corey only
  const signed char *active;                                // nullptr = every cell active
  const signed char *satf;                                  // per-cell saturation function (SatFunc ids) when the mesh mixes them, else nullptr
  const double *por_pref, *por_slope;                       // POROSITY_LINEAR in both equations; nullptr = constant model
  const double *tc_wet, *tc_dry, *talpha, *heat_cap, *den_soil;
  int sat_func, density_type, ie_type;
  // auxvars stored by pass 1 for passes 2 and 3
  double *kr, *dkr, *den_m, *ddenP_m, *ddenT_m, *den_e, *ddenP_e, *ddenT_e, *hl, *dhl_dP, *dhl_dT, *tc, *dtc_dP;
  // internal connections
  const int *cup, *cdn, *rc_ptr, *rc_conn, *rc_slot;
  const int *crowptr, *cdiag;              // the mesh (cell) pattern: row pointer and absolute diagonal slots
  const double *Dq_m, *Dq_e, *upw, *dgrav, *area, *cdu, *cdd;
  const double *accum_prev;                // [mass; heat]
  const double *src;                       // [mass-rate kmol/s; heat-rate W] dense, or nullptr
  double *F;                               // [mass; heat]
  double *val;                             // values of the 2N x 2N CSR
  int nnzc;                                // entries of the cell pattern
  double dtInv;
  int ncols, nz, use_cols;                 // vertical-only grids: fused column march instead of passes 1 + 2
  const double *powtab;                    // the tables of fpow (fastpow.hpp) in device memory
};
struct ThBcArgs {
  ThArgs t;
  int ncond_cells;
  const int *cc_cell, *cc_ptr;
  const int *type;                         // cond_type | swap_order << 16 | (1 << 20 for the heat equation)
  const double *Dq, *dgrav, *area, *du, *dd;
  const double *value;                     // condition value (pressure / temperature)
  const double *other;                     // the other field at the boundary auxvar (temperature / pressure)
};
// mode 0 residual, 2 residual + Jacobian, 3 PreSolve (accumulation terms into F)
void launch_th_assemble(int mode, const ThBcArgs &b, hipStream_t stream);

// coupled TH block-Jacobi: one block per soil column = exact LU of the column's block-tridiagonal system (2 x 2 blocks
// {P_k, T_k}, block Thomas), for grids whose internal connections are all vertical
struct ThTriArgs {
  int ncols, nz, n, nnzc;
  const int *crowptr, *tri_lo, *cdiag, *tri_up;   // cell-pattern slots of the (k-1), k, (k+1) entries of cell row (k,c); -1 absent
  const double *val;                               // the 2N x 2N CSR values
  double *L, *inv, *C;                             // 2 x 2 blocks as four planes of n doubles each: [q*n + i], q = pp, pt, tp, tt:
                                                   // L = the lower blocks (a copy), inv = the pivots' inverses, C = inv * upper block
  int stage = 1;                                   // rows of `val` through LDS (option th_stage; 0 for the A/B)
  int inv_fly = 0;                                 // plane inv neither written by the combined factorisation + solve nor read by the fused products
};
void launch_th_tri_factor(const ThTriArgs &a, hipStream_t stream);
// the factorisation and z = M^{-1} r with its dot products (as launch_th_tri_solve_dots) in one sweep; plane L is NOT written
int launch_th_tri_factor_solve_dots(const ThTriArgs &a, const double *r, double *z, const double *w, double *partials, hipStream_t stream);
// y = A x of the coupled system on a vertical-only grid from the values at their computed places (no index; a.L / a.inv / a.C unused)
void launch_th_spmv_cols(const ThTriArgs &a, const double *x, double *y, hipStream_t stream);
// the same product with the line search's sums: partials[j * blocks + b] = x.x, f.(A x), q.q (q may be null: two sums), A x not written;
// returns the blocks (= th_spmv_dots_blocks(n))
int th_spmv_dots_blocks(int n);
int launch_th_spmv_cols_dots(const ThTriArgs &a, const double *x, const double *f, const double *q, double *partials, hipStream_t stream);
// z = M^{-1} (A x) with its dot products (w: z.w and z.z; else z.z), A x never written; partials as launch_th_tri_solve_dots; returns the blocks
// fdot != nullptr (with w): also the partial sums of fdot . (A x) over both fields, row 2 of `partials`
int launch_th_spmv_tri(const ThTriArgs &a, const double *x, double *z, const double *w, double *partials, hipStream_t stream,
                       const double *fdot = nullptr);
void launch_th_tri_solve(const ThTriArgs &a, const double *r, double *z, hipStream_t stream);   // r, z: [P; T]
// ... with the partial sums of z.z (w == nullptr) or of z.w and z.z over both fields, one per block of 256 columns (returns the
// number of blocks; layout partials[j * blocks + block]); finish with launch_finalize
int launch_th_tri_solve_dots(const ThTriArgs &a, const double *r, double *z, const double *w, double *partials, hipStream_t stream);

// ---- vectors (device scalars in `sc`) ----
void launch_copy(int n, const double *x, double *y, hipStream_t stream);
void launch_set(int n, double v, double *y, hipStream_t stream);
void launch_gather(int n, const int *idx, const double *in, double *out, hipStream_t stream);   // out[i] = in[idx[i]]
// out[i] = in[idx[i]] as a flag byte (zero_is_one: out[i] = in[idx[i]] == 0)
void launch_gather_flags(int n, const int *idx, const int *in, signed char *out, int zero_is_one, hipStream_t stream);
// ---- set-up on the device (setup_kernels.hip) ----
// per-row incident-connection lists: ptr (n + 1), rconn = (internal connection << 1 | side), rslot = matrix slot of the other end's
// column (-1: none), roth = the other end (internal cell); inputs: the caller's connection ends (0-based) and the numberings
struct IncidenceArgs {
  int n, nc;
  const int *ext2int, *cup, *cdn, *c2i, *rowptr, *col;
  int *ptr, *rconn, *rslot, *roth, *flags;
};
// column detection: scratch arrays of n ints each (nxt, below, deg_up, deg_dn) and 8 flags, all allocated by the caller
struct ColumnArgs {
  int n, nc;
  const int *cup, *cdn;
  const signed char *is_vert;
  int *nxt, *below, *deg_up, *deg_dn, *flags;
};
struct ColumnLayout { bool columns, vertical_only, identity_perm; int ncols, nz, nc_int; };
void device_detect_columns(const ColumnArgs &a, bool allow_columns, int *ext2int, int *int2ext, int *c2i, ColumnLayout &out, hipStream_t stream);
long long device_inclusive_scan(int *a, long long n, hipStream_t stream);
void device_fill_int(int *a, long long n, int v, hipStream_t stream);
long long device_incidence_pointers(const IncidenceArgs &a, hipStream_t stream);
void device_incidence_fill(const IncidenceArgs &a, int *cursor, hipStream_t stream);
long long device_csr_pointers(const IncidenceArgs &a, int *rowptr, hipStream_t stream);
void device_csr_columns(const IncidenceArgs &a, const int *rowptr, int *col, int *diag, int *max_row, hipStream_t stream);
void device_incidence_slots(const IncidenceArgs &a, hipStream_t stream);
void device_incidence_ell(const IncidenceArgs &a, int *enc, int *oth, int *slt, hipStream_t stream);
// out[idx[i]] = map[in[i]] (set-up: connection end cells into the internal numberings)
void launch_scatter_mapped(int n, const int *idx, const int *map, const int *in, int *out, hipStream_t stream);
void launch_scatter(int n, const int *idx, const double *in, double *out, hipStream_t stream);  // out[idx[i]] = in[i] (idx without repeats)
void launch_pointwise_mult(int n, const double *a, const double *x, double *y, hipStream_t stream);
// w = y + a*x with a = sign * sc[slot] (or the literal `aval` when slot < 0)
void launch_waxpy(int n, double *w, const double *sc, int slot, double sign, double aval, const double *x,
                  const double *y, hipStream_t stream);
// BiCGStab fused updates (PETSc bcgs.c order of operations)
void launch_bcgs_p(int n, double *P, const double *Pold, const double *R, const double *V, const double *sc,
                   hipStream_t stream);
// x <- alpha p + omega s + x (x not read when `first`: zero initial guess), r <- s - omega t; also leaves the
// partial sums of r.r in `partials` and, unless `first`, those of r.rp in its second row (the next iteration's rho);
// returns the number of partial blocks
// wide: one 16-byte pair per lane (bcgs_xr_blocks(n, true) partial sums per vector: finish with
// launch_finalize) instead of the capped reduction grid
int bcgs_xr_blocks(int n, bool wide);
// xx (with first): x.x as a second sum (row 1 of `partials`)
int launch_bcgs_xr(int n, bool first, bool wide, double *X, double *R, const double *P, const double *S, const double *T,
                   const double *RP, const double *sc, double *partials, hipStream_t stream, bool xx = false);
// multi-dot: sums[j] = x . ys[j], j < nvec; two-stage deterministic reduction (partials -> sc[S_SUM0+j])
struct DotArgs { int n, nvec; const double *x[kMaxMultiDot]; const double *y[kMaxMultiDot]; };
// stage 1 only (returns number of partial blocks); stage 2 is launch_finalize
int launch_dots(const DotArgs &a, double *partials, hipStream_t stream);
// Stage 2 of every reduction, one launch: sums[S_SUM0 + j] = sum of partials[j*nblocks .. +nblocks), j < nvec, in an order fixed
// by nblocks alone (chunks of 2048 partials, then the chunk sums; up to 4096 partials in one pass).  The chunk sums go through
// `stage` (kFinalizeStage doubles) and the block that takes the last ticket of `counter` (one word, zero between launches) adds
// them up; step >= 0: that block then runs BiCGStab's scalar recurrence `step` (launch_bcgs_scalars' codes) on `sc`.
constexpr int kFinalizeStage = 8192;
// pub_host != nullptr: the block that finishes also copies the scalar block sc[0 .. S_COUNT) into pinned host memory (device
// pointer pub_host) and then stores pub_value at pub_seq: the host polls that word instead of waiting for the stream.
struct Finalize { double *stage; unsigned *counter; double *sc; int step; double *pub_host = nullptr; unsigned long long *pub_seq = nullptr;
                  unsigned long long pub_value = 0; };
int finalize_chunks(int nblocks);   // per vector; nvec * finalize_chunks(nblocks) <= kFinalizeStage is the caller's responsibility
void launch_finalize(int nblocks, int nvec, const double *partials, double *sums, const Finalize &f, hipStream_t stream);
// y -= sum_j h[j] * v_j  (h on device at sc[S_SUM0+j]); GMRES classical Gram-Schmidt update
struct MaxpyArgs { int n, nvec; const double *v[kMaxMultiDot]; };
// partials (maxpy_blocks(n) doubles + maxpy_blocks(n) / 2048 + 1 more, or nullptr): also y.y of the result;
// returns the number of blocks
int maxpy_blocks(int n);
int launch_maxpy_neg(const MaxpyArgs &a, const double *h, double *y, double *partials, hipStream_t stream);   // h[j]: coefficient of a.v[j]
void launch_scale(int n, double *x, const double *sc, int slot_inv_of, hipStream_t stream);  // x *= 1/sc[slot]
// x += sum_j yv[j]*v_j with yv passed by value
struct MaxpyValArgs { int n, nvec; const double *v[kMaxMultiDot]; double c[kMaxMultiDot]; bool zero_x = false; /* x = sum (x not read) */ };
void launch_maxpy_val(const MaxpyValArgs &a, double *x, hipStream_t stream);
// single-thread scalar updates of the BiCGStab recurrence; step = 0 init, 1 rho/beta, 2 alpha, 3 omega
void launch_bcgs_scalars(int step, double *sc, hipStream_t stream);
// the scalar step `step` (< 0: none) followed by the publication of the scalar block to the host (see Finalize)
void launch_publish(int step, double *sc, double *host, unsigned long long *seq, unsigned long long value, hipStream_t stream);

}  // namespace mpp
