// solver.cpp -- problem layout, Newton / Krylov drivers and the C ABI (include/mppgpu.h).
//
// What it stands in for (reference file:line):
//   mpp%SetupProblem / MPPSetupProblemSNES     src/mpp/mpp/MultiPhysicsProbBaseType.F90:1089-1213  (DM/Mat/Vec creation)
//   soe%StepDT -> SOEBaseStepDT_SNES           src/mpp/soe/SystemOfEquationsBaseType.F90:368-552
//   VSFMSOEPreSolve / PostSolve / Residual / Jacobian   src/mpp/soe/SystemOfEquationsVSFMType.F90:94-403, 506-660
//   PETSc v3.16.2 SNESSolve_NEWTONLS, SNESLineSearchApply_BT, KSPSolve_BCGS, KSPSolve_GMRES, PCApply (bjacobi)
//     (third-party; call sites SystemOfEquationsBaseType.F90:478, MultiPhysicsProbBaseType.F90:1193-1205)
// All arithmetic on vectors and matrices happens in HIP kernels (kernels.hip); the host only sequences launches
// and reads back a handful of scalars per iteration.  There is no CPU compute path.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <numeric>
#include <tuple>

#include "../../include/mppgpu.h"
#include "runtime.hpp"
#include "th_physics.hpp"

namespace mpp {

namespace {
int map_sat_func(int reference_id) {
  switch (reference_id) {
    case MPPGPU_SAT_FUNC_VAN_GENUCHTEN: return kSatVG;
    case MPPGPU_SAT_FUNC_BROOKS_COREY: return kSatBC;
    case MPPGPU_SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ2:
    case MPPGPU_SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ3: return kSatSBC;
    default: return -1;
  }
}

// findGu_SBC_zeroCoeff (src/mpp/util/SaturationFunction.F90:425-518): set-up time only, host
double find_gu_sbc_zero_coeff(double lambda, int AA, double gs) {
  const double relTol = 1e-12;
  double gu = std::pow((double)AA / ((double)AA + lambda), -1.0 / lambda);
  if (gs > 0.0) {
    double guLeft = 1.0, guRight = gu;
    for (int guard = 0; guard < 200; ++guard) {
      if (gu <= guLeft || gu >= guRight) gu = guLeft + 0.5 * (guRight - guLeft);
      double guInv = 1.0 / gu, guToMinusLam = std::pow(gu, -lambda), gsOnGu = gs * guInv;
      double resid = AA - guToMinusLam * (AA + lambda - lambda * gsOnGu);
      if (resid < 0.0) guLeft = gu; else guRight = gu;
      double dr = (1.0 + lambda) * (1.0 - gsOnGu) + (AA - 1);
      dr = lambda * guToMinusLam * guInv * dr;
      double delta = resid / dr;
      gu = gu - delta;
      if (std::fabs(delta) < relTol * std::fabs(gu)) break;
    }
  }
  return gu;
}
}  // namespace

struct Solver {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  bool setup_done = false;
  Profiler prof;
  Comm comm;
  Options opt;

  // ---------------- staging, caller's ordering, 0-based ----------------
  int n = 0, nlev = 0;
  int nrow = 0;   // unknowns = rows of the linear system: n, or 2n for the coupled thermal-hydrology system [P; T]
  // ---------------- ghost cells + halo exchange (3-D meshes split across ranks) ----------------
  // The LAST nghost cells of the mesh are copies of cells another rank owns.  They are cells like any other for the
  // assembly kernels (auxvars, fluxes with their neighbours) but their own rows are identity rows; vector operations,
  // reductions and SpMV rows run over the first nred = nrow - nghost entries only, and the ghost tail of a vector is
  // (a) refreshed from the owners before it is read as a state or as an SpMV operand, (b) zeroed before a preconditioner
  // apply -- which makes the local ILU(0) exactly the block-Jacobi/ILU(0) PETSc applies in parallel.
  // Coupled thermal-hydrology: the unknowns are field-blocked [P(1..n); T(1..n)], so the ghost rows are two ranges, not a tail.
  // There nred = nrow and the ghost entries of EVERY vector are zero whenever a vector operation or a reduction can see them:
  // halo_exchange() fills them right before an assembly pass or an SpMV reads them, ghost_zero() clears them right after
  // (the operand's, and the identity rows' outputs); residual rows of ghost cells are zero by construction (th_kernels.hip).
  int nghost = 0, nred = 0;
  struct HaloPlan {
    std::vector<int> peer, send_ptr, send_cells, recv_ptr;   // per neighbour: send list (owned cells, 0-based), ghost range
    DevBuf<int> d_send_idx;
    DevBuf<double> sendbuf;
    bool set = false;
  } halo;
  BigVec<double> vol;
  BigVec<int> active;
  bool any_inactive = false;
  int nc = 0;
  BigVec<int> cup_e, cdn_e;
  BigVec<signed char> cvert;   // 1: the connection is (mostly) vertical -- a chain link of the column layout
  BigVec<double> cdu, cdd, carea, cuv;
  std::vector<Condition> bcs, sss;
  // mesh%lateral_conn_set_list (operator split: RichardsODEComputeLateralFlux, GE:2615-2743)
  int nlat = 0;
  std::vector<int> lat_up_e, lat_dn_e;
  std::vector<double> lat_du, lat_dd, lat_area, lat_uv;
  DevBuf<int> d_lat_up, d_lat_dn;
  DevBuf<double> d_lat_Dq, d_lat_upw, d_lat_dgrav, d_lat_area, d_lat_flux;
  std::vector<double> permx, permy, permz, por, alpha, lam, sat_res;
  std::vector<int> satf;
  int sat_func = -1;
  int sbc_kind = 0;   // smoothed Brooks-Corey cells of the mesh: 2 all bz2, 3 all bz3, 0 none or both kinds
  bool mixed_satf = false;     // cells of one mesh use different saturation functions (satParams are per cell in the reference)
  bool all_vg_half = false;
  std::vector<double> por_pref, por_slope;   // POROSITY_LINEAR (PorosityFunctionMod.F90:75-96); empty = constant model
  DevBuf<signed char> d_satf;
  // byte savers of the column march (ColsArgs): soil classes and per-level connection geometry
  DevBuf<unsigned char> d_soil_cls;
  DevBuf<double> d_powtab;
  DevBuf<double> d_cls_table, d_lev_upw, d_lev_dgrav, d_lev_porvol;
  int ncls = 0;
  bool lev_geo = false, lev_pv = false;
  DevBuf<double> d_por_pref, d_por_slope;
  int density_type = MPPGPU_DENSITY_CONSTANT;  // RichODEPressureAuxVarInit default (AuxType :120)
  double temperature = 273.15 + 25.0, frac_liq = 1.0;
  bool soil_dirty = true;

  // ---------------- layout ----------------
  bool columns = false, vertical_only = false, identity_perm = true;
  int ncols = 0, nz = 0;
  BigVec<int> ext2int, int2ext, cext2int;
  int nc_int = 0;
  BigVec<int> h_rowptr, h_col, h_diag;
  int nnz = 0, max_row_nnz = 0;
  int nb = 0, ns = 0;
  std::vector<int> bc_pos_of_g, h_bc_cell_int;
  int ncond_cells = 0;

  // ---------------- device ----------------
  DevBuf<double> d_vol, d_por, d_sat_res, d_alpha, d_m, d_pu, d_ps, d_b2, d_b3;
  DevBuf<signed char> d_active;
  DevBuf<double> d_Dq, d_upw, d_dgrav, d_area;
  DevBuf<int> d_cup, d_cdn;
  DevBuf<int> d_rowptr, d_col, d_diag, d_rc_ptr, d_rc_conn, d_rc_slot, d_tri_lo, d_tri_up, d_ell_enc, d_ell_o, d_ell_slot;
  int ell_k = 0;   // incident connections per row in the ELL copy (0: none)
  // a rank's piece of a split 3-D mesh (ghost cells: identity layout): the soil columns of its owned cells as linked chains, for the
  // per-column block-Jacobi preconditioner (ColChainArgs); gc_n = 0: none
  DevBuf<int> d_gc_start, d_gc_nxt, d_gc_prv;
  int gc_n = 0;
  std::vector<int> h_gc_nxt, h_gc_prv;
  DevBuf<double> d_tri3;   // the march's tridiagonal of a 3-D mesh of complete columns, three arrays of n (ColsArgs::tri3)
  DevBuf<int> d_lat_enc, d_lat_o, d_lat_slot;   // the lateral entries of those lists, four wide (3-D meshes of complete columns)
  int lat_k = 0;
  DevBuf<double> d_val;
  // compact SpMV index (see k_spmv_pipe_c): built when the pattern allows it
  DevBuf<int> d_tileptr;
  DevBuf<unsigned short> d_roff;
  DevBuf<short> d_dcol;
  DevBuf<unsigned short> d_mcol;   // multi-base entries (window << 13) | offset
  DevBuf<int> d_tbase;             // 8 window bases per 64-row tile
  bool multibase_index = false;    // variants 22/23 available (built when the single-base compact index is not)
  bool compact_index = false;
  int compact_split = 0;   // > 0: the compact index of a field-blocked system (entries carry their block, offsets are per block)
  DevBuf<int> d_cc_cell, d_cc_ptr, d_bc_type;
  DevBuf<double> d_bc_Dq, d_bc_dgrav, d_bc_area, d_bc_value, d_bc_flux;
  DevBuf<double> d_src, d_frac;
  std::vector<double> h_frac;             // per-cell frac_liq_sat, caller's ordering (empty = 1 everywhere)
  // exchanged-mass bookkeeping of the ELM coupling (GE:33-34, 1150-1166, 1338-1342); option "track_mass_exchange"
  std::vector<double> lat_mass_exc, bnd_mass_exc;   // per cell (caller's ordering) / per boundary connection (offset numbering)
  std::vector<char> conn_horizontal;
  bool has_clm_state = false;
  bool has_src = false;
  // down-regulated sinks, grouped by cell
  int ndr_cells = 0, ndr = 0;
  std::vector<int> dr_pos_of_g;          // position of (condition offset + i) in the grouped arrays, -1 if not down-regulated
  std::vector<double> h_dr_value, h_dr_pc, h_dr_n;
  DevBuf<int> d_dr_cell, d_dr_ptr, d_dr_type;
  DevBuf<double> d_dr_value, d_dr_pc, d_dr_n, d_dr_flux;
  DevBuf<int> d_ext2int, d_int2ext;
  DevBuf<double> soln, soln_prev, accum_prev, F, Y, W, G, R, RP, V, T, S, Pv, Wk, tmp_ext, tmp_int;
  DevBuf<double> aux_kr, aux_dkr, aux_den, aux_dden;
  DevBuf<double> tri_l, tri_invb, tri_c, jac_invd;
  DevBuf<double> partials, sc, spmv_partials, col_partials, xr_partials, tri_scratch, fin_stage;
  DevBuf<unsigned> fin_counter;   // the ticket of the reductions' second stage (zero between launches)
  // complete-row column march (plan_complete_rows): boundary cells sit on column ends only
  bool rows_plan = false, asm_sums = false;
  struct Trial { const double *x = nullptr, *y = nullptr; double a = 0.0; } ls_point;
  // The line search's first reductions (||Y||, F.(J Y), ||X||) issued by the Krylov solver itself, behind the update of what
  // is expected to be its last iteration: one read-back and one all-reduce serve both.  Armed when the previous solve of
  // this handle converged in one iteration (the column-structured problems); a solve that goes on discards them.
  // spec: the line search's first trial point (the full step W = X - Y) is evaluated there as well -- boundary kernel, column
  // march (with the Jacobian when the search would have asked for it), ||F(W)||^2 and ||W||^2 in the same read-back and the
  // same all-reduce: a Newton iteration whose Krylov solve takes one iteration and whose full step is accepted costs the host
  // ONE wait.  A solve that goes on finds its matrix overwritten by the trial point's: it is assembled again (spec_void; as
  // rare as a second Krylov iteration on a column-structured problem, counted in spec_misses).
  struct LsAhead {
    bool armed = false, done = false;
    const double *F = nullptr, *X = nullptr;
    double r3[3] = {0, 0, 0};
    bool spec = false, spec_issued = false, spec_done = false, spec_fused = false;
    double *Xv = nullptr, *Wv = nullptr, *Gv = nullptr;
    double g2 = 0.0, w2 = 0.0;
  } ls_ahead;
  long long spec_misses = 0;
  long long ls_folds = 0;    // line searches whose F . (J Y) came out of the Krylov solve's own kernels (option ls_fold)
  int asm_slot = 0;          // complete-row march: where its two sums go (sc[S_SUM0 + asm_slot ..]) ...
  bool asm_defer = false;    // ... and whether the caller all-reduces and reads them itself
  bool ksp_one_shot = false;   // pending X + a Y of the line search
  int debug_fail_at_fetch = 0; // test hook: the n-th scalar read-back from now throws (multi-rank failure handling)
  long long chain_fallbacks = 0;   // chained ILU(0) applications redone with per-level launches after a dependency timeout
  DevBuf<int> d_col_bc;
  DevBuf<double> bc_side_f, bc_side_d, d_DqA, d_porvol, fwd_elim;
  const double *fwd_rhs = nullptr;   // the residual vector whose forward elimination fwd_elim holds (for the current factors)
  std::vector<int> h_cc_cell;
  double *h_sc = nullptr;  // pinned mirror of sc
  // Scalar read-backs without a copy in the queue: the kernel that finishes a reduction writes the scalar block into h_sc
  // itself (h_sc_dev: its device address) and then a sequence number into h_seq, which the host polls (fetch_scalars).
  double *h_sc_dev = nullptr;
  unsigned long long *h_seq = nullptr, *h_seq_dev = nullptr;
  unsigned long long pub_issued = 0;   // sequence number of the most recent publication in the queue
  bool pub_arm = false;                // the next finalize() is the last writer of sc before a read-back: it publishes
  bool pub_valid = false;              // ... and nothing has written sc since
  long long n_fetch = 0;               // scalar read-backs since the last mppgpu_reset_stats
  std::vector<DevBuf<double>> vv;
  DevBuf<double> gs_h;                 // Gram-Schmidt coefficients of a GMRES step with more than kMaxMultiDot basis vectors

  // ---------------- thermal KSP path ----------------
  bool thermal = false;
  double cnfac = 0.5;  // mpp_varcon.F90:28
  std::vector<double> th_dz, th_por, th_tkmg, th_tkdry, th_csol, th_liq, th_ice, th_snow, th_tuning;
  std::vector<int> th_itype, th_shallow, th_nsnow;
  bool th_cell_dirty = true;
  DevBuf<double> d_th_dz, d_th_por, d_th_tkmg, d_th_tkdry, d_th_csol, d_th_liq, d_th_ice, d_th_snow, d_th_tuning;
  DevBuf<signed char> d_th_itype, d_th_shallow, d_th_nsnow0;
  DevBuf<double> d_therm_cond, d_heat_cap, d_cdu, d_cdd, d_carea, soln_prev_clm;
  DevBuf<double> d_bc_du, d_bc_dd, d_bc_rawarea, d_bc_stalearea, d_bc_frac, d_bc_dhsdT;
  DevBuf<signed char> d_bc_act;
  std::vector<double> h_bc_frac, h_bc_dhsdT;
  std::vector<signed char> h_bc_act;
  // ---------------- coupled thermal-hydrology ----------------
  bool th = false;
  int ie_type = 1;                    // INT_ENERGY_ENTHALPY_CONSTANT (ThermEnthalpyAuxVarInit :97)
  std::vector<double> the_heat_cap, the_tcw, the_tcd, the_talpha, the_dsoil, the_permx, the_permy, the_permz;
  BigVec<int> hc_rowptr, hc_diag;   // the mesh (cell) pattern; h_rowptr / h_col / h_diag hold the 2N x 2N one
  int nnzc = 0;
  DevBuf<int> d_crowptr, d_cdiag;
  DevBuf<double> d_tcw, d_tcd, d_talpha, d_hcap, d_dsoil, d_Dq_e, d_bc_other, th_scratch;
  DevBuf<double> tha[13];
  DevBuf<double> thtri_L, thtri_inv, thtri_C;   // block-Thomas factors of the per-column blocks (4 planes each)
  std::vector<double> h_bc_other;
  // ---------------- ILU(0) ----------------
  DevBuf<int> d_ord, d_lrows, d_urows;
  std::vector<int> h_lptr, h_uptr;
  DevBuf<double> d_lu, d_idiag;
  bool ilu_ready = false;
  // level-ordered ELL copies of the factors for the triangular solves (kernels.hpp)
  int ell_KL = 0, ell_KU = 0;
  DevBuf<int> d_Lslot, d_Uslot, d_Lcol, d_Ucol, d_l2u, d_posU, d_fprog;
  DevBuf<long long> d_fptr;
  DevBuf<double> d_Lval, d_Uval, d_Uidiag, ilu_y, ilu_x;
  // chained triangular solves (ChainArgs): tiles of one level, the tiles each waits for, completion flags
  struct Chain {
    DevBuf<int> tile_lo, tile_hi, dep;
    DevBuf<unsigned> flags, ticket;
    int ntiles = 0, tiles_per_block = 1, nblocks = 0;
    unsigned launches = 0;
    bool ok = false;
  } chainL, chainU;
  static constexpr int kChainDeps = 8;
  unsigned chain_epoch = 0;
  // A fixed sequence of launches replayed from hipGraphs of at most opt.ilu_graph_nodes kernel nodes each (run_graph)
  struct GraphSeq {
    std::vector<hipGraph_t> graphs;
    std::vector<hipGraphExec_t> execs;
    void destroy() {
      for (auto e : execs) (void)hipGraphExecDestroy(e);
      for (auto g : graphs) (void)hipGraphDestroy(g);
      execs.clear(); graphs.clear();
    }
  };
  GraphSeq ilu_graph;
  GraphSeq *capturing = nullptr;   // the sequence being captured (graph_tick cuts it into chunks)
  int cap_nodes = 0;
  // box wavefront solves on a structured grid (BoxGrid, kernels.hpp)
  BoxGrid box{};
  bool box_ok = false, box_tried = false, ell_built = false;
  long long box_slots = 0;
  DevBuf<int> box_cellmap, box_cellpos, box_order;
  std::vector<int> box_lptr;                 // boxes of box level l: box_order[box_lptr[l] .. box_lptr[l+1])
  DevBuf<double> box_L[3], box_U[3], box_Di, box_v, box_scratch, box_T[3];
  GraphSeq box_graph, box_fgraph;
  // one launch per triangle (k_box_lower_queue / k_box_upper_queue): boxes by ticket in level order, completion flags
  DevBuf<int> box_order_rev, pen_order, pen_order_rev;   // pencils (J + NB2 K) by J + K, ascending / descending
  DevBuf<double> box_faces;                               // face buffers of the streamed solves (4 x box_stream_face_doubles), all bits set when idle
  unsigned pen_ticket_base[2] = {0, 0};
  bool box_stream_ready = false;
  int setup_verified = 0;   // option setup_verify: device-built set-up structures found equal to the host-built ones
  DevBuf<unsigned> box_flags, box_ticket;     // flags: [lower: nboxes][upper: nboxes]; ticket: one counter per triangle
  unsigned box_epoch = 0, box_ticket_base[2] = {0, 0};
  long long box_queue_fallbacks = 0;

  double dtime = 0.0, time = 0.0;
  long long cum_newton = 0, cum_linear = 0, cum_cuts = 0, cum_funcs = 0;
  int nfuncs = 0;
  bool ls_backtracked_last = false;
  bool ls_first_fused = false;    // option ls_trace: what the line search's first trial point evaluated
  int ls_trace_it = 0;
  // Will the line search of Newton iteration `ls_trace_it` reject the full step?  (Then its first trial point is evaluated without the
  // Jacobian.)  Yes if the previous iteration's did, or if the iteration of the same index did in the previous solve: on the
  // benchmark's steps the first k iterations backtrack and k moves slowly from step to step -- 33 instead of 50 wrong guesses in 274
  // line searches, profiles/r04/linesearch_trace_warmup5_steps20.txt.  Changes which passes assemble the Jacobian, never a result.
  std::vector<char> bt_hist_prev, bt_hist_cur;
  bool predict_backtrack() const {
    return ls_backtracked_last || (ls_trace_it < (int)bt_hist_prev.size() && bt_hist_prev[ls_trace_it]);
  }
  double ls_ynorm = -1.0;         // linesearch->ynorm as the previous application left it (cp reads it at entry)
  long long cum_ls_switches = 0;  // line searches swapped in by use_dynamic_linesearch
  std::vector<double> last_substeps;   // sub-steps the most recent StepDT accepted (dt, dt/2 ... after time-step cuts)
  bool pc_ready = false;
  std::vector<double> h_bc_value, h_ss_value;
  double *h_bc_stage = nullptr;          // pinned staging of boundary values on their way to the device (set_data)
  DevBuf<double> d_bc_stage;
  DevBuf<int> d_bc_pos;                  // bc_pos_of_g on the device
  hipEvent_t bc_stage_ev = nullptr;
  std::vector<char> bc_stage_pending;

  ~Solver() {
    ilu_graph.destroy(); box_graph.destroy(); box_fgraph.destroy();
    if (h_sc) (void)hipHostFree(h_sc);
    if (h_seq) (void)hipHostFree(h_seq);
    if (h_bc_stage) (void)hipHostFree(h_bc_stage);
    if (bc_stage_ev) (void)hipEventDestroy(bc_stage_ev);
    if (stream) (void)hipStreamDestroy(stream);
  }

  // =====================================================================================
  // staging setters
  // =====================================================================================
  void need_setup(bool want) const {
    if (want && !setup_done) throw Error("mppgpu_setup has not been called");
    if (!want && setup_done) throw Error("the mesh cannot change after mppgpu_setup");
  }

  void set_mesh(int ncells, int nlev_, const double *v, const int *act) {
    need_setup(false);
    if (ncells <= 0) throw Error("ncells must be positive");
    n = ncells; nlev = nlev_;
    vol.resize(n); active.resize(n);   // (first touched by the threads that fill them)
    std::atomic<bool> inact{false};
    host_parallel_for(n, [&](long long i) {
      vol[i] = v[i];
      active[i] = act ? (act[i] != 0) : 1;
      if (!active[i]) inact.store(true, std::memory_order_relaxed);
    });
    any_inactive = inact;
    nghost = 0;
  }

  void set_ghost_cells(int ng) {
    need_setup(false);
    if (ng < 0 || ng >= n) throw Error("mppgpu_set_ghost_cells: need 0 <= nghost < ncells");
    for (int i = 0; i < n; ++i) if (active[i] == 2) active[i] = 1;
    nghost = ng;
    for (int i = n - ng; i < n; ++i) active[i] = 2;
    any_inactive = false;
    for (int i = 0; i < n; ++i) any_inactive |= (active[i] != 1);
  }
  void set_halo(int nneigh, const int *peer, const int *send_ptr, const int *send_cells, const int *recv_count) {
    need_setup(false);
    if (nghost <= 0 && nneigh > 0) throw Error("mppgpu_set_halo: no ghost cells");
    halo.peer.assign(peer, peer + nneigh);
    halo.send_ptr.assign(send_ptr, send_ptr + nneigh + 1);
    halo.send_cells.resize(send_ptr[nneigh]);
    for (int k = 0; k < send_ptr[nneigh]; ++k) {
      if (send_cells[k] < 1 || send_cells[k] > n - nghost) throw Error("mppgpu_set_halo: only owned cells can be sent");
      halo.send_cells[k] = send_cells[k] - 1;
    }
    halo.recv_ptr.assign(nneigh + 1, 0);
    for (int k = 0; k < nneigh; ++k) halo.recv_ptr[k + 1] = halo.recv_ptr[k] + recv_count[k];
    if (halo.recv_ptr[nneigh] != nghost) throw Error("mppgpu_set_halo: the receive counts must add up to nghost");
    halo.set = true;
  }

  void set_connections(int nconn, const int *up, const int *dn, const double *du, const double *dd, const double *ar,
                       const double *uv) {
    need_setup(false);
    if (n == 0) throw Error("mppgpu_set_mesh must precede mppgpu_set_connections");
    PhaseTimer pt("set_connections");
    nc = nconn;
    cup_e.resize(nc); cdn_e.resize(nc);
    cdu.resize(nc); cdd.resize(nc); carea.resize(nc); cuv.resize(3 * (size_t)nc); cvert.resize(nc);
    host_parallel_for(nc, [&](long long c) {   // (the first exception any thread throws reaches the caller)
      if (up[c] < 1 || up[c] > n || dn[c] < 1 || dn[c] > n)
        throw Error("connection " + std::to_string(c + 1) + ": cell id out of range");
      if (up[c] == dn[c]) throw Error("connection " + std::to_string(c + 1) + ": id_up == id_dn");
      cup_e[c] = up[c] - 1; cdn_e[c] = dn[c] - 1;
      cdu[c] = du[c]; cdd[c] = dd[c]; carea[c] = ar[c];
      cuv[3 * (size_t)c] = uv[3 * (size_t)c]; cuv[3 * (size_t)c + 1] = uv[3 * (size_t)c + 1]; cuv[3 * (size_t)c + 2] = uv[3 * (size_t)c + 2];
      cvert[c] = is_vertical(&uv[3 * (size_t)c]);
    });
  }

  void set_lateral_connections(int nconn, const int *up, const int *dn, const double *du, const double *dd, const double *ar,
                               const double *uv) {
    if (n == 0) throw Error("mppgpu_set_mesh must precede mppgpu_set_lateral_connections");
    lat_up_e.resize(nconn); lat_dn_e.resize(nconn);
    for (int c = 0; c < nconn; ++c) {
      if (up[c] < 1 || up[c] > n || dn[c] < 1 || dn[c] > n)
        throw Error("lateral connection " + std::to_string(c + 1) + ": cell id out of range");
      lat_up_e[c] = up[c] - 1; lat_dn_e[c] = dn[c] - 1;
    }
    nlat = nconn;
    lat_du.assign(du, du + nconn); lat_dd.assign(dd, dd + nconn); lat_area.assign(ar, ar + nconn);
    lat_uv.assign(uv, uv + 3 * (size_t)nconn);
    soil_dirty = true;
  }

  // igoveqn: 1 = the (only, or mass) equation, 2 = the heat equation of the coupled thermal-hydrology system
  int add_condition(int kind, int cond_type, int nconn, const int *id_dn, const double *du, const double *dd,
                    const double *ar, const double *uv, int swap_order, int igoveqn = 1) {
    need_setup(false);
    if (n == 0) throw Error("mppgpu_set_mesh must precede mppgpu_add_condition");
    if (igoveqn != 1 && !(th && igoveqn == 2)) throw Error("no governing equation " + std::to_string(igoveqn));
    const bool heat = thermal || igoveqn == 2;
    std::vector<Condition> *list;
    if (kind == MPPGPU_COND_BC) {
      bool ok;
      if (thermal) ok = (cond_type == MPPGPU_COND_DIRICHLET || cond_type == MPPGPU_COND_HEAT_FLUX);
      else if (heat) ok = (cond_type == MPPGPU_COND_DIRICHLET);   // ThermEnthalpySoilUpdateAuxVarsBC aborts on others (GE :964-1017)
      else ok = (cond_type == MPPGPU_COND_DIRICHLET || cond_type == MPPGPU_COND_SEEPAGE_BC || cond_type == MPPGPU_COND_MASS_FLUX);
      if (!ok) throw Error("unsupported boundary cond_type " + std::to_string(cond_type));
      list = &bcs;
    } else if (kind == MPPGPU_COND_SS) {
      const bool downreg = (cond_type == MPPGPU_COND_DOWNREG_MASS_RATE_CAMPBELL || cond_type == MPPGPU_COND_DOWNREG_MASS_RATE_FETCH2);
      if (cond_type != (heat ? MPPGPU_COND_HEAT_RATE : MPPGPU_COND_MASS_RATE) && !(downreg && !heat && !th))
        throw Error("unsupported source-sink cond_type " + std::to_string(cond_type));
      list = &sss;
    } else throw Error("unknown condition kind " + std::to_string(kind));
    Condition c;
    c.kind = kind; c.cond_type = cond_type; c.swap_order = swap_order ? 1 : 0; c.n = nconn;
    c.eqn = igoveqn - 1;
    // boundary auxvar defaults: temperature 298.15 K (RichODEPressureAuxVarInit :91), pressure 0 (:90)
    c.other.assign(nconn, c.eqn == 0 ? 273.15 + 25.0 : 0.0);
    c.cell.resize(nconn);
    for (int i = 0; i < nconn; ++i) {
      if (id_dn[i] < 1 || id_dn[i] > n) throw Error("condition: cell id out of range");
      c.cell[i] = id_dn[i] - 1;
    }
    if (kind == MPPGPU_COND_BC) {
      c.dist_up.assign(du, du + nconn); c.dist_dn.assign(dd, dd + nconn); c.area.assign(ar, ar + nconn);
      c.uvec.assign(uv, uv + 3 * (size_t)nconn);
    }
    c.value.assign(nconn, 0.0);
    c.sink_p.assign(nconn, 0.0); c.sink_n.assign(nconn, 0.0);
    // the SoE numbers conditions equation by equation (mass first): keep the list in that order
    auto pos = list->end();
    if (th && c.eqn == 0) pos = std::find_if(list->begin(), list->end(), [](const Condition &q) { return q.eqn == 1; });
    const int id = (int)(pos - list->begin()) + 1;
    list->insert(pos, std::move(c));
    return id;
  }

  void set_perm(const double *px, const double *py, const double *pz) {
    if (!n) throw Error("mesh not set");
    permx.assign(px, px + n); permy.assign(py, py + n); permz.assign(pz, pz + n);
    soil_dirty = true;
  }
  void set_por(const double *p) {   // PorosityFunctionSetConstantModel
    if (!n) throw Error("mesh not set");
    por.assign(p, p + n);
    por_pref.clear(); por_slope.clear();
    soil_dirty = true;
  }
  void set_por_linear(const double *base, const double *pref, const double *slope) {   // PorosityFunctionSetLinearModel
    if (!n) throw Error("mesh not set");
    if (thermal) throw Error("the linear porosity model belongs to the VSFM and the coupled thermal-hydrology system");
    por.assign(base, base + n); por_pref.assign(pref, pref + n); por_slope.assign(slope, slope + n);
    soil_dirty = true;
  }
  void set_satfunc(const int *type, const double *a, const double *l, const double *sr) {
    if (!n) throw Error("mesh not set");
    bool mixed = false;
    for (int i = 0; i < n; ++i) {
      if (map_sat_func(type[i]) < 0) throw Error("unsupported saturation function " + std::to_string(type[i]) + " at cell " + std::to_string(i + 1));
      mixed |= map_sat_func(type[i]) != map_sat_func(type[0]);
      const double alpha_i = a[i], lam_i = l[i], sr_i = sr[i];
      bool bad = sr_i < 0.0 || sr_i > 0.5 || alpha_i <= 0.0 || alpha_i > 2.0 || lam_i <= 0.0;
      if (type[i] == MPPGPU_SAT_FUNC_VAN_GENUCHTEN) bad |= lam_i >= 1.0; else bad |= lam_i >= 2.0;
      if (bad) throw Error("bad saturation parameters at cell " + std::to_string(i + 1));  // SatFunc_Set_* argument checks
    }
    satf.assign(type, type + n); alpha.assign(a, a + n); lam.assign(l, l + n); sat_res.assign(sr, sr + n);
    mixed_satf = mixed;
    sat_func = map_sat_func(type[0]);   // the mesh's function when !mixed_satf
    all_vg_half = !mixed && (sat_func == kSatVG);
    for (int i = 0; i < n && all_vg_half; ++i) all_vg_half = (l[i] == 0.5);
    soil_dirty = true;
  }

  // =====================================================================================
  // layout
  // =====================================================================================
  // tiled level-major layout: columns in tiles of 64 (one wavefront); see the header of kernels.hip
  int cell_index(int lev, int col) const {
    const int t = col >> 6, tw = std::min(64, ncols - (t << 6));
    return t * 64 * nz + lev * tw + (col & 63);
  }

  // The set-up passes over the caller's cells and connections run on the host's cores (host_parallel_for): where the serial
  // code relied on the order of its loop, the parallel one fills with relaxed atomics in any order and then sorts each cell's short
  // list by connection number, so the results are the same for any number of threads.
  // a scratch array of the set-up passes whose pages are first touched -- and whose initial value is written -- by all host
  // threads (a std::vector of 10^8 ints is value-initialised by one: 0.15 s each, and these passes use a dozen)
  template <class T>
  struct Scratch {
    T *p = nullptr;
    size_t n = 0;
    Scratch(size_t count, T init) : p(static_cast<T *>(std::malloc(std::max<size_t>(count, 1) * sizeof(T)))), n(count) {
      if (!p) throw Error("out of host memory in set-up");
      T *q = p;
      host_parallel_for((long long)count, [q, init](long long i) { q[i] = init; });
    }
    explicit Scratch(size_t count) : p(static_cast<T *>(std::malloc(std::max<size_t>(count, 1) * sizeof(T)))), n(count) {
      if (!p) throw Error("out of host memory in set-up");
    }
    Scratch(const Scratch &) = delete;
    Scratch &operator=(const Scratch &) = delete;
    ~Scratch() { std::free(p); }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    T *begin() { return p; }
  };
  static int fetch_inc(int *p) { return __atomic_fetch_add(p, 1, __ATOMIC_RELAXED); }
  static void put(int *p, int v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }

  static bool is_vertical(const double *u) { return std::fabs(u[2]) > 0.5 * std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]); }
  void detect_columns() {
    // vertical chains of the internal connection set (any cell numbering)
    Scratch<int> nxt(n, -1), below(n, -1), deg_up(n, 0), deg_dn(n, 0);
    Scratch<char> is_vert(nc, 0);
    std::atomic<bool> ok{true};
    std::atomic<long long> nvert{0};
    host_parallel_ranges(nc, [&](long long c0, long long c1) {
      long long mine = 0;
      for (long long c = c0; c < c1; ++c) {
        if (is_vertical(&cuv[3 * (size_t)c])) {
          is_vert[c] = 1;
          ++mine;
          put(&nxt[cup_e[c]], cdn_e[c]); put(&below[cdn_e[c]], cup_e[c]);
          // a chain has at most ONE link above and one below every cell: counted, because two links between the SAME pair of cells
          // store the same values twice and cannot be told from one by comparing what was stored
          fetch_inc(&deg_up[cup_e[c]]); fetch_inc(&deg_dn[cdn_e[c]]);
        }
      }
      nvert += mine;
    });
    host_parallel_for(n, [&](long long i) { if (deg_up[i] > 1 || deg_dn[i] > 1) ok = false; });
    PhaseTimer pt2("  columns: chains and numbering");
    ext2int.resize(n); int2ext.resize(n);   // (every entry is written below)
    columns = false;
    if (nghost > 0) ok = false;   // ghost cells stay last: identity layout
    if (ok) {
      std::vector<int> starts;   // (ascending, gathered chunk by chunk)
      {
        const int nchunk = 4 * host_threads();
        std::vector<std::vector<int>> part((size_t)nchunk);
        host_parallel_tasks(nchunk, [&](int q) {
          for (long long i = (long long)n * q / nchunk, e = (long long)n * (q + 1) / nchunk; i < e; ++i)
            if (below[i] == -1) part[(size_t)q].push_back((int)i);
        });
        for (auto &v : part) starts.insert(starts.end(), v.begin(), v.end());
      }
      const int nchains = (int)starts.size();
      if (nchains > 0 && n % nchains == 0) {
        const int len = n / nchains;
        std::atomic<bool> equal{true};
        Scratch<int> lev(n, -1), colof(n, -1);
        host_parallel_for(nchains, [&](long long ch) {
          if (!equal.load(std::memory_order_relaxed)) return;
          int cur = starts[ch], k = 0;
          while (cur != -1 && k <= len) { lev[cur] = k; colof[cur] = (int)ch; cur = nxt[cur]; ++k; }
          if (k != len) equal = false;
        });
        if (equal) {
          ncols = nchains; nz = len;
          std::atomic<bool> all{true};
          host_parallel_for(n, [&](long long i) {
            if (lev[i] < 0) { all = false; return; }
            ext2int[i] = cell_index(lev[i], colof[i]);
          });
          columns = all;
        }
      }
    }
    if (!columns) {  // every cell its own block: identity layout
      ncols = n; nz = 1;
      host_parallel_for(n, [&](long long i) { ext2int[i] = (int)i; });
    }
    {
      std::atomic<bool> ident{true};
      host_parallel_for(n, [&](long long i) { int2ext[ext2int[i]] = (int)i; if (ext2int[i] != i) ident = false; });
      identity_perm = ident;
    }
    // internal connection numbering: chain links at the index of their lower cell, the rest appended
    cext2int.resize(nc);   // (every entry is written below)
    if (columns && nvert == nc) {   // nothing but chain links (the soil-column grids)
      host_parallel_for(nc, [&](long long c) { cext2int[c] = ext2int[cup_e[c]]; });
      nc_int = n;
      vertical_only = true;
    } else if (columns) {
      // lateral connections: the j-th one (in the caller's order) whose upwind cell is u sits at n + j*n + u, so that the rows
      // of a column tile (consecutive u) read consecutive geometry entries whichever end of the connection they are (holes
      // carry zeros).  j = the rank of the connection in its upwind cell's list: lists filled in any order, then sorted.
      Scratch<int> ptr((size_t)n + 1, 0);
      host_parallel_for(nc, [&](long long c) {
        if (is_vert[c]) cext2int[c] = ext2int[cup_e[c]];
        else fetch_inc(&ptr[(size_t)ext2int[cup_e[c]] + 1]);
      });
      const int jmax = host_prefix_sums(ptr.p, n);
      const int nlat = ptr[n];
      if (nlat > 0) {
        if ((long long)n * (jmax + 1) > 0x3fffffffLL) throw Error("too many lateral connections per cell for 32-bit connection ids");
        Scratch<int> fill((size_t)n), list((size_t)nlat);
        host_parallel_for(n, [&](long long i) { fill[i] = ptr[i]; });
        host_parallel_for(nc, [&](long long c) { if (!is_vert[c]) list[fetch_inc(&fill[ext2int[cup_e[c]]])] = (int)c; });
        host_parallel_for(n, [&](long long u) {
          std::sort(list.begin() + ptr[u], list.begin() + ptr[u + 1]);
          for (int e = ptr[u], j = 0; e < ptr[u + 1]; ++e, ++j) cext2int[list[e]] = n + j * n + (int)u;
        });
      }
      nc_int = n + jmax * n;
      vertical_only = (nlat == 0);
    } else {
      host_parallel_for(nc, [&](long long c) { cext2int[c] = (int)c; });
      nc_int = nc;
      vertical_only = false;
    }
  }

  int find_slot(int row, int col) const {
    const int *b = h_col.data() + h_rowptr[row], *e = h_col.data() + h_rowptr[row + 1];
    const int *p = std::lower_bound(b, e, col);
    return (p != e && *p == col) ? (int)(p - h_col.data()) : -1;
  }

  // soil columns with nothing but chain links: a cell's row is the cell below, the cell, the cell above -- written down directly
  // (no counting, no sorting; mppgpu_setup on the headline grid spent 0.7 s in the general route)
  void build_pattern_columns() {
    PhaseTimer pt("setup: CSR pattern (soil columns)");
    const long long per_tile = 64LL * nz;
    auto where = [&](long long i, int &k, int &tw) {
      const long long tile = i / per_tile;
      tw = (int)std::min<long long>(64, ncols - tile * 64);
      k = (int)((i - tile * per_tile) / tw);
    };
    h_rowptr.resize((size_t)n + 1);
    h_rowptr[0] = 0;
    host_parallel_for(n, [&](long long i) { int k, tw; where(i, k, tw); h_rowptr[i + 1] = 1 + (k > 0) + (k < nz - 1); });
    max_row_nnz = host_prefix_sums(h_rowptr.data(), n);
    nnz = h_rowptr[n];
    h_col.resize(nnz); h_diag.resize(n);
    host_parallel_for(n, [&](long long i) {
      int k, tw;
      where(i, k, tw);
      int q = h_rowptr[i];
      if (k > 0) h_col[q++] = (int)i - tw;
      h_diag[i] = q;
      h_col[q++] = (int)i;
      if (k < nz - 1) h_col[q++] = (int)i + tw;
    });
  }
  void build_pattern() {
    if (columns && vertical_only && opt.setup_device) {
      build_pattern_columns();
      if (opt.setup_verify) {
        const BigVec<int> rp = h_rowptr, cj = h_col, dg = h_diag;
        const int mx = max_row_nnz;
        build_pattern_general();
        if (rp != h_rowptr || cj != h_col || dg != h_diag || mx != max_row_nnz)
          throw Error("setup_verify: the column layout's CSR pattern differs from the one built from the connection list");
        ++setup_verified;
      }
      return;
    }
    build_pattern_general();
  }
  void build_pattern_general() {
    PhaseTimer pt("setup: CSR pattern");
    // CSR pattern = diagonal + both directions of every internal connection: exactly the entries the reference
    // creates with MatSetValuesLocal (GoveqnRichardsODEPressureType.F90:1689, 2051-2069), rows in internal order
    Scratch<int> cnt((size_t)n + 1, 1);   // (the diagonal)
    cnt[0] = 0;
    host_parallel_for(nc, [&](long long c) { fetch_inc(&cnt[(size_t)ext2int[cup_e[c]] + 1]); fetch_inc(&cnt[(size_t)ext2int[cdn_e[c]] + 1]); });
    host_prefix_sums(cnt.p, n);
    Scratch<int> col((size_t)cnt[n]), fill((size_t)n);
    host_parallel_for(n, [&](long long i) { fill[i] = cnt[i]; col[fill[i]++] = (int)i; });   // (each row's first entry: no other writer yet)
    host_parallel_for(nc, [&](long long c) {
      const int u = ext2int[cup_e[c]], d = ext2int[cdn_e[c]];
      col[fetch_inc(&fill[u])] = d; col[fetch_inc(&fill[d])] = u;
    });
    // sort every row, drop repeated columns (two connections between the same pair of cells), compact
    h_rowptr.resize((size_t)n + 1);   // ([i + 1] written row by row below)
    h_rowptr[0] = 0;
    host_parallel_for(n, [&](long long i) {
      int *b = col.p + cnt[i], *e = col.p + cnt[i + 1];
      std::sort(b, e);
      h_rowptr[i + 1] = (int)(std::unique(b, e) - b);
    });
    max_row_nnz = host_prefix_sums(h_rowptr.data(), n);
    nnz = h_rowptr[n];
    h_col.resize(nnz);
    host_parallel_for(n, [&](long long i) { std::copy(col.p + cnt[i], col.p + cnt[i] + (h_rowptr[i + 1] - h_rowptr[i]), h_col.begin() + h_rowptr[i]); });
    h_diag.resize(n);
    host_parallel_for(n, [&](long long i) { h_diag[i] = find_slot((int)i, (int)i); });
  }

  void setup() {
    PhaseTimer pt_all("setup (all of mppgpu_setup)");
    need_setup(false);
    if (n == 0) throw Error("mppgpu_set_mesh has not been called");
    HIP_CHECK(hipSetDevice(device));
    nrow = th ? 2 * n : n;
    nred = th ? nrow : nrow - nghost;
    if (nghost > 0 && !halo.set) throw Error("ghost cells need a halo plan (mppgpu_set_halo)");
    DevBuf<int> c2i, raw_up, raw_dn;   // cext2int and the caller's connection ends on the device while set-up needs them
    raw_up.upload(cup_e.data(), (size_t)nc, stream); raw_dn.upload(cdn_e.data(), (size_t)nc, stream);
    if (opt.setup_device) {
      // chains, the tiled level-major cell numbering and the connection numbering by kernels (setup_kernels.hip); the three maps
      // come back to the host, which keeps them for the caller-facing permutations and the conditions
      PhaseTimer pt("setup: column detection (device)");
      DevBuf<int> scratch[4], flags;
      DevBuf<signed char> vert;
      for (auto &b : scratch) b.alloc((size_t)n);
      flags.alloc(8);
      vert.upload(cvert.data(), (size_t)nc, stream);
      d_ext2int.alloc((size_t)n); d_int2ext.alloc((size_t)n); c2i.alloc((size_t)nc);
      ColumnLayout lay{};
      device_detect_columns(ColumnArgs{n, nc, raw_up.p, raw_dn.p, vert.p, scratch[0].p, scratch[1].p, scratch[2].p, scratch[3].p, flags.p},
                            nghost == 0, d_ext2int.p, d_int2ext.p, c2i.p, lay, stream);
      columns = lay.columns; vertical_only = lay.vertical_only; identity_perm = lay.identity_perm;
      ncols = lay.ncols; nz = lay.nz; nc_int = lay.nc_int;
      ext2int.resize(n); int2ext.resize(n); cext2int.resize(nc);
      host_parallel_for(n, [&](long long i) { ext2int[i] = 0; int2ext[i] = 0; });   // (pages touched by all threads before the copies land)
      host_parallel_for(nc, [&](long long c) { cext2int[c] = 0; });
      HIP_CHECK(hipMemcpyAsync(ext2int.data(), d_ext2int.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipMemcpyAsync(int2ext.data(), d_int2ext.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, stream));
      if (nc > 0) HIP_CHECK(hipMemcpyAsync(cext2int.data(), c2i.p, (size_t)nc * sizeof(int), hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      if (opt.setup_verify) {
        const BigVec<int> e2i = ext2int, i2e = int2ext, ce = cext2int;
        const ColumnLayout dev = lay;
        detect_columns();
        if (dev.columns != columns || dev.vertical_only != vertical_only || dev.identity_perm != identity_perm || dev.ncols != ncols ||
            dev.nz != nz || dev.nc_int != nc_int || e2i != ext2int || i2e != int2ext || ce != cext2int)
          throw Error("setup_verify: the device-built column layout differs from the host-built one");
        ++setup_verified;
      }
    } else {
      PhaseTimer pt("setup: column detection");
      detect_columns();
      d_ext2int.upload(ext2int, stream);
      d_int2ext.upload(int2ext, stream);
      c2i.upload(cext2int.data(), (size_t)nc, stream);
    }
    // per-row incident connection lists, in the caller's connection order (general kernels only)
    const bool want_incidence = !vertical_only || opt.assembly == 0 || thermal || th;
    const bool dev_lists = want_incidence && opt.setup_device;   // incidence lists and CSR pattern on the device (setup_kernels.hip)
    if ((long long)nc * 2 > 0x7fffffffLL && want_incidence) throw Error("too many connections for the 32-bit incidence keys");
    DevBuf<int> inc_roth, inc_flags;
    IncidenceArgs ia{n, nc, d_ext2int.p, raw_up.p, raw_dn.p, c2i.p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (dev_lists) {
      // count, scan, fill in any order, one pass per row (sort by the caller's (connection, side), other end); the CSR pattern is
      // the diagonal plus the distinct other ends of a row; only the pattern comes back to the host (find_slot and the index builders)
      PhaseTimer pt("setup: incidence lists + CSR pattern (device)");
      d_rc_ptr.alloc((size_t)n + 1); d_rc_ptr.zero(stream);
      inc_flags.alloc(3); inc_flags.zero(stream);
      ia.ptr = d_rc_ptr.p; ia.flags = inc_flags.p;
      const long long ninc = device_incidence_pointers(ia, stream);
      d_rc_conn.alloc((size_t)ninc); d_rc_slot.alloc((size_t)ninc); inc_roth.alloc((size_t)ninc);
      ia.rconn = d_rc_conn.p; ia.rslot = d_rc_slot.p; ia.roth = inc_roth.p;
      {
        DevBuf<int> cursor;
        cursor.alloc((size_t)n); cursor.zero(stream);
        device_incidence_fill(ia, cursor.p, stream);
        HIP_CHECK(hipStreamSynchronize(stream));   // (cursor goes out of scope)
      }
      d_rowptr.alloc((size_t)n + 1); d_rowptr.zero(stream);
      nnz = (int)device_csr_pointers(ia, d_rowptr.p, stream);
      d_col.alloc((size_t)nnz); d_diag.alloc((size_t)n);
      device_csr_columns(ia, d_rowptr.p, d_col.p, d_diag.p, inc_flags.p + 2, stream);
      ia.rowptr = d_rowptr.p; ia.col = d_col.p;
      h_rowptr.resize((size_t)n + 1); h_col.resize((size_t)nnz); h_diag.resize((size_t)n);
      host_parallel_for(n, [&](long long i) { h_rowptr[i] = 0; h_diag[i] = 0; });   // (pages touched by all threads before the copies land)
      host_parallel_for(nnz, [&](long long k) { h_col[k] = 0; });
      HIP_CHECK(hipMemcpyAsync(h_rowptr.data(), d_rowptr.p, ((size_t)n + 1) * sizeof(int), hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipMemcpyAsync(h_col.data(), d_col.p, (size_t)nnz * sizeof(int), hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipMemcpyAsync(h_diag.data(), d_diag.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipMemcpyAsync(&max_row_nnz, inc_flags.p + 2, sizeof(int), hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      if (opt.setup_verify) {   // the host-built pattern against it
        const BigVec<int> rp = h_rowptr, cj = h_col, dg = h_diag;
        const int mx = max_row_nnz;
        build_pattern();
        if (rp != h_rowptr || cj != h_col || dg != h_diag || mx != max_row_nnz)
          throw Error("setup_verify: the device-built CSR pattern differs from the host-built one");
        ++setup_verified;
      }
    } else {
      build_pattern();
    }
    auto ptu = std::make_unique<PhaseTimer>("setup: pattern + mesh uploads");
    if (!dev_lists) {
      d_rowptr.upload(h_rowptr, stream);
      d_col.upload(h_col, stream);
      d_diag.upload(h_diag, stream);
    }
    d_val.alloc(nnz); d_val.zero(stream);

    // mesh arrays in internal order
    BigVec<double> v(n);
    host_parallel_for(n, [&](long long i) { v[i] = vol[int2ext[i]]; });
    d_vol.upload(v, stream);
    if (any_inactive) {
      BigVec<signed char> a(n);
      host_parallel_for(n, [&](long long i) { a[i] = (signed char)active[int2ext[i]]; });
      d_active.upload(a.data(), a.size(), stream);
    }
    ptu.reset();
    // connection end cells (internal numbering both ways)
    {
      PhaseTimer pt("setup: connection end cells");
      // on the device: out[cext2int[c]] = ext2int[end cell of c] from the caller's lists as they are (holes of the lateral numbering: 0)
      const std::pair<const int *, DevBuf<int> *> ends[2] = {{raw_up.p, &d_cup}, {raw_dn.p, &d_cdn}};
      for (const auto &e : ends) {
        e.second->alloc((size_t)nc_int); e.second->zero(stream);
        launch_scatter_mapped(nc, c2i.p, d_ext2int.p, e.first, e.second->p, stream);
      }
      HIP_CHECK(hipStreamSynchronize(stream));
    }
    std::vector<int> v_ptr, v_conn, v_slot, v_enc, v_o, v_slt;   // option setup_verify: what the device route built
    int v_ellk = -1;
    if (dev_lists) {
      PhaseTimer pt("setup: incidence slots (device)");
      device_incidence_slots(ia, stream);
      int hf[2] = {0, 0};
      HIP_CHECK(hipMemcpyAsync(hf, inc_flags.p, sizeof hf, hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      aux_kr.alloc(n); aux_dkr.alloc(n); aux_den.alloc(n); aux_dden.alloc(n);
      ell_k = 0;
      if (!thermal && !th && hf[0] > 0 && hf[0] <= 8 && hf[1] == 0) {   // ELL copy for k_flux_rows_ell (see the host route below)
        const size_t cnt = (size_t)hf[0] * n;
        d_ell_enc.alloc(cnt); d_ell_o.alloc(cnt); d_ell_slot.alloc(cnt);
        device_fill_int(d_ell_enc.p, (long long)cnt, -1, stream); d_ell_o.zero(stream); device_fill_int(d_ell_slot.p, (long long)cnt, -1, stream);
        device_incidence_ell(ia, d_ell_enc.p, d_ell_o.p, d_ell_slot.p, stream);
        ell_k = hf[0];
      }
      HIP_CHECK(hipStreamSynchronize(stream));
      if (opt.setup_verify) {
        auto down = [&](std::vector<int> &h, const DevBuf<int> &d) {
          h.resize(d.n);
          if (d.n) HIP_CHECK(hipMemcpyAsync(h.data(), d.p, d.n * sizeof(int), hipMemcpyDeviceToHost, stream));
        };
        down(v_ptr, d_rc_ptr); down(v_conn, d_rc_conn); down(v_slot, d_rc_slot);
        if (ell_k) { down(v_enc, d_ell_enc); down(v_o, d_ell_o); down(v_slt, d_ell_slot); }
        HIP_CHECK(hipStreamSynchronize(stream));
        v_ellk = ell_k;
      }
    }
    if (want_incidence && (!opt.setup_device || opt.setup_verify)) {
      PhaseTimer pt("setup: incidence lists");
      Scratch<int> ptr((size_t)n + 1, 0);
      host_parallel_for(nc, [&](long long c) { fetch_inc(&ptr[(size_t)ext2int[cup_e[c]] + 1]); fetch_inc(&ptr[(size_t)ext2int[cdn_e[c]] + 1]); });
      const int kmax_rows = host_prefix_sums(ptr.p, n);
      Scratch<int> fill((size_t)n);
      host_parallel_for(n, [&](long long i) { fill[i] = ptr[i]; });
      BigVec<int> rconn(ptr[n]), rslot(ptr[n]), roth(ptr[n]);   // (every entry is written below)
      // a row's list keeps the caller's connection order: filled in any order as (connection << 1 | side), sorted per row
      if ((long long)nc * 2 > 0x7fffffffLL) throw Error("too many connections for the 32-bit incidence keys");
      host_parallel_for(nc, [&](long long c) {
        rconn[fetch_inc(&fill[ext2int[cup_e[c]]])] = (int)(c << 1);
        rconn[fetch_inc(&fill[ext2int[cdn_e[c]]])] = (int)(c << 1) | 1;
      });
      host_parallel_for(n, [&](long long i) {
        std::sort(rconn.begin() + ptr[i], rconn.begin() + ptr[i + 1]);
        for (int e = ptr[i]; e < ptr[i + 1]; ++e) {
          const int c = rconn[e] >> 1, side = rconn[e] & 1;
          roth[e] = side ? ext2int[cup_e[c]] : ext2int[cdn_e[c]];
          rconn[e] = (cext2int[c] << 1) | side;
          rslot[e] = find_slot((int)i, roth[e]);
        }
      });
      d_rc_ptr.upload(ptr.p, (size_t)n + 1, stream); d_rc_conn.upload(rconn, stream); d_rc_slot.upload(rslot, stream);
      aux_kr.alloc(n); aux_dkr.alloc(n); aux_den.alloc(n); aux_dden.alloc(n);
      // ELL copy for k_flux_rows_ell (rows of <= 8 incident connections, one connection per matrix entry)
      ell_k = 0;
      if (!thermal && !th) {
        const int kmax = kmax_rows;
        std::atomic<bool> one_to_one{true};
        host_parallel_for(n, [&](long long i) {
          for (int e = ptr[i]; e < ptr[i + 1]; ++e)
            for (int e2 = ptr[i]; e2 < e; ++e2)
              if (rslot[e2] == rslot[e]) one_to_one = false;
        });
        if (kmax > 0 && kmax <= 8 && one_to_one) {
          Scratch<int> enc((size_t)kmax * n, -1), oth((size_t)kmax * n, 0), slt((size_t)kmax * n, -1);
          host_parallel_for(n, [&](long long i) {
            for (int e = ptr[i], k = 0; e < ptr[i + 1]; ++e, ++k) {
              const size_t q = (size_t)k * n + i;
              enc[q] = rconn[e]; slt[q] = rslot[e]; oth[q] = roth[e];   // the other end of the incidence (internal numbering)
            }
          });
          d_ell_enc.upload(enc.p, enc.n, stream); d_ell_o.upload(oth.p, oth.n, stream); d_ell_slot.upload(slt.p, slt.n, stream);
          ell_k = kmax;
          if (v_ellk > 0 && (v_enc.size() != enc.n || !std::equal(v_enc.begin(), v_enc.end(), enc.p) || !std::equal(v_o.begin(), v_o.end(), oth.p) ||
                             !std::equal(v_slt.begin(), v_slt.end(), slt.p)))
            throw Error("setup_verify: the device-built ELL incidence arrays differ from the host-built ones");
        }
      }
      if (v_ellk >= 0) {   // option setup_verify: the device route ran first; its arrays against these
        if (v_ellk != ell_k) throw Error("setup_verify: ELL width " + std::to_string(v_ellk) + " (device) vs " + std::to_string(ell_k) + " (host)");
        if (v_ptr.size() != (size_t)n + 1 || !std::equal(v_ptr.begin(), v_ptr.end(), ptr.p))
          throw Error("setup_verify: the device-built incidence pointers differ from the host-built ones");
        if (v_conn.size() != rconn.size() || !std::equal(v_conn.begin(), v_conn.end(), rconn.begin()))
          throw Error("setup_verify: the device-built incidence lists differ from the host-built ones");
        if (!std::equal(v_slot.begin(), v_slot.end(), rslot.begin()))
          throw Error("setup_verify: the device-built incidence slots differ from the host-built ones");
        ++setup_verified;
      }
    }
    // tridiagonal slots of the per-column blocks
    {
      Scratch<int> lo((size_t)n, -1), up((size_t)n, -1);
      gc_n = 0;
      if (!columns && nghost > 0 && !th && !thermal) {
        // ghost cells keep the caller's order (identity layout), but the owned cells of a rank that was given whole soil columns
        // (mpp_amd.parallel.partition_mesh) still form vertical chains: found here from the vertical connections between owned cells
        const int nown = n - nghost;
        h_gc_nxt.assign(n, -1); h_gc_prv.assign(n, -1);
        bool ok = true;
        for (int c = 0; c < nc && ok; ++c) {
          if (!cvert[c]) continue;
          const int u = ext2int[cup_e[c]], d = ext2int[cdn_e[c]];
          if (u >= nown || d >= nown) continue;   // a link to a ghost cell: not part of this rank's blocks
          if (h_gc_nxt[u] != -1 || h_gc_prv[d] != -1) ok = false;   // two links on one side of a cell: not chains
          h_gc_nxt[u] = d; h_gc_prv[d] = u;
        }
        if (ok) {
          std::vector<int> start;
          for (int i = 0; i < n; ++i) if (h_gc_prv[i] == -1) start.push_back(i);   // (ghost cells and cells without a vertical neighbour: chains of one)
          for (int i = 0; i < n; ++i) {
            if (h_gc_prv[i] >= 0) lo[i] = find_slot(i, h_gc_prv[i]);
            if (h_gc_nxt[i] >= 0) up[i] = find_slot(i, h_gc_nxt[i]);
          }
          gc_n = (int)start.size();
          d_gc_start.upload(start, stream); d_gc_nxt.upload(h_gc_nxt, stream); d_gc_prv.upload(h_gc_prv, stream);
          HIP_CHECK(hipStreamSynchronize(stream));
        }
      }
      if (columns) {
        PhaseTimer pt("setup: column block slots");
        host_parallel_for(ncols, [&](long long col) {
          for (int k = 0; k < nz; ++k) {
            const int i = cell_index(k, (int)col);
            if (k > 0) lo[i] = find_slot(i, cell_index(k - 1, (int)col));
            if (k < nz - 1) up[i] = find_slot(i, cell_index(k + 1, (int)col));
          }
        });
      }
      d_tri_lo.upload(lo.p, lo.n, stream); d_tri_up.upload(up.p, up.n, stream);
      tri_l.alloc(n); tri_invb.alloc(n); tri_c.alloc(n); jac_invd.alloc(nrow);
    }
    // boundary conditions: entries grouped by cell, (condition, connection) order kept inside a cell
    auto ptc = std::make_unique<PhaseTimer>("setup: conditions, vectors, geometry");
    nb = 0;
    for (auto &c : bcs) { c.offset = nb; nb += c.n; }
    ns = 0;
    for (auto &c : sss) { c.offset = ns; ns += c.n; }
    h_bc_value.assign(nb, 0.0);
    h_ss_value.assign(ns, 0.0);
    bc_pos_of_g.assign(nb, -1);
    h_bc_cell_int.assign(nb, 0);
    {
      std::vector<std::pair<int, int>> key(nb);  // (internal cell, g)
      std::vector<int> gtype(nb);
      for (auto &c : bcs)
        for (int i = 0; i < c.n; ++i) {
          key[c.offset + i] = {ext2int[c.cell[i]], c.offset + i};
          gtype[c.offset + i] = c.cond_type | (c.swap_order << 16) | (c.eqn << 20);
        }
      std::sort(key.begin(), key.end());
      std::vector<int> cc_cell, cc_ptr, type(nb);
      for (int e = 0; e < nb; ++e) {
        if (e == 0 || key[e].first != key[e - 1].first) { cc_cell.push_back(key[e].first); cc_ptr.push_back(e); }
        bc_pos_of_g[key[e].second] = e;
        h_bc_cell_int[e] = key[e].first;
        type[e] = gtype[key[e].second];
      }
      cc_ptr.push_back(nb);
      ncond_cells = (int)cc_cell.size();
      h_cc_cell = cc_cell;
      d_cc_cell.upload(cc_cell, stream); d_cc_ptr.upload(cc_ptr, stream); d_bc_type.upload(type, stream);
      d_bc_value.alloc(nb); d_bc_value.zero(stream);
      d_bc_flux.alloc(nb); d_bc_flux.zero(stream);
      d_bc_Dq.alloc(nb); d_bc_dgrav.alloc(nb); d_bc_area.alloc(nb);
    }
    if (ns > 0) { d_src.alloc(nrow); d_src.zero(stream); has_src = true; }
    {  // down-regulated sinks grouped by cell, (condition, connection) order kept inside a cell
      std::vector<std::pair<int, int>> key;
      std::vector<int> gtype(ns, 0);
      for (auto &c : sss)
        if (c.cond_type == MPPGPU_COND_DOWNREG_MASS_RATE_CAMPBELL || c.cond_type == MPPGPU_COND_DOWNREG_MASS_RATE_FETCH2)
          for (int i = 0; i < c.n; ++i) { key.push_back({ext2int[c.cell[i]], c.offset + i}); gtype[c.offset + i] = c.cond_type; }
      std::sort(key.begin(), key.end());
      ndr = (int)key.size();
      dr_pos_of_g.assign(ns, -1);
      std::vector<int> cell, ptr, type(ndr);
      for (int e = 0; e < ndr; ++e) {
        if (e == 0 || key[e].first != key[e - 1].first) { cell.push_back(key[e].first); ptr.push_back(e); }
        dr_pos_of_g[key[e].second] = e;
        type[e] = gtype[key[e].second];
      }
      ptr.push_back(ndr);
      ndr_cells = (int)cell.size();
      h_dr_value.assign(ndr, 0.0); h_dr_pc.assign(ndr, 0.0); h_dr_n.assign(ndr, 0.0);
      if (ndr > 0) {
        d_dr_cell.upload(cell, stream); d_dr_ptr.upload(ptr, stream); d_dr_type.upload(type, stream);
        d_dr_flux.alloc(ndr);
        upload_sink_params();
      }
    }
    // vectors
    for (DevBuf<double> *b : {&soln, &soln_prev, &soln_prev_clm, &accum_prev, &F, &Y, &W, &G, &R, &RP, &V, &T, &S, &Pv, &Wk, &tmp_ext, &tmp_int}) {
      b->alloc(nrow); b->zero(stream);
    }
    partials.alloc((size_t)kMaxPartialBlocks * kMaxMultiDot);
    {  // one partial sum per block of 256 columns (column solves, complete-row march): two vectors + the staging tail
      const size_t nbc = ((size_t)ncols + 63) / 64;   // (the LDS column solve leaves one per 64-column tile)
      col_partials.alloc(4 * (nbc + nbc / 2048 + 1));   // (up to four sums per launch: k_tri_solve_lds FDOT)
      tri_scratch.alloc(64);
    }
    sc.alloc(S_COUNT); sc.zero(stream);
    d_powtab.upload(host_pow_tables(), kPowTableDoubles, stream);   // fpow's tables (fastpow.hpp)
    fin_stage.alloc(kFinalizeStage); fin_counter.alloc(1); fin_counter.zero(stream);
    HIP_CHECK(hipHostMalloc(&h_sc, S_COUNT * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    HIP_CHECK(hipHostMalloc(&h_seq, 64, hipHostMallocMapped | hipHostMallocCoherent));
    *h_seq = 0;
    HIP_CHECK(hipHostGetDevicePointer((void **)&h_sc_dev, h_sc, 0));
    HIP_CHECK(hipHostGetDevicePointer((void **)&h_seq_dev, h_seq, 0));
    d_Dq.alloc(nc_int); d_upw.alloc(nc_int); d_dgrav.alloc(nc_int); d_area.alloc(nc_int);
    if (thermal || th) {
      {   // connection geometry into the internal connection order, on the device: the caller's arrays are uploaded as they are and
          // scattered through cext2int there (one-to-one; the holes of the lateral numbering keep zeros)
        DevBuf<double> raw;
        raw.alloc((size_t)nc);
        const std::pair<const BigVec<double> *, DevBuf<double> *> geo[3] = {{&cdu, &d_cdu}, {&cdd, &d_cdd}, {&carea, &d_carea}};
        for (const auto &g : geo) {
          g.second->alloc((size_t)nc_int); g.second->zero(stream);
          if (nc > 0) HIP_CHECK(hipMemcpyAsync(raw.p, g.first->data(), (size_t)nc * sizeof(double), hipMemcpyHostToDevice, stream));
          launch_scatter(nc, c2i.p, raw.p, g.second->p, stream);
          HIP_CHECK(hipStreamSynchronize(stream));   // (raw is reused; the temporaries are freed at the end of the block)
        }
      }
      std::vector<double> bdu(nb, 0.0), bdd(nb, 0.0), bar(nb, 0.0);
      for (auto &cd : bcs)
        for (int i = 0; i < cd.n; ++i) {
          const int e = bc_pos_of_g[cd.offset + i];
          bdu[e] = cd.dist_up[i]; bdd[e] = cd.dist_dn[i]; bar[e] = cd.area[i];
        }
      d_bc_du.upload(bdu, stream); d_bc_dd.upload(bdd, stream); d_bc_rawarea.upload(bar, stream);
      if (thermal) {
        // option thermal_reference_defects: the value the reference routine's local `area` holds when it reaches a COND_DIRICHLET
        // connection (GoveqnThermalKSPTemperatureSoilType.F90:787, 906-911): the last internal connection between two active cells,
        // or the last COND_HEAT_FLUX connection of an active cell met before it; undefined in the reference without either (0 here)
        double stale = 0.0;   // (the LAST such internal connection: found from the end, not by walking all 4 x 10^8 of a 512^3 mesh)
        for (int c = nc - 1; c >= 0; --c) if (active[cup_e[c]] == 1 && active[cdn_e[c]] == 1) { stale = carea[c]; break; }
        std::vector<double> bst(nb, 0.0);
        for (auto &cd : bcs)
          for (int i = 0; i < cd.n; ++i) {
            if (active[cd.cell[i]] != 1) continue;
            if (cd.cond_type == MPPGPU_COND_HEAT_FLUX) stale = cd.area[i];
            bst[bc_pos_of_g[cd.offset + i]] = stale;
          }
        d_bc_stalearea.upload(bst, stream);
      }
    }
    if (th) th_setup();
    if (thermal) {
      // boundary auxvars start as ThermKSPTempBaseAuxVarInit leaves them (ThermalKSPTemperatureBaseAuxType.F90:56-60): inactive,
      // frac 0 -- the caller switches them on (mppgpu_set_data(AUXVAR_BC, VAR_ACTIVE / VAR_FRAC), as thermal_mms_problem.F90:664-668)
      h_bc_frac.assign(nb, 0.0); h_bc_dhsdT.assign(nb, 0.0); h_bc_act.assign(nb, 0);
      d_bc_frac.upload(h_bc_frac, stream); d_bc_dhsdT.upload(h_bc_dhsdT, stream);
      d_bc_act.upload(h_bc_act.data(), h_bc_act.size(), stream);
      d_therm_cond.alloc(n); d_heat_cap.alloc(n);
      if (ns > 0 && !has_src) { d_src.alloc(n); d_src.zero(stream); has_src = true; }
      // (liquid / ice / snow water, tuning factor 1, snow layers 0: host copies only once the caller sets them -- thermal_set_data)
      th_liq.clear(); th_ice.clear(); th_snow.clear(); th_tuning.clear(); th_nsnow.clear();
    }
    if (vertical_only && columns && !th && !thermal) {   // the column march computes the first slot of a row instead of loading it
      for (int col = 0; col < ncols; col += std::max(1, ncols / 257)) {
        const int tile = col >> 6, lane = col & 63, tw = std::min(64, ncols - (tile << 6));
        const long long st = (long long)tile * 64 * (3LL * nz - 2);
        for (int k = 0; k < nz; ++k) {
          const long long want = nz == 1 ? st + lane : k == 0 ? st + 2 * lane : (k == nz - 1 ? st + 2 * tw + 3LL * (nz - 2) * tw + 2 * lane : st + 2 * tw + 3LL * ((k - 1) * (long long)tw + lane));
          if (h_rowptr[cell_index(k, col)] != want) throw Error("internal: CSR pattern of the column layout is not the expected one");
        }
      }
    }
    ptc.reset();
    { PhaseTimer pt("setup: SpMV indices"); build_compact_index(); build_multibase_index(); }
    plan_complete_rows();
    lat_k = 0;
    if (columns && !vertical_only && !th && !thermal && nghost == 0 && ell_k > 4 && ell_k <= 8) {   // lateral pass of use_cols3d(): <= 4 laterals per row
      const size_t cnt = 4 * (size_t)n;
      d_lat_enc.alloc(cnt); d_lat_o.alloc(cnt); d_lat_slot.alloc(cnt);
      DevBuf<int> ovf;
      ovf.alloc(1); ovf.zero(stream);
      launch_ell_lateral(n, ell_k, d_ell_enc.p, d_ell_o.p, d_ell_slot.p, 4, d_lat_enc.p, d_lat_o.p, d_lat_slot.p, ovf.p, stream);
      int h = 0;
      HIP_CHECK(hipMemcpyAsync(&h, ovf.p, sizeof(int), hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      if (!h) lat_k = 4; else { d_lat_enc.release(); d_lat_o.release(); d_lat_slot.release(); }
    }
    if (nghost > 0) {
      halo.d_send_idx.upload(halo.send_cells, stream);
      halo.sendbuf.alloc(std::max<size_t>(halo.send_cells.size(), 1));
    }
    HIP_CHECK(hipStreamSynchronize(stream));
    setup_done = true;
    soil_dirty = true;
    check_ilu();
  }

  // The column march can finish every row itself (boundary fluxes included) when the cells that carry boundary
  // conditions are the end cells of their columns and nothing else touches F or the diagonal afterwards: then ||F||,
  // ||X|| and the block-Jacobi factors come out of the assembly pass (k_assemble_cols<.., FUSE>).
  void plan_complete_rows() {
    rows_plan = false;
    if (th || thermal || !vertical_only || !columns || nghost > 0 || ndr_cells > 0 || nz < 2) return;
    std::vector<int> cb(2 * (size_t)ncols, -1);
    for (int j = 0; j < ncond_cells; ++j) {
      const long long cell = h_cc_cell[j];
      const int tile = (int)(cell / (64LL * nz)), tw = std::min(64, ncols - tile * 64);
      const long long rem = cell - (long long)tile * 64 * nz;
      const int k = (int)(rem / tw), col = tile * 64 + (int)(rem % tw);
      if (k == 0) cb[col] = j;
      else if (k == nz - 1) cb[(size_t)ncols + col] = j;
      else return;
    }
    d_col_bc.upload(cb, stream);
    bc_side_f.alloc(std::max(ncond_cells, 1)); bc_side_d.alloc(std::max(ncond_cells, 1));
    rows_plan = true;
  }
  bool rows_complete() const { return rows_plan && opt.fused_rows && use_cols() && pick_nseg() == 1; }

  void upload_sink_params() {
    for (auto &c : sss)
      for (int i = 0; i < c.n; ++i) {
        const int e = dr_pos_of_g.empty() ? -1 : dr_pos_of_g[c.offset + i];
        if (e < 0) continue;
        h_dr_value[e] = c.value[i]; h_dr_pc[e] = c.sink_p[i]; h_dr_n[e] = c.sink_n[i];
      }
    if (ndr > 0) { d_dr_value.upload(h_dr_value, stream); d_dr_pc.upload(h_dr_pc, stream); d_dr_n.upload(h_dr_n, stream); }
  }
  // VSFMMPPSetSourceSinkAuxVarRealValue (MultiPhysicsProbVSFM.F90:1437-1520): values over the concatenated SS connections
  void set_ss_auxvar(int var_type, int cnt, const double *values) {
    if (var_type != MPPGPU_VAR_POT_MASS_SINK_PRESSURE && var_type != MPPGPU_VAR_POT_MASS_SINK_EXPONENT)
      throw Error("VSFMMPPSetSourceSinkAuxVarRealValue: Unknown var_type");
    int sum_conn = 0;
    for (auto &c : sss)
      for (int i = 0; i < c.n; ++i, ++sum_conn) {
        if (sum_conn >= cnt) throw Error("fewer values than source-sink connections");
        (var_type == MPPGPU_VAR_POT_MASS_SINK_PRESSURE ? c.sink_p : c.sink_n)[i] = values[sum_conn];
      }
    if (setup_done) { upload_sink_params(); HIP_CHECK(hipStreamSynchronize(stream)); }
  }
  SinkArgs sink_args(const double *P, double *Fout) const {
    return SinkArgs{P, any_inactive ? d_active.p : nullptr, ndr_cells, d_dr_cell.p, d_dr_ptr.p, d_dr_type.p, d_dr_value.p,
                    d_dr_pc.p, d_dr_n.p, d_diag.p, Fout, d_val.p, d_dr_flux.p};
  }

  // coupled thermal-hydrology: the 2N x 2N pattern (four blocks with the mesh pattern, rows [P(1..N); T(1..N)]),
  // the per-cell auxvar store and the boundary "other field" array
  void th_setup() {
    hc_rowptr = h_rowptr; hc_diag = h_diag; nnzc = nnz;
    const BigVec<int> ccol = h_col;
    d_crowptr.upload(hc_rowptr, stream); d_cdiag.upload(hc_diag, stream);
    h_rowptr.assign((size_t)nrow + 1, 0); h_col.assign(4 * (size_t)nnzc, 0); h_diag.assign(nrow, 0);
    int k = 0;
    for (int r = 0; r < nrow; ++r) {
      const int i = r < n ? r : r - n;
      h_rowptr[r] = k;
      for (int q = hc_rowptr[i]; q < hc_rowptr[i + 1]; ++q) { if (r < n && q == hc_diag[i]) h_diag[r] = k; h_col[k++] = ccol[q]; }
      for (int q = hc_rowptr[i]; q < hc_rowptr[i + 1]; ++q) { if (r >= n && q == hc_diag[i]) h_diag[r] = k; h_col[k++] = n + ccol[q]; }
    }
    h_rowptr[nrow] = k;
    nnz = k; max_row_nnz *= 2;
    d_rowptr.upload(h_rowptr, stream); d_col.upload(h_col, stream); d_diag.upload(h_diag, stream);
    d_val.alloc(nnz); d_val.zero(stream);
    for (auto &b : tha) { b.alloc(n); b.zero(stream); }
    th_scratch.alloc(nrow);
    d_Dq_e.alloc(nc_int);
    h_bc_other.assign(nb, 0.0);
    for (auto &cd : bcs)
      for (int i = 0; i < cd.n; ++i) h_bc_other[bc_pos_of_g[cd.offset + i]] = cd.other[i];
    d_bc_other.upload(h_bc_other, stream);
  }

  // soil parameters + connection geometry -> device (lazy: the reference sets soils after SetupProblem)
  void ensure_ready() {
    PhaseTimer pt("ensure_ready");
    need_setup(true);
    if (thermal) return;   // (the heat-diffusion operator has no Richards soils: its cell properties go up in thermal_upload_cells)
    if (!soil_dirty) return;
    if ((int)permx.size() != n || (int)por.size() != n || (int)alpha.size() != n)
      throw Error("soil properties are incomplete: set permeability, porosity and saturation function");
    // (the per-cell passes of this function run on the host's cores: they were 4.5 of the 9 s the 105 M-cell grid took to set up)
    BigVec<double> a(n), b(n), c(n), d(n);   // (first touched by the threads that fill them)
    host_parallel_for(n, [&](long long i) { const int e = int2ext[i]; a[i] = por[e]; b[i] = sat_res[e]; c[i] = alpha[e]; d[i] = lam[e]; });
    d_por.upload(a, stream); d_sat_res.upload(b, stream); d_alpha.upload(c, stream); d_m.upload(d, stream);
    if (use_cols3d() && opt.cols3d_rows && d_tri3.n < 3 * (size_t)n) d_tri3.alloc(3 * (size_t)n);
    if (col_operands()) {   // the column march streams porosity x volume as one operand
      HIP_CHECK(hipStreamSynchronize(stream));   // (a is about to be reused)
      host_parallel_for(n, [&](long long i) { a[i] = a[i] * vol[int2ext[i]]; });
      d_porvol.upload(a, stream);
      lev_pv = opt.level_geometry && por_slope.empty();   // ... or reads it per level when every column agrees
      std::vector<double> lp(nz, 0.0);
      if (lev_pv) {
        for (int k = 0; k < nz; ++k) lp[k] = a[cell_index(k, 0)];
        std::atomic<bool> same{true};
        host_parallel_for(ncols, [&](long long col) {
          if (!same.load(std::memory_order_relaxed)) return;
          for (int k = 0; k < nz; ++k)
            if (a[cell_index(k, (int)col)] != lp[k]) { same = false; return; }
        });
        lev_pv = same;
      }
      if (lev_pv) d_lev_porvol.upload(lp, stream);
    }
    build_soil_classes();
    if (mixed_satf) {
      BigVec<signed char> t(n);
      host_parallel_for(n, [&](long long i) { t[i] = (signed char)map_sat_func(satf[int2ext[i]]); });
      d_satf.upload(t.data(), t.size(), stream);
    } else d_satf.release();
    if (!por_slope.empty()) {
      HIP_CHECK(hipStreamSynchronize(stream));   // (a and b are reused)
      host_parallel_for(n, [&](long long i) { a[i] = por_pref[int2ext[i]]; b[i] = por_slope[int2ext[i]]; });
      d_por_pref.upload(a, stream); d_por_slope.upload(b, stream);
    } else { d_por_pref.release(); d_por_slope.release(); }
    bool any_sbc = false;
    sbc_kind = 0;
    {   // which smoothing kinds occur: 2 = bz2 only, 3 = bz3 only, 0 = none or both
      std::atomic<bool> f2{false}, f3{false};
      host_parallel_ranges(n, [&](long long i0, long long i1) {
        bool m2 = false, m3 = false;
        for (long long i = i0; i < i1; ++i) {
          m2 |= satf[i] == MPPGPU_SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ2;
          m3 |= satf[i] == MPPGPU_SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ3;
        }
        if (m2) f2 = true;
        if (m3) f3 = true;
      });
      const bool bz2 = f2, bz3 = f3;
      any_sbc = bz2 || bz3;
      if (bz2 != bz3) sbc_kind = bz2 ? 2 : 3;
    }
    if (any_sbc) {
      // SatFunc_Set_SBC_bz2 / _bz3 (SaturationFunction.F90:260-372), ps = -0.9/alpha (:1409,1412)
      BigVec<double> pu(n), ps(n), b2(n), b3(n);   // (per cell, two real powers each: on all host threads)
      host_parallel_for(n, [&](long long i) {
        const int e = int2ext[i];
        pu[i] = ps[i] = b2[i] = b3[i] = 0.0;
        if (map_sat_func(satf[e]) != kSatSBC) return;
        sbc_coefficients(satf[e], alpha[e], lam[e], pu[i], ps[i], b2[i], b3[i]);
      });
      d_pu.upload(pu, stream); d_ps.upload(ps, stream); d_b2.upload(b2, stream); d_b3.upload(b3, stream);
      HIP_CHECK(hipStreamSynchronize(stream));   // (the host arrays go out of scope)
    }
    // internal connection geometry (RichardsMod.F90:236-272)
    {
      Scratch<double> Dq(nc_int, 0.0), upw(nc_int, 0.0), dg(nc_int, 0.0), ar(nc_int, 0.0);   // (first touch by all host threads)
      host_parallel_for(nc, [&](long long cidx) {   // (cext2int is one-to-one: no two connections write the same entry)
        const int u = cup_e[cidx], dn = cdn_e[cidx];
        const double pu[3] = {permx[u], permy[u], permz[u]}, pd[3] = {permx[dn], permy[dn], permz[dn]};
        ConnGeom g = make_conn_geom(true, carea[cidx], cdu[cidx], cdd[cidx], &cuv[3 * (size_t)cidx], pu, pd);
        const int ic = cext2int[cidx];
        Dq[ic] = g.Dq; upw[ic] = g.upweight; dg[ic] = g.dist_gravity; ar[ic] = g.area;
      });
      d_Dq.upload(Dq.p, (size_t)nc_int, stream); d_upw.upload(upw.p, (size_t)nc_int, stream);
      d_dgrav.upload(dg.p, (size_t)nc_int, stream); d_area.upload(ar.p, (size_t)nc_int, stream);
      if (col_operands()) {   // ... and conductance x face area
        HIP_CHECK(hipStreamSynchronize(stream));   // (Dq is about to be reused)
        host_parallel_for(nc_int, [&](long long ic) { Dq[ic] = Dq[ic] * ar[ic]; });
        d_DqA.upload(Dq.p, (size_t)nc_int, stream);
        // extruded grids: the connection below level k has the same upwind weight and gravity distance in every column
        lev_geo = opt.level_geometry && nz >= 2;
        std::vector<double> lu(nz, 0.0), lg(nz, 0.0);
        if (lev_geo) {
          for (int k = 1; k < nz; ++k) { const int i0 = cell_index(k - 1, 0); lu[k] = upw[i0]; lg[k] = dg[i0]; }
          std::atomic<bool> same{true};
          host_parallel_for(ncols, [&](long long col) {
            if (!same.load(std::memory_order_relaxed)) return;
            for (int k = 1; k < nz; ++k) {
              const int ic = cell_index(k - 1, (int)col);
              if (upw[ic] != lu[k] || dg[ic] != lg[k]) { same = false; return; }
            }
          });
          lev_geo = same;
        }
        if (lev_geo) { d_lev_upw.upload(lu, stream); d_lev_dgrav.upload(lg, stream); }
      }
    }
    // boundary geometry: the BC auxvar carries the soil of its id_dn cell (MultiPhysicsProbVSFM.F90:422-446)
    if (nb > 0) {
      std::vector<double> Dq(nb), dg(nb), ar(nb);
      for (auto &cd : bcs)
        for (int i = 0; i < cd.n; ++i) {
          const int cell = cd.cell[i];
          const double pp[3] = {permx[cell], permy[cell], permz[cell]};
          ConnGeom g = make_conn_geom(false, cd.area[i], cd.dist_up[i], cd.dist_dn[i], &cd.uvec[3 * (size_t)i], pp, pp);
          const int e = bc_pos_of_g[cd.offset + i];
          Dq[e] = g.Dq; dg[e] = g.dist_gravity; ar[e] = g.area;
        }
      if (th) {  // heat-equation boundaries see the heat equation's own permeability
        for (auto &cd : bcs) {
          if (cd.eqn != 1) continue;
          for (int i = 0; i < cd.n; ++i) {
            const int cell = cd.cell[i];
            double pp[3];
            th_energy_perm(cell, pp);
            ConnGeom g = make_conn_geom(false, cd.area[i], cd.dist_up[i], cd.dist_dn[i], &cd.uvec[3 * (size_t)i], pp, pp);
            Dq[bc_pos_of_g[cd.offset + i]] = g.Dq;
          }
        }
      }
      d_bc_Dq.upload(Dq, stream); d_bc_dgrav.upload(dg, stream); d_bc_area.upload(ar, stream);
    }
    if (nlat > 0) {
      std::vector<double> Dq(nlat), upw(nlat), dg(nlat), ar(nlat);
      std::vector<int> up(nlat), dn(nlat);
      for (int c = 0; c < nlat; ++c) {
        const int u = lat_up_e[c], d = lat_dn_e[c];
        const double pu[3] = {permx[u], permy[u], permz[u]}, pd[3] = {permx[d], permy[d], permz[d]};
        ConnGeom g = make_conn_geom(true, lat_area[c], lat_du[c], lat_dd[c], &lat_uv[3 * (size_t)c], pu, pd);
        Dq[c] = g.Dq; upw[c] = g.upweight; dg[c] = g.dist_gravity; ar[c] = g.area;
        up[c] = ext2int[u]; dn[c] = ext2int[d];
      }
      d_lat_Dq.upload(Dq, stream); d_lat_upw.upload(upw, stream); d_lat_dgrav.upload(dg, stream); d_lat_area.upload(ar, stream);
      d_lat_up.upload(up, stream); d_lat_dn.upload(dn, stream);
      d_lat_flux.alloc(nlat);
    }
    if (th) {
      if ((int)the_heat_cap.size() != n) throw Error("heat-equation soil properties are not set (mppgpu_th_set_soils)");
      Scratch<double> Dq((size_t)nc_int, 0.0);
      host_parallel_for(nc, [&](long long cidx) {
        double pu[3], pd[3];
        th_energy_perm(cup_e[cidx], pu); th_energy_perm(cdn_e[cidx], pd);
        Dq[cext2int[cidx]] = make_conn_geom(true, carea[cidx], cdu[cidx], cdd[cidx], &cuv[3 * (size_t)cidx], pu, pd).Dq;
      });
      d_Dq_e.upload(Dq.p, (size_t)nc_int, stream);
      HIP_CHECK(hipStreamSynchronize(stream));
      auto perm_d = [&](const std::vector<double> &v, DevBuf<double> &d) {
        BigVec<double> t(n);
        host_parallel_for(n, [&](long long i) { t[i] = v[int2ext[i]]; });
        d.upload(t, stream);
        HIP_CHECK(hipStreamSynchronize(stream));   // (t goes out of scope)
      };
      perm_d(the_heat_cap, d_hcap); perm_d(the_tcw, d_tcw); perm_d(the_tcd, d_tcd); perm_d(the_talpha, d_talpha);
      perm_d(the_dsoil, d_dsoil);
    }
    HIP_CHECK(hipStreamSynchronize(stream));
    soil_dirty = false;
  }
  // soil classes of the column march: at most 256 distinct (saturation function, sat_res, alpha, m) tuples in the mesh ->
  // one byte per cell + a table (the smoothed Brooks-Corey coefficients are functions of the tuple), staged in LDS by every block.
  // (Wider ids with the table read through the caches were built and measured in round 3: a wave then gathers up to 64
  // different cache lines per parameter and level, 4.9 ms against the 2.7 ms of streaming the per-cell arrays at 4096 random
  // classes, and the second code path cost the one-byte kernels registers: profiles/r03_kbench_laws.json, not kept.)
  void build_soil_classes() {
    ncls = 0;
    d_soil_cls.release();
    if (!opt.soil_classes || !col_operands() || mixed_satf) return;
    struct Key { int f; double a, b, c; bool operator<(const Key &o) const { return std::tie(f, a, b, c) < std::tie(o.f, o.a, o.b, o.c); } };
    std::vector<Key> keys;             // sorted, <= 256 entries: binary search per cell, early exit on the 257th tuple
    std::vector<unsigned char> cls(n);
    int last = -1;
    for (int i = 0; i < n; ++i) {
      const int e = int2ext[i];
      const Key k{satf[e], sat_res[e], alpha[e], lam[e]};
      if (last >= 0 && !(k < keys[last]) && !(keys[last] < k)) continue;   // same tuple as the previous cell: the common case
      auto it = std::lower_bound(keys.begin(), keys.end(), k);
      if (it == keys.end() || k < *it) {
        if (keys.size() == 256) return;
        it = keys.insert(it, k);
      }
      last = (int)(it - keys.begin());
    }
    host_parallel_for(n, [&](long long i) {
      const int e = int2ext[i];
      const Key k{satf[e], sat_res[e], alpha[e], lam[e]};
      cls[i] = (unsigned char)(std::lower_bound(keys.begin(), keys.end(), k) - keys.begin());
    });
    ncls = (int)keys.size();
    std::vector<double> tab(7 * (size_t)ncls, 0.0);
    for (int q = 0; q < ncls; ++q) {
      tab[q] = keys[q].a; tab[ncls + q] = keys[q].b; tab[2 * ncls + q] = keys[q].c;
      if (map_sat_func(keys[q].f) == kSatSBC) {
        double pu, ps, b2, b3;
        sbc_coefficients(keys[q].f, keys[q].b, keys[q].c, pu, ps, b2, b3);
        tab[3 * ncls + q] = pu; tab[4 * ncls + q] = ps; tab[5 * ncls + q] = b2; tab[6 * ncls + q] = b3;
      }
    }
    d_soil_cls.upload(cls.data(), cls.size(), stream);
    d_cls_table.upload(tab, stream);
  }
  // SatFunc_Set_SBC_bz2 / _bz3 (SaturationFunction.F90:260-372), ps = -0.9/alpha (:1409,1412)
  static void sbc_coefficients(int satf_ext, double al, double la, double &p_u, double &p_s, double &b2, double &b3) {
    const bool bz2 = satf_ext == MPPGPU_SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ2;
    p_s = -0.9 / al;
    p_u = find_gu_sbc_zero_coeff(la, bz2 ? 3 : 2, -al * p_s) / (-al);
    const double bcAtPu = std::pow(-al * p_u, -la), ldp = la * (1.0 - p_s / p_u), inv = 1.0 / (p_u - p_s);
    if (bz2) { b2 = 0.0; b3 = (2.0 - bcAtPu * (2.0 + ldp)) * inv * inv * inv; }
    else { b2 = -(3.0 - bcAtPu * (3.0 + ldp)) * inv * inv; b3 = 0.0; }
  }
  // permeability of the heat equation's auxvars: 8.3913e-12 unless set (ThermEnthalpyAuxVarInit,
  // ThermalEnthalpySoilAuxType.F90:93; MPPTHSetSoils leaves it, MultiPhysicsProbTH.F90:293)
  void th_energy_perm(int cell_ext, double *pp) const {
    if ((int)the_permx.size() == n) { pp[0] = the_permx[cell_ext]; pp[1] = the_permy[cell_ext]; pp[2] = the_permz[cell_ext]; }
    else pp[0] = pp[1] = pp[2] = 8.3913e-12;
  }
  ThBcArgs th_args(const double *x, double *Fout) const {
    ThBcArgs b;
    ThArgs &a = b.t;
    a.n = n; a.x = x;
    a.por = d_por.p; a.sat_res = d_sat_res.p; a.alpha = d_alpha.p; a.m = d_m.p; a.vol = d_vol.p;
    a.sbc_pu = d_pu.p; a.sbc_ps = d_ps.p; a.sbc_b2 = d_b2.p; a.sbc_b3 = d_b3.p;
    a.active = any_inactive ? d_active.p : nullptr;
    a.satf = mixed_satf ? d_satf.p : nullptr;
    a.por_pref = d_por_pref.p; a.por_slope = d_por_slope.p;
    a.tc_wet = d_tcw.p; a.tc_dry = d_tcd.p; a.talpha = d_talpha.p; a.heat_cap = d_hcap.p; a.den_soil = d_dsoil.p;
    a.sat_func = sat_func; a.density_type = density_type; a.ie_type = ie_type;
    a.kr = tha[0].p; a.dkr = tha[1].p; a.den_m = tha[2].p; a.ddenP_m = tha[3].p; a.ddenT_m = tha[4].p;
    a.den_e = tha[5].p; a.ddenP_e = tha[6].p; a.ddenT_e = tha[7].p; a.hl = tha[8].p; a.dhl_dP = tha[9].p;
    a.dhl_dT = tha[10].p; a.tc = tha[11].p; a.dtc_dP = tha[12].p;
    a.cup = d_cup.p; a.cdn = d_cdn.p; a.rc_ptr = d_rc_ptr.p; a.rc_conn = d_rc_conn.p; a.rc_slot = d_rc_slot.p;
    a.crowptr = d_crowptr.p; a.cdiag = d_cdiag.p;
    a.Dq_m = d_Dq.p; a.Dq_e = d_Dq_e.p; a.upw = d_upw.p; a.dgrav = d_dgrav.p; a.area = d_area.p; a.cdu = d_cdu.p; a.cdd = d_cdd.p;
    a.accum_prev = accum_prev.p; a.src = has_src ? d_src.p : nullptr;
    a.F = Fout; a.val = d_val.p; a.nnzc = nnzc; a.dtInv = 1.0 / dtime;
    a.ncols = ncols; a.nz = nz;
    a.powtab = d_powtab.p;
    // the column march needs enough columns to fill the chip (one lane per column); otherwise the 3-pass kernels
    a.use_cols = (vertical_only && columns && opt.assembly != 0 && (opt.assembly == 1 || ncols >= 16384)) ? 1 : 0;
    b.ncond_cells = ncond_cells; b.cc_cell = d_cc_cell.p; b.cc_ptr = d_cc_ptr.p; b.type = d_bc_type.p;
    b.Dq = d_bc_Dq.p; b.dgrav = d_bc_dgrav.p; b.area = d_bc_area.p; b.du = d_bc_du.p; b.dd = d_bc_dd.p;
    b.value = d_bc_value.p; b.other = d_bc_other.p;
    return b;
  }

  // =====================================================================================
  // kernel argument packs
  // =====================================================================================
  CellArrays cells(const double *P) const {
    CellArrays ca;
    ca.P = P; ca.por = d_por.p; ca.sat_res = d_sat_res.p; ca.alpha = d_alpha.p; ca.m = d_m.p; ca.vol = d_vol.p;
    ca.sbc_pu = d_pu.p; ca.sbc_ps = d_ps.p; ca.sbc_b2 = d_b2.p; ca.sbc_b3 = d_b3.p;
    ca.active = any_inactive ? d_active.p : nullptr;
    ca.satf = mixed_satf ? d_satf.p : nullptr;
    ca.por_pref = d_por_pref.p; ca.por_slope = d_por_slope.p;
    ca.sat_func = sat_func; ca.all_vg_half = all_vg_half ? 1 : 0; ca.frac_liq = frac_liq;
    ca.frac_cell = d_frac.p;   // nullptr until VAR_FRAC_LIQ_SAT is set
    ca.eos = make_eos(density_type, temperature);
    ca.powtab = d_powtab.p;
    return ca;
  }
  ConnArrays conns() const { return ConnArrays{d_Dq.p, d_upw.p, d_dgrav.p, d_area.p}; }
  bool use_cols() const { return vertical_only && opt.assembly != 0 && !th; }
  // a 3-D mesh whose cells form complete soil columns (tiled level-major layout) with lateral connections on top: the column march
  // for the accumulation term and the chain links, then the lateral connections row by row from the ELL incidence lists (round 5;
  // launch_assemble_cols_lateral).  Per-row sum order: chain links first, then the lateral connections in the caller's order -- the
  // reference's order where the caller lists the vertical connections first (vsfm_vchannel_problem.F90 does), a permutation of it
  // otherwise (MeshCreate lists x, y, then z): residual rows then agree to rounding, not to the bit.
  bool use_cols3d() const {
    return opt.cols3d && columns && !vertical_only && !th && !thermal && opt.assembly != 0 && nghost == 0 && nz >= 2 && ell_k > 0 &&
           ell_k <= 8 && opt.ell_assembly;
  }
  bool col_operands() const { return columns && !th && !thermal && (vertical_only || use_cols3d()); }   // porosity x volume, Dq x area, per-level data
  int pick_nseg() const {
    if (opt.nseg > 0) return std::min(opt.nseg, nz);
    // The complete-row march (one thread per column: rows_complete) folds the column LU, the forward elimination, the trial
    // point and the norms into the assembly pass; from 32 768 columns that outweighs the occupancy of a split march
    // (65 536 x 50: 2283 against 1963 Newton it/s; 16 384 x 50: 2783 against 2855 -- profiles/r03_nseg_small_grids.json)
    if (rows_plan && opt.fused_rows && ncols >= (1 << 15)) return 1;
    // enough threads for >= 8 waves on each of the 256 CUs; one halo level per segment end is the price
    const long long want = 256LL * 8 * 64;
    int s = (int)std::min<long long>((want + ncols - 1) / std::max(ncols, 1), std::max(nz / 8, 1));
    return std::max(s, 1);
  }
  ColsArgs cols_args(const double *P, double *Fout) const {
    ColsArgs a;
    a.cell = cells(P); a.conn = conns();
    a.accum_prev = accum_prev.p; a.src = has_src ? d_src.p : nullptr;
    a.F = Fout; a.val = d_val.p; a.rowptr = d_rowptr.p;
    a.ncols = ncols; a.nz = nz; a.nseg = pick_nseg(); a.dtInv = 1.0 / dtime;
    a.DqA = d_DqA.p; a.porvol = d_porvol.p;
    if (ncls > 0) { a.soil_cls = d_soil_cls.p; a.cls_table = d_cls_table.p; a.ncls = ncls; }
    if (lev_geo) { a.lev_upw = d_lev_upw.p; a.lev_dgrav = d_lev_dgrav.p; }
    if (lev_pv) a.lev_porvol = d_lev_porvol.p;
    if (sbc_kind == 2) { a.sbc_b = d_b3.p; a.sbc_b_is_b3 = 1; } else if (sbc_kind == 3) a.sbc_b = d_b2.p;
    if (use_cols3d()) {   // 7-point rows: tridiagonal at its CSR slots, auxvars for the lateral pass, the source term left to that pass
      a.slot_lo = d_tri_lo.p; a.slot_di = d_diag.p; a.slot_up = d_tri_up.p;
      a.aux = AuxOut{aux_kr.p, aux_dkr.p, aux_den.p, aux_dden.p};
      a.src = nullptr;
      if (opt.cols3d_rows) a.tri3 = d_tri3.p;   // (allocated by ensure_ready)
    }
    return a;
  }
  GenArgs gen_args(const double *P, double *Fout) const {
    GenArgs a;
    a.cell = cells(P); a.conn = conns();
    a.cup = d_cup.p; a.cdn = d_cdn.p;
    a.rc_ptr = d_rc_ptr.p; a.rc_conn = d_rc_conn.p; a.rc_slot = d_rc_slot.p;
    if (ell_k > 0 && opt.ell_assembly) { a.ell_enc = d_ell_enc.p; a.ell_o = d_ell_o.p; a.ell_slot = d_ell_slot.p; a.ell_k = ell_k; }
    a.diag_slot = d_diag.p; a.rowptr = d_rowptr.p;
    a.accum_prev = accum_prev.p; a.src = has_src ? d_src.p : nullptr;
    a.aux = AuxOut{aux_kr.p, aux_dkr.p, aux_den.p, aux_dden.p};
    a.F = Fout; a.val = d_val.p; a.n = n; a.dtInv = 1.0 / dtime;
    return a;
  }
  GenArgs lat_args(const double *P, double *Fout) const {   // the lateral pass of use_cols3d(): the compacted lists where they exist
    GenArgs a = gen_args(P, Fout);
    if (lat_k > 0) { a.ell_enc = d_lat_enc.p; a.ell_o = d_lat_o.p; a.ell_slot = d_lat_slot.p; a.ell_k = lat_k; }
    if (d_DqA.p) a.conn = ConnArrays{d_DqA.p, d_upw.p, d_dgrav.p, nullptr};   // conductance x face area as one operand (col_operands)
    if (opt.cols3d_rows && d_tri3.p) { a.tri3 = d_tri3.p; a.slot_lo = d_tri_lo.p; a.slot_up = d_tri_up.p; }
    return a;
  }
  BcArgs bc_args(const double *P, double *Fout) const {
    BcArgs a;
    a.cell = cells(P);
    a.ncond_cells = ncond_cells; a.cc_cell = d_cc_cell.p; a.cc_ptr = d_cc_ptr.p;
    a.Dq = d_bc_Dq.p; a.dgrav = d_bc_dgrav.p; a.area = d_bc_area.p; a.value = d_bc_value.p; a.type = d_bc_type.p;
    a.diag_slot = d_diag.p; a.F = Fout; a.val = d_val.p; a.flux_out = d_bc_flux.p;
    return a;
  }

  // =====================================================================================
  // operators (device pointers, internal ordering)
  // =====================================================================================
  // refresh the ghost tail of `x` from the ranks that own those cells (no-op without ghost cells)
  void halo_exchange(const double *x_const) {
    if (nghost <= 0) return;
    if (!comm.active()) throw Error("ghost cells need a communicator (mppgpu_comm_init / mppgpu_comm_init_local)");
    ProfScope ps(prof, KC_VEC);
    double *x = const_cast<double *>(x_const);
    launch_gather((int)halo.send_cells.size(), halo.d_send_idx.p, x, halo.sendbuf.p, stream);
    comm.exchange(halo.sendbuf.p, halo.peer, halo.send_ptr, x + (n - nghost), halo.recv_ptr, stream);
    if (th) {   // ... and the temperatures
      launch_gather((int)halo.send_cells.size(), halo.d_send_idx.p, x + n, halo.sendbuf.p, stream);
      comm.exchange(halo.sendbuf.p, halo.peer, halo.send_ptr, x + n + (n - nghost), halo.recv_ptr, stream);
    }
  }
  // the ghost entries of a vector back to zero (coupled TH: see the note at nghost; elsewhere the tail is simply never read)
  void ghost_zero(const double *v_const) {
    if (nghost <= 0 || !th) return;
    double *v = const_cast<double *>(v_const);
    launch_set(nghost, 0.0, v + (n - nghost), stream);
    launch_set(nghost, 0.0, v + n + (n - nghost), stream);
  }

  void op_residual(const double *P, double *Fout) {
    if (dtime <= 0.0) throw Error("time step not set");
    halo_exchange(P);
    ProfScope ps(prof, KC_RESIDUAL);
    if (th) { launch_th_assemble(0, th_args(P, Fout), stream); ghost_zero(P); ++nfuncs; ++cum_funcs; return; }
    asm_sums = false;
    if (Fout == fwd_rhs) fwd_rhs = nullptr;
    if (rows_complete()) { assemble_complete_rows(0, P, Fout); ++nfuncs; ++cum_funcs; return; }
    if (use_cols()) launch_assemble_cols(0, cols_args(P, Fout), stream);
    else if (use_cols3d()) { launch_assemble_cols(0, cols_args(P, Fout), stream); launch_assemble_cols_lateral(0, lat_args(P, Fout), stream); }
    else {
      if (!d_rc_ptr.p) throw Error("option assembly=general must be set before mppgpu_setup");
      launch_assemble_general(0, gen_args(P, Fout), stream);
    }
    launch_bc(0, bc_args(P, Fout), stream);
    launch_sinks(0, sink_args(P, Fout), stream);
    ++nfuncs; ++cum_funcs;
  }
  // boundary fluxes first (into the side arrays), then the march that finishes every row; mode 0 / 2 leave ||F||^2 and
  // ||P||^2 in sc[S_SUM0], sc[S_SUM0+1] (asm_sums), mode 1 / 2 with the column preconditioner leave its factors (pc_ready)
  // have_F (mode 1): Fout holds the residual of this very state, so its forward elimination can ride along
  void assemble_complete_rows(int mode, const double *P, double *Fout, bool have_F = true) {
    BcArgs b = bc_args(P, Fout);
    ColsArgs a = cols_args(P, Fout);
    if (opt.bc_inline) {   // the march evaluates the boundary fluxes of its column's end cells itself (no k_bc launch)
      a.bc_ptr = b.cc_ptr; a.bc_type = b.type; a.bc_Dq = b.Dq; a.bc_dgrav = b.dgrav; a.bc_area = b.area; a.bc_value = b.value;
    } else {
      b.side_f = bc_side_f.p; b.side_d = bc_side_d.p;
      b.trial_x = ls_point.x; b.trial_y = ls_point.y; b.trial_a = ls_point.a;
      launch_bc(mode, b, stream);
    }
    a.trial_x = ls_point.x; a.trial_y = ls_point.y; a.trial_a = ls_point.a; a.trial_w = const_cast<double *>(P);
    ls_point = Trial{};
    a.col_bc = d_col_bc.p; a.side_f = bc_side_f.p; a.side_d = bc_side_d.p; a.partials = col_partials.p;
    const bool factor = mode != 0 && opt.pc_type == 2;
    if (factor) {
      a.tl = tri_l.p; a.tinvb = tri_invb.p; a.tc = tri_c.p;
      // the solves that form A x on the way redo the elimination factors from the rows they read: where those are the ones a solve
      // will use, neither l nor 1 / pivot is written (16 of the march's 107 bytes per cell: their back sweeps read g = up / pivot alone)
      // and both are rebuilt only if a plain solve asks (ensure_tri_l: the factorisation kernel, same arithmetic, same bits)
      // (BiCGStab only: every product of its iterations goes through op_BA_dots; GMRES uses the fused kernel for its first product alone
      // and would pay a factorisation pass per Newton iteration for the 16 bytes the march saves: ADVICE r04)
      a.store_l = !(opt.lazy_tri_l && opt.ksp_type == 2 && fused_spmv_pc_possible());
      tri_l_valid = a.store_l;
    }
    if (mode != 1 || Fout == fwd_rhs) fwd_rhs = nullptr;   // the residual in Fout or the factors are being replaced
    if (factor && have_F) {
      if (!fwd_elim.p) fwd_elim.alloc(nrow);
      a.fwd = fwd_elim.p;
    }
    launch_assemble_cols(mode, a, stream);
    if (a.fwd) fwd_rhs = Fout;
    if (mode != 1) {
      pub_arm = true;   // the caller reads the two sums next (residual_sums, or the Krylov solver for a speculative trial point)
      finalize((ncols + 255) / 256, 2, col_partials.p, asm_slot);
      if (!asm_defer) {
        if (comm.active()) comm.allreduce_sum(sc.p + S_SUM0, 2, stream);
        asm_sums = true;
      }
    }
    if (mode != 0) pc_ready = factor;
  }
  // ||F||^2 and ||P||^2 of the state just assembled: from the assembly pass when it produced them, else by reduction
  void residual_sums(const double *P, const double *Fv, double *out) {
    if (asm_sums) { fetch_scalars(); out[0] = h_sc[S_SUM0]; out[1] = h_sc[S_SUM0 + 1]; }
    else dots_host({{Fv, Fv}, {P, P}}, out);
  }
  // residual and Jacobian of the same state in one pass: the auxvars (the expensive part) are evaluated once
  void op_residual_jacobian(const double *P, double *Fout) {
    if (dtime <= 0.0) throw Error("time step not set");
    halo_exchange(P);
    ProfScope ps(prof, KC_JACOBIAN);
    if (th) { launch_th_assemble(2, th_args(P, Fout), stream); ghost_zero(P); ++nfuncs; ++cum_funcs; pc_ready = false; return; }
    asm_sums = false;
    fwd_rhs = nullptr;
    if (rows_complete()) { assemble_complete_rows(2, P, Fout); ++nfuncs; ++cum_funcs; return; }
    if (use_cols()) launch_assemble_cols(2, cols_args(P, Fout), stream);
    else if (use_cols3d()) { launch_assemble_cols(2, cols_args(P, Fout), stream); launch_assemble_cols_lateral(2, lat_args(P, Fout), stream); }
    else {
      if (!d_rc_ptr.p) throw Error("option assembly=general must be set before mppgpu_setup");
      launch_assemble_general(2, gen_args(P, Fout), stream);
    }
    launch_bc(2, bc_args(P, Fout), stream);
    launch_sinks(2, sink_args(P, Fout), stream);
    ++nfuncs; ++cum_funcs;
    pc_ready = false;
  }
  // Fres: the residual of the same state when the caller has it (lets the column march forward-eliminate it), else nullptr
  void op_jacobian(const double *P, double *Fres = nullptr) {
    if (dtime <= 0.0) throw Error("time step not set");
    halo_exchange(P);
    ProfScope ps(prof, KC_JACOBIAN);
    fwd_rhs = nullptr;
    if (th) { launch_th_assemble(2, th_args(P, th_scratch.p), stream); ghost_zero(P); pc_ready = false; return; }
    if (rows_complete()) { assemble_complete_rows(1, P, Fres ? Fres : F.p, Fres != nullptr); return; }
    if (use_cols()) launch_assemble_cols(1, cols_args(P, F.p), stream);
    else if (use_cols3d()) { launch_assemble_cols(1, cols_args(P, F.p), stream); launch_assemble_cols_lateral(1, lat_args(P, F.p), stream); }
    else launch_assemble_general(1, gen_args(P, F.p), stream);
    launch_bc(1, bc_args(P, F.p), stream);
    launch_sinks(1, sink_args(P, F.p), stream);
    pc_ready = false;
  }
  void op_pre_solve() {  // VSFMSOEPreSolve: accum_prev from soln_prev
    halo_exchange(soln_prev.p);
    ProfScope ps(prof, KC_RESIDUAL);
    if (th) { launch_th_assemble(3, th_args(soln_prev.p, accum_prev.p), stream); ghost_zero(soln_prev.p); return; }  // SOETHPreSolve :119-272
    launch_accum(cells(soln_prev.p), n, 1.0 / dtime, accum_prev.p, stream);
  }
  void op_spmv(const double *x, double *y) {
    halo_exchange(x);
    ProfScope ps(prof, KC_SPMV);
    const int v = spmv_variant();
    if (th_index_free()) {
      launch_th_spmv_cols(th_tri_args(), x, y, stream);   // coupled system on soil columns: no index at all (k_th_spmv_cols)
    } else if (v == 22 || v == 23) {
      if (!multibase_index) throw Error("spmv_variant 22/23 need the multi-base compact index (built when 20/21 are not available: rows of <= 8 entries, at most 8 windows of 8192 rows per tile)");
      launch_spmv_multibase(opt.spmv_tpw > 0 ? opt.spmv_tpw : (v == 22 ? 2 : 4), nred, max_row_nnz, d_tileptr.p, d_roff.p, d_mcol.p,
                            d_tbase.p, d_val.p, x, y, stream);
    } else if (v == 20 || v == 21) {
      if (!compact_index) throw Error("spmv_variant 20/21 need the compact index (rows of <= 8 entries whose columns lie within 2^15 of their 64-row tile)");
      launch_spmv_compact(opt.spmv_tpw > 0 ? opt.spmv_tpw : (v == 20 ? 2 : 4), nred, max_row_nnz, compact_split, d_tileptr.p, d_roff.p, d_dcol.p, d_val.p, x, y, stream);
    } else launch_spmv(v, nred, max_row_nnz, d_rowptr.p, d_col.p, d_val.p, x, y, stream);
    ghost_zero(x); ghost_zero(y);   // coupled TH with ghost cells: the operand's copies and the identity rows' outputs
  }
  // the line search's first reductions in one pass over the matrix: out = {y.y, f.(A y), x.x (x != nullptr)}
  // coupled system on soil columns: the index-free block-tridiagonal product (k_th_spmv_cols)
  bool th_index_free() const {
    return th && opt.th_spmv_cols && opt.spmv_variant < 0 && vertical_only && columns && nghost == 0 && nz >= 2;
  }
  // scalar system on soil columns: the line search's product from the column tridiagonals at their computed places (k_spmv_cols_dots)
  bool cols_index_free() const {
    return opt.spmv_cols_dots && !th && !thermal && opt.spmv_variant < 0 && vertical_only && columns && nghost == 0 && nz >= 2 &&
           max_row_nnz <= 3;
  }
  bool spmv_dots_fused() const {
    if (opt.fused_ls_dots && (th_index_free() || cols_index_free())) return true;
    const int v = spmv_variant();
    return opt.fused_ls_dots && (v == 20 || v == 21) && compact_index && nghost == 0;
  }
  // launches only: the sums go to sc[S_SUM0 + slot ..]; nreduce values from sc[S_SUM0] on are all-reduced (0: none)
  int op_spmv_dots_launch(const double *y, const double *f, const double *x, int slot, int nreduce) {
    // (with the reductions the kernel carries three more streams and one partial sum per block: 6 tiles per wave measured best,
    // 0.958 ms against 1.07 at 2 and 0.977 at 4: profiles/r02_kbench_spmv_dots_tpw.json)
    const int tpw = opt.spmv_tpw > 0 ? opt.spmv_tpw : 6, nv = x ? 3 : 2;
    if (cols_index_free()) {
      const size_t nb = (size_t)spmv_cols_dots_blocks(n), need = 3 * (nb + nb / 2048 + 1);
      if (spmv_partials.n < need) spmv_partials.alloc(need);
      int nbk;
      { ProfScope ps(prof, KC_SPMV); nbk = launch_spmv_cols_dots(n, ncols, nz, d_val.p, y, f, x, spmv_partials.p, stream); }
      ProfScope ps(prof, KC_REDUCE);
      finalize(nbk, nv, spmv_partials.p, slot);
      if (comm.active() && nreduce > 0) comm.allreduce_sum(sc.p + S_SUM0, nreduce, stream);
      return nv;
    }
    if (th_index_free()) {
      const size_t nb = (size_t)th_spmv_dots_blocks(n), need = 3 * (nb + nb / 2048 + 1);
      if (spmv_partials.n < need) spmv_partials.alloc(need);
      int nbk;
      { ProfScope ps(prof, KC_SPMV); nbk = launch_th_spmv_cols_dots(th_tri_args(), y, f, x, spmv_partials.p, stream); }
      ProfScope ps(prof, KC_REDUCE);
      finalize(nbk, nv, spmv_partials.p, slot);
      if (comm.active() && nreduce > 0) comm.allreduce_sum(sc.p + S_SUM0, nreduce, stream);
      return nv;
    }
    const size_t nbmax = (size_t)spmv_compact_dots_blocks(tpw, nred);
    const size_t need = 3 * (nbmax + nbmax / 2048 + 1);
    if (spmv_partials.n < need) spmv_partials.alloc(need);
    int nbk;
    {   // a CSR SpMV pass like any other (values, index, x in; f in instead of y out): timed in the SpMV class
      ProfScope ps(prof, KC_SPMV);
      nbk = launch_spmv_compact_dots(tpw, nred, max_row_nnz, compact_split, d_tileptr.p, d_roff.p, d_dcol.p, d_val.p, y, f, x,
                                     spmv_partials.p, stream);
    }
    ProfScope ps(prof, KC_REDUCE);
    finalize(nbk, nv, spmv_partials.p, slot);
    if (comm.active() && nreduce > 0) comm.allreduce_sum(sc.p + S_SUM0, nreduce, stream);
    return nv;
  }
  void op_spmv_dots(const double *y, const double *f, const double *x, double *out) {
    pub_arm = true;
    const int nv = op_spmv_dots_launch(y, f, x, 0, x ? 3 : 2);
    fetch_scalars();
    for (int j = 0; j < nv; ++j) out[j] = h_sc[S_SUM0 + j];
  }
  // 16-bit in-tile row offsets + 16-bit column offsets from the tile's first row (the pattern is fixed after setup)
  void build_compact_index() {
    compact_index = false;
    compact_split = 0;
    if (max_row_nnz > 8 || nghost > 0) return;   // (with ghost rows the last owned tile would need its own extent)
    const int split = th ? n : 0;                // field-blocked [P; T]: offsets per block, block bit in the entry
    const int ntiles = (nrow + 63) / 64;
    std::vector<int> tp((size_t)ntiles + 2, 0);
    BigVec<unsigned short> ro(nrow);   // (every entry is written below)
    BigVec<short> dc(nnz);
    std::atomic<bool> fits{true};
    host_parallel_for(ntiles, [&](long long t) {
      if (!fits) return;
      const int r0 = (int)t * 64, r1 = std::min(nrow, r0 + 64);
      const int base_lo = (split && r0 >= split) ? r0 - split : r0;
      if (split && r0 < split && r1 > split) { fits = false; return; }   // a tile across the block boundary (N not a multiple of 64)
      tp[t] = h_rowptr[r0];
      for (int r = r0; r < r1; ++r) {
        ro[r] = (unsigned short)(h_rowptr[r] - h_rowptr[r0]);
        for (int k = h_rowptr[r]; k < h_rowptr[r + 1]; ++k) {
          if (!split) {
            const int d = h_col[k] - r0;
            if (d < -32768 || d > 32767) { fits = false; return; }
            dc[k] = (short)d;
          } else {
            const int blk = h_col[k] >= split ? 1 : 0;
            const int d = h_col[k] - (base_lo + blk * split);
            if (d < -16384 || d > 16383) { fits = false; return; }
            dc[k] = (short)(d * 2 + blk);     // arithmetic shift in the kernel recovers d, bit 0 the block
          }
        }
      }
    });
    if (!fits) return;
    tp[ntiles] = nnz; tp[ntiles + 1] = nnz;
    d_tileptr.upload(tp, stream); d_roff.upload(ro.data(), ro.size(), stream); d_dcol.upload(dc.data(), dc.size(), stream);
    compact_index = true;
    compact_split = split;
  }
  // the windowed index for patterns the single-base one cannot hold (see k_spmv_pipe_mb): per tile the sorted columns are
  // covered greedily by windows of 8192 rows; more than 8 windows in any tile: no index
  void build_multibase_index() {
    multibase_index = false;
    const bool asked = opt.spmv_variant == 22 || opt.spmv_variant == 23;   // (also on request, next to the single-base index)
    if ((compact_index && !asked) || th || max_row_nnz > 8 || nghost > 0) return;
    const int ntiles = (nrow + 63) / 64;
    std::vector<int> tp((size_t)ntiles + 2, 0), tb((size_t)ntiles * 8, 0);
    BigVec<unsigned short> ro(nrow), mc(nnz);   // (every entry is written below)
    std::atomic<bool> fits{true};
    host_parallel_ranges(ntiles, [&](long long t0, long long t1) {
      std::vector<int> cols;
      for (long long t = t0; t < t1 && fits; ++t) {
        const int r0 = (int)t * 64, r1 = std::min(nrow, r0 + 64);
        tp[t] = h_rowptr[r0];
        cols.assign(h_col.begin() + h_rowptr[r0], h_col.begin() + h_rowptr[r1]);
        std::sort(cols.begin(), cols.end());
        int nb = 0;
        int *b = &tb[(size_t)t * 8];
        for (int c : cols)
          if (nb == 0 || c >= b[nb - 1] + 8192) {
            if (nb == 8) { fits = false; break; }
            b[nb++] = c;
          }
        if (!fits) break;
        for (int k = nb; k < 8; ++k) b[k] = nb ? b[nb - 1] : 0;
        for (int r = r0; r < r1; ++r) {
          ro[r] = (unsigned short)(h_rowptr[r] - h_rowptr[r0]);
          for (int k = h_rowptr[r]; k < h_rowptr[r + 1]; ++k) {
            int wdw = nb - 1;
            while (wdw > 0 && h_col[k] < b[wdw]) --wdw;
            mc[k] = (unsigned short)((wdw << 13) | (h_col[k] - b[wdw]));
          }
        }
      }
    });
    if (!fits) return;
    tp[ntiles] = nnz; tp[ntiles + 1] = nnz;
    d_tileptr.upload(tp, stream); d_roff.upload(ro.data(), ro.size(), stream); d_mcol.upload(mc.data(), mc.size(), stream);
    d_tbase.upload(tb, stream);
    multibase_index = true;
  }
  int spmv_variant() const {
    if (opt.spmv_variant >= 0) return opt.spmv_variant;
    if (compact_index && nrow >= (1 << 21)) return 20;   // measured: ahead of the stream kernel from ~2 M rows (3- and 7-point)
    if (multibase_index && nrow >= (1 << 21)) return 22;
    // measured on MI355X (profiles/): the software-pipelined wave kernel wins once there are enough row tiles to
    // fill the chip several times over; the block CSR-stream kernel wins on small, cache-resident matrices
    return (max_row_nnz <= 8 && nrow >= (1 << 23)) ? 17 : 1;
  }
  TriArgs tri_args() const {
    return TriArgs{ncols, nz, d_tri_lo.p, d_diag.p, d_tri_up.p, d_val.p, tri_l.p, tri_invb.p, tri_c.p};
  }
  ColChainArgs chain_args() const { return ColChainArgs{gc_n, d_gc_start.p, d_gc_nxt.p, d_gc_prv.p, tri_args()}; }
  ThTriArgs th_tri_args() const {
    return ThTriArgs{ncols, nz, n, nnzc, d_crowptr.p, d_tri_lo.p, d_cdiag.p, d_tri_up.p, d_val.p, thtri_L.p, thtri_inv.p, thtri_C.p, opt.th_stage ? 1 : 0,
                     opt.th_inv_fly ? 1 : 0};
  }
  void op_pc_setup() {
    if (pc_ready) return;
    if (opt.pc_type == 2 && th) {
      if (!vertical_only || !columns) throw Error("pc_type bjacobi of the coupled system needs a grid whose connections are all vertical (use ilu)");
      if (!thtri_L.p) { thtri_L.alloc(4 * (size_t)n); thtri_inv.alloc(4 * (size_t)n); thtri_C.alloc(4 * (size_t)n); }
      // BiCGStab's first act is z = M^{-1} b with its norm: the factorisation rides on that solve's sweep (k_th_tri_factor<true, .>,
      // op_pc_apply_dots); any other consumer of the factors runs the factorisation by itself first (ensure_th_factor)
      th_factor_pending = opt.th_lazy_factor && opt.ksp_type == 2 && fused_pc_dots() && th_index_free() && opt.th_spmv_tri;
      if (th_factor_pending) { pc_ready = true; return; }   // (no launch of the set-up class)
    }
    ProfScope ps(prof, KC_PC_SETUP);
    if (opt.pc_type == 2 && th) {
      launch_th_tri_factor(th_tri_args(), stream);
      th_l_valid = true;
    } else if (opt.pc_type == 2) {
      if (gc_n > 0) launch_chain_tri_factor(chain_args(), stream); else launch_tri_factor(tri_args(), stream);
      tri_l_valid = true;
    }
    else if (opt.pc_type == 3) launch_jacobi_setup(nrow, d_diag.p, d_val.p, jac_invd.p, stream);
    else if (opt.pc_type == 1) {
      ilu_box_analyse();
      if (opt.ilu_box == 2 && !box_ok) throw Error("ilu_box = 2: the matrix is not a structured grid in the caller's natural ordering");
      double *const L[3] = {box_L[0].p, box_L[1].p, box_L[2].p}, *const U[3] = {box_U[0].p, box_U[1].p, box_U[2].p};
      if (box_factored()) {   // structured grid: the factorisation itself runs box by box, no level analysis of the rows at all
        if (!box_T[0].p) for (auto &t : box_T) t.alloc((size_t)box_slots);
        double *const T[3] = {box_T[0].p, box_T[1].p, box_T[2].p};
        launch_box_load(box_slots, box, box_cellmap.p, d_int2ext.p, d_rowptr.p, d_col.p, nred, d_val.p, L, U, box_Di.p, T, stream);
        run_graph(box_fgraph, [&] {
          for (size_t l = 0; l + 1 < box_lptr.size(); ++l)
            graph_tick(), launch_box_factor(box, box_order.p + box_lptr[l], box_lptr[l + 1] - box_lptr[l], L, T, box_Di.p, box_scratch.p, stream);
        });
        pc_ready = true;
        return;
      }
      ilu_analyse();
      launch_copy(nnz, d_val.p, d_lu.p, stream);
      for (size_t l = 0; l + 1 < h_lptr.size(); ++l)
        launch_ilu_factor_program(d_fptr.p, d_fprog.p, d_lrows.p, d_diag.p, h_lptr[l], h_lptr[l + 1], d_lu.p, d_idiag.p, stream);
      if (box_active()) {   // into the box layout of the structured-grid solves
        launch_box_fill(box_slots, box, box_cellmap.p, d_int2ext.p, d_rowptr.p, d_col.p, nred, d_lu.p, d_idiag.p, L, U, box_Di.p, stream);
      } else {              // into the level-ordered ELL blocks the general solves stream through
        ilu_build_ell();
        launch_ilu_repack((long long)ell_KL * nrow, d_Lslot.p, d_lu.p, d_Lval.p, stream);
        launch_ilu_repack((long long)ell_KU * nrow, d_Uslot.p, d_lu.p, d_Uval.p, stream);
        launch_gather(nrow, d_urows.p, d_idiag.p, d_Uidiag.p, stream);
      }
    }
    pc_ready = true;
  }
  // every level of both triangular solves, on the fixed buffers ilu_y / ilu_x (captured once into a hipGraph)
  void ilu_solve_levels() {
    for (size_t l = 1; l + 1 < h_lptr.size(); ++l)   // level 0 of L has no dependencies: y = r already
      graph_tick(), launch_ell_lower_level(nrow, ell_KL, d_Lcol.p, d_Lval.p, h_lptr[l], h_lptr[l + 1], ilu_y.p, stream);
    graph_tick();
    launch_gather(nrow, d_l2u.p, ilu_y.p, ilu_x.p, stream);
    for (size_t l = 0; l + 1 < h_uptr.size(); ++l)
      graph_tick(), launch_ell_upper_level(nrow, ell_KU, d_Ucol.p, d_Uval.p, d_Uidiag.p, h_uptr[l], h_uptr[l + 1], ilu_x.p, stream);
  }
  // the box levels of both solves on the fixed buffer box_v
  void box_solve_levels() {
    const double *const L[3] = {box_L[0].p, box_L[1].p, box_L[2].p}, *const U[3] = {box_U[0].p, box_U[1].p, box_U[2].p};
    const int nl = (int)box_lptr.size() - 1;
    for (int l = 0; l < nl; ++l) graph_tick(), launch_box_lower(box, box_order.p + box_lptr[l], box_lptr[l + 1] - box_lptr[l], L, box_v.p, box_scratch.p, stream);
    for (int l = nl - 1; l >= 0; --l) graph_tick(), launch_box_upper(box, box_order.p + box_lptr[l], box_lptr[l + 1] - box_lptr[l], U, box_Di.p, box_v.p, box_scratch.p, stream);
  }
  // both box solves as streams (k_box_lower_stream): set-up on first use
  bool box_stream() const { return opt.ilu_box_stream && box_stream_applies(box); }
  void box_solve_stream(const double *r = nullptr, double *z = nullptr) {
    const int npen = box.NB2 * box.NB3;
    if (!box_stream_ready) {
      std::vector<int> ord(npen);
      std::iota(ord.begin(), ord.end(), 0);
      std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return a % box.NB2 + a / box.NB2 < b % box.NB2 + b / box.NB2; });
      pen_order.upload(ord, stream);
      std::vector<int> rev(ord.rbegin(), ord.rend());
      pen_order_rev.upload(rev, stream);
      box_faces.alloc(4 * box_stream_face_doubles(box));
      HIP_CHECK(hipMemsetAsync(box_faces.p, 0xFF, box_faces.n * sizeof(double), stream));   // "not yet": every value a NaN with all bits set
      box_stream_ready = true;
    }
    const double *const L[3] = {box_L[0].p, box_L[1].p, box_L[2].p}, *const U[3] = {box_U[0].p, box_U[1].p, box_U[2].p};
    const int nworkers = std::max(1, std::min(npen, opt.ilu_box_workers));
    BoxStreamArgs a{pen_order.p, pen_order_rev.p, npen, nworkers, box_ticket.p + 2, {pen_ticket_base[0], pen_ticket_base[1]}, opt.ilu_chain_spins, sc.p, box_faces.p};
    launch_box_solve_stream(box, a, L, U, box_Di.p, box_v.p, box_scratch.p, stream, r, z, box_cellmap.p);
    pen_ticket_base[0] += (unsigned)npen + (unsigned)nworkers;
    pen_ticket_base[1] += (unsigned)npen + (unsigned)nworkers;
  }
  // both box solves with one launch per triangle (tickets + flags)
  void box_solve_queue() {
    const double *const L[3] = {box_L[0].p, box_L[1].p, box_L[2].p}, *const U[3] = {box_U[0].p, box_U[1].p, box_U[2].p};
    const int nboxes = box.NB1 * box.NB2 * box.NB3;
    const int nworkers = std::max(1, std::min(nboxes, opt.ilu_box_workers));
    if (++box_epoch == 0) box_epoch = 1;   // (0 is what fresh flags hold)
    BoxQueueArgs lo{box_order.p, nboxes, nworkers, box_flags.p, box_epoch, box_ticket.p, box_ticket_base[0], opt.ilu_chain_spins, sc.p};
    BoxQueueArgs up{box_order_rev.p, nboxes, nworkers, box_flags.p + nboxes, box_epoch, box_ticket.p + 1, box_ticket_base[1], opt.ilu_chain_spins, sc.p};
    launch_box_solve_queue(box, lo, up, L, U, box_Di.p, box_v.p, box_scratch.p, stream);
    box_ticket_base[0] += (unsigned)nboxes + (unsigned)nworkers;   // every wave's last ticket finds the queue empty
    box_ticket_base[1] += (unsigned)nboxes + (unsigned)nworkers;
  }
  bool box_active() const { return box_ok && opt.ilu_box != 0; }
  bool box_factored() const { return box_active() && opt.ilu_box_factor && box_factor_applies(box); }
  // A fixed sequence of launches, captured the first time and replayed afterwards as hipGraphs of at most opt.ilu_graph_nodes kernel
  // nodes each (graph_tick, called in front of every launch of such a sequence, cuts the capture).
  // Profilers: rocprofv3 --kernel-trace (rocprofiler-sdk 1.1.0, ROCm 7.2) dies inside its queue-write interceptor when a graph replay
  // is submitted next to this library's HIP-event records.  The faulting instruction is the interceptor's read of a packet header
  // (`cmpb $0x2,(%rax)` at librocprofiler-sdk +0x1e72fb, in a loop that copies 64-byte AQL packets) at the first byte past the mapping
  // that holds the queue's ring: a multi-packet submission that wraps around the end of the ring is walked linearly
  // (profiles/r03_rocprofv3_graph_crash_attribution.md, disassembly added in round 4).  Only batched submissions can wrap, and the
  // event records next to the replays are what made it happen within a run: chunk sizes did not matter (16-node chunks died too), so
  // the chunking is not a remedy.  The remedy taken: while the library's own event profiling is on (option "profile"), the
  // sequences are issued launch by launch -- one packet per submission cannot wrap -- and replayed as graphs otherwise.
  // That keys on THIS library's profiling, not on the tool's presence (the library does not look at the environment): a run under
  // rocprofv3 with the library's events OFF still replays graphs and can still meet the interceptor's bug -- such runs pass
  // ilu_graph=0 (the scripts under tools/gpu/ do).  Round 5 re-ran the combination that crashed with the
  // remedy in place (events on, per-level box solves, rocprofv3 --kernel-trace --stats): exit code 0,
  // profiles/r05/rocprofv3_graph_events_on.log.
  void graph_chunk_begin() {
    HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    cap_nodes = 0;
  }
  void graph_chunk_end() {
    hipGraph_t g = nullptr;
    hipGraphExec_t e = nullptr;
    HIP_CHECK(hipStreamEndCapture(stream, &g));
    capturing->graphs.push_back(g);
    HIP_CHECK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
    capturing->execs.push_back(e);
  }
  void graph_tick() {
    if (!capturing) return;
    if (cap_nodes >= std::max(1, opt.ilu_graph_nodes)) { graph_chunk_end(); graph_chunk_begin(); }
    ++cap_nodes;
  }
  template <class F>
  void run_graph(GraphSeq &gs, F &&launches) {
    if (!opt.ilu_graph || prof.enabled) { launches(); return; }
    if (gs.execs.empty()) {
      capturing = &gs;
      try {
        graph_chunk_begin();
        launches();
        graph_chunk_end();
      } catch (...) {
        hipGraph_t g = nullptr;
        (void)hipStreamEndCapture(stream, &g);
        if (g) (void)hipGraphDestroy(g);
        capturing = nullptr;
        gs.destroy();
        throw;
      }
      capturing = nullptr;
    }
    for (auto e : gs.execs) HIP_CHECK(hipGraphLaunch(e, stream));
  }
  void op_pc_apply(const double *r, double *z) {
    ProfScope ps(prof, KC_PC_APPLY);
    if (nghost > 0 && !th) launch_set(nghost, 0.0, const_cast<double *>(r) + nred, stream);   // block-Jacobi across ranks (TH: zero already)
    if (opt.pc_type == 2 && th) { ensure_th_factor(true); launch_th_tri_solve(th_tri_args(), r, z, stream); }
    else if (opt.pc_type == 2 && gc_n > 0) launch_chain_tri_solve(chain_args(), r, z, stream);   // (a rank of a split mesh: columns as chains)
    else if (opt.pc_type == 2) { ensure_tri_l(); launch_tri_solve(tri_args(), r, z, stream); }
    else if (opt.pc_type == 3) launch_pointwise_mult(nred, jac_invd.p, r, z, stream);
    else if (opt.pc_type == 1 && box_active()) {
      auto gather_in = [&] {   // r into the box layout, padding zero
        if (box_gather_in_fits(box)) launch_box_gather_in(box, d_ext2int.p, r, box_v.p, stream);
        else launch_gather_or_zero(box_slots, box_cellmap.p, r, box_v.p, stream);
      };
      // the streamed solves read r and write z themselves, through the slot map (no pass over the vectors in front of / behind them)
      const bool fused_io = box_stream() && opt.ilu_box_fused_io;
      if (!fused_io) gather_in();
      bool solved = false;
      if (box_stream() || opt.ilu_box_queue) {
        if (box_stream()) box_solve_stream(fused_io ? r : nullptr, fused_io ? z : nullptr); else box_solve_queue();
        solved = !chain_timed_out();   // (a wait ran out of spins: the application is redone per level, the queue stays off)
        if (!solved) { ++box_queue_fallbacks; gather_in(); }
      }
      if (!solved) run_graph(box_graph, [&] { box_solve_levels(); });
      if (!(solved && fused_io)) {   // back to the internal row order
        if (opt.ilu_box_img_out && box_gather_in_fits(box)) launch_box_gather_out(box, d_ext2int.p, box_v.p, z, stream);
        else launch_gather(nred, box_cellpos.p, box_v.p, z, stream);
      }
      if (nghost > 0) launch_set(nghost, 0.0, z + nred, stream);             // (identity rows, zero right-hand side)
    } else if (opt.pc_type == 1) {
      launch_gather(nrow, d_lrows.p, r, ilu_y.p, stream);            // r into the L level order
      bool solved = false;
      if (ilu_chained()) {
        ilu_solve_chained();
        // A tile that gave up waiting for a dependency (spin budget) has computed from unready values: the whole
        // application is redone, on this rank alone, with one launch per level.  The answer is then the one every other
        // rank expects, so nothing has to be agreed across ranks; the chain stays off for this handle afterwards.
        solved = !chain_timed_out();
        if (!solved) launch_gather(nrow, d_lrows.p, r, ilu_y.p, stream);
      }
      if (!solved) run_graph(ilu_graph, [&] { ilu_solve_levels(); });
      launch_gather(nrow, d_posU.p, ilu_x.p, z, stream);             // back to the internal row order
    } else launch_copy(nred, r, z, stream);
  }
  // t = M^{-1} A v with the Krylov dot products of t, in ONE kernel: vertical-only grids, column preconditioner, LDS column solve
  // everything fused_spmv_pc() asks for except the state of the factors (decided while they are being made)
  bool fused_spmv_pc_possible() const {
    return opt.fused_spmv_pc && opt.pc_type == 2 && !th && !thermal && vertical_only && columns && nghost == 0 && nz >= 2 && tri_lds() &&
           fused_pc_dots();
  }
  bool tri_l_valid = true;
  // coupled system, block Thomas: the factors not computed yet (the first solve will, on its way) / planes L and inv (what the
  // stand-alone solves read besides G: the combined kernel writes no L, and no inv either under th_inv_fly) hold the present factors
  bool th_factor_pending = false, th_l_valid = false;
  void ensure_th_factor(bool need_l) {
    if (!th || opt.pc_type != 2 || !pc_ready) return;
    if (!th_factor_pending && (th_l_valid || !need_l)) return;
    launch_th_tri_factor(th_tri_args(), stream);
    th_factor_pending = false;
    th_l_valid = true;
  }
  void ensure_tri_l() {
    if (tri_l_valid || th || thermal) return;
    ProfScope ps(prof, KC_PC_SETUP);
    launch_tri_factor(tri_args(), stream);
    tri_l_valid = true;
  }
  bool fused_spmv_pc() const {
    return opt.fused_spmv_pc && opt.pc_type == 2 && !th && !thermal && vertical_only && columns && nghost == 0 && nz >= 2 && tri_lds() &&
           pc_ready && fused_pc_dots();
  }
  // coupled system on soil columns: t = M^{-1} A v with its dot products as one kernel (k_th_spmv_tri), where the index-free SpMV and
  // the block-Thomas solve with fused dots would run one after the other
  bool th_fused_spmv_tri() const {
    return opt.th_spmv_tri && th_index_free() && pc_ready && fused_pc_dots();
  }
  // fdot (coupled system's fused kernel only, with w): a third sum, fdot . (A v), lands in sc[S_SUM0 + slot + 2] (ls_fold)
  void op_BA_dots(const double *v, double *t, const double *w, int slot = 0, int nreduce = -1, int step = -1, const double *fdot = nullptr) {
    if (th_fused_spmv_tri()) {
      const int nv = (w ? 2 : 1) + ((fdot && w) ? 1 : 0);
      {
        ProfScope ps(prof, KC_SPMV_PC);
        ensure_th_factor(false);
        const int nbk = launch_th_spmv_tri(th_tri_args(), v, t, w, col_partials.p, stream, w ? fdot : nullptr);
        finalize(nbk, nv, col_partials.p, slot, step);
      }
      if (nreduce < 0) nreduce = nv;
      if (comm.active() && nreduce > 0) comm.allreduce_sum(sc.p + S_SUM0, nreduce, stream);
      bcgs_step(step);
      return;
    }
    if (!fused_spmv_pc()) {
      op_spmv(v, Wk.p);
      op_pc_apply_dots(Wk.p, t, w, slot, nreduce, step);
      return;
    }
    {
      ProfScope ps(prof, KC_SPMV_PC);
      const Fold fd = make_fold(tri_solve_lds_blocks(ncols), w ? 2 : 1, slot, step);
      const int nbk = launch_spmv_tri_lds(tri_args(), v, t, w, col_partials.p, tri_scratch.p, stream, fd);
      if (!fd.counter) finalize(nbk, w ? 2 : 1, col_partials.p, slot, step);
    }
    if (nreduce < 0) nreduce = w ? 2 : 1;
    if (comm.active() && nreduce > 0) comm.allreduce_sum(sc.p + S_SUM0, nreduce, stream);
    bcgs_step(step);
  }
  void op_BA(const double *v, double *t) {  // t = M^{-1} A v  (left preconditioning)
    op_spmv(v, Wk.p);
    op_pc_apply(Wk.p, t);
  }
  // z = M^{-1} r together with the dot products (z.z) or (z.w, z.z) of the result, left in sc[S_SUM0..]: one pass less
  // over z (and w) when the preconditioner is the column solver; otherwise the apply followed by the reduction kernels
  bool fused_pc_dots() const {
    return opt.pc_type == 2 && nghost == 0 && (!th || opt.th_fused_dots);
  }
  // the column solve that parks L^{-1} r in LDS between its sweeps (k_tri_solve_lds) instead of going through z
  bool tri_lds() const {
    if (nz > kTriLdsMaxLevels || nz < 2) return false;
    if (opt.tri_lds >= 0) return opt.tri_lds != 0;
    // auto: when a column tile's slab leaves three waves per CU (<= 53 KB) and the grid fills the chip with them several
    // times over; measured on MI355X (profiles/r02_kbench_tri_lds.json)
    return ncols >= (1 << 16) && nz <= 104;
  }
  // slot: the sums land in sc[S_SUM0 + slot ..]; nreduce: how many values from sc[S_SUM0] on are all-reduced afterwards
  // (-1: just this call's; 0: none -- the caller folds them into a later all-reduce).  slot / nreduce other than the
  // defaults are for the fused path only.
  // step >= 0: BiCGStab's scalar recurrence that consumes the sums follows (bcgs_step)
  void op_pc_apply_dots(const double *r, double *z, const double *w, int slot = 0, int nreduce = -1, int step = -1) {
    if (!fused_pc_dots()) {
      op_pc_apply(r, z);
      if (w) dots_dev({{w, z}, {z, z}}, step); else dots_dev({{z, z}}, step);
      return;
    }
    {
      ProfScope ps(prof, KC_PC_APPLY);
      int nbk;
      if (th && th_factor_pending) {
        nbk = launch_th_tri_factor_solve_dots(th_tri_args(), r, z, w, col_partials.p, stream);
        th_factor_pending = false;
        th_l_valid = false;
      } else if (th) { ensure_th_factor(true); nbk = launch_th_tri_solve_dots(th_tri_args(), r, z, w, col_partials.p, stream); }
      else if (!w && r == fwd_rhs && pc_ready)
        nbk = launch_tri_back_dots(tri_args(), fwd_elim.p, z, col_partials.p, stream);
      else if (tri_lds()) { ensure_tri_l(); nbk = launch_tri_solve_lds(tri_args(), r, z, w, col_partials.p, tri_scratch.p, opt.tri_group, stream); }
      else { ensure_tri_l(); nbk = launch_tri_solve_dots(tri_args(), r, z, w, col_partials.p, stream); }
      finalize(nbk, w ? 2 : 1, col_partials.p, slot, step);
    }
    if (nreduce < 0) nreduce = w ? 2 : 1;
    if (comm.active() && nreduce > 0) comm.allreduce_sum(sc.p + S_SUM0, nreduce, stream);
    bcgs_step(step);
  }

  // ---- reductions: deterministic two-stage sums, RCCL all-reduce across ranks, results on device and host ----
  // Second stage of a reduction: sc[S_SUM0 + slot + j] = the sum of partials[j*nbk ..], one launch.  Without a communicator
  // the block that finishes the sums also runs the BiCGStab scalar step that consumes them; across ranks that step waits for
  // the all-reduce (bcgs_step, after it).
  void finalize(int nbk, int nvec, const double *partials_, int slot = 0, int step = -1) {
    if ((size_t)nvec * finalize_chunks(nbk) > (size_t)kFinalizeStage)
      throw Error(std::to_string(nvec) + " sums of " + std::to_string(nbk) + " partial sums exceed the reductions' staging buffer");
    Finalize f{fin_stage.p, fin_counter.p, sc.p, comm.active() ? -1 : step};
    pub_valid = false;
    if (pub_arm && !comm.active() && opt.host_poll) {   // (across ranks the all-reduce follows: see bcgs_step / fetch_scalars)
      f.pub_host = h_sc_dev; f.pub_seq = h_seq_dev; f.pub_value = ++pub_issued;
      pub_valid = true;
      pub_arm = false;
    }
    launch_finalize(nbk, nvec, partials_, sc.p + slot, f, stream);
  }
  // the same second stage finished by the kernel that produces the partial sums (kernels.hpp: Fold) -- option fold_finalize, for
  // reductions of at most kFoldMaxPartials partial sums; what finalize() decides about the scalar step and the publication is
  // decided here, before the producer is launched.  A Fold without a counter: the caller follows up with finalize().
  Fold make_fold(int nbk, int nvec, int slot, int step) {
    Fold f;
    if (!opt.fold_finalize || nbk > kFoldMaxPartials || nbk < 1) return f;
    f.counter = fin_counter.p; f.sums = sc.p + slot; f.sc = sc.p; f.step = comm.active() ? -1 : step; f.nvec = nvec;
    pub_valid = false;
    if (pub_arm && !comm.active() && opt.host_poll) {
      f.pub_host = h_sc_dev; f.pub_seq = h_seq_dev; f.pub_value = ++pub_issued;
      pub_valid = true;
      pub_arm = false;
    }
    return f;
  }
  void bcgs_step(int step) {
    if (step < 0 || !comm.active()) return;
    if (pub_arm && opt.host_poll) {
      launch_publish(step, sc.p, h_sc_dev, h_seq_dev, ++pub_issued, stream);
      pub_valid = true;
      pub_arm = false;
    } else { launch_bcgs_scalars(step, sc.p, stream); pub_valid = false; }
  }
  void dots_dev(std::initializer_list<std::pair<const double *, const double *>> pairs, int step = -1) {
    ProfScope ps(prof, KC_REDUCE);
    DotArgs a;
    a.n = nred; a.nvec = 0;
    for (auto &p : pairs) { a.x[a.nvec] = p.first; a.y[a.nvec] = p.second; ++a.nvec; }
    const int nbk = launch_dots(a, partials.p, stream);
    finalize(nbk, a.nvec, partials.p, 0, step);
    if (comm.active()) comm.allreduce_sum(sc.p + S_SUM0, a.nvec, stream);
    bcgs_step(step);
  }
  // The device scalars on the host.  Default: the kernel that wrote them last has already published them (pub_valid), or a
  // one-block kernel does now; the host polls the sequence word in pinned memory -- no copy command in the queue, no wait for
  // the stream's completion signal.  host_poll = 0: hipMemcpyAsync + hipStreamSynchronize.
  void fetch_scalars() {
    ++n_fetch;
    pub_arm = false;
    if (!opt.host_poll && !comm.comm) {
      HIP_CHECK(hipMemcpyAsync(h_sc, sc.p, S_COUNT * sizeof(double), hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
    } else {   // (with an RCCL communicator always: the wait is bounded, see wait_published)
      if (!pub_valid) launch_publish(-1, sc.p, h_sc_dev, h_seq_dev, ++pub_issued, stream);
      pub_valid = false;
      wait_published(pub_issued);
    }
    if (debug_fail_at_fetch > 0 && --debug_fail_at_fetch == 0) throw Error("injected failure (option debug_fail_at_fetch)");
  }
  void wait_published(unsigned long long want) {
    volatile unsigned long long *seq = h_seq;
    unsigned spins = 0;
    int idle_checks = 0;
    const bool bounded = comm.comm != nullptr && opt.collective_timeout > 0.0;
    const auto t0 = std::chrono::steady_clock::now();
    while (*seq < want) {
      __builtin_ia32_pause();
      if ((++spins & 0x3FFF) != 0) continue;
      // Across ranks the scalars sit behind an all-reduce: when a peer has failed or never arrived it never completes.  The wait
      // is bounded; the error ends this rank's call (and, in bench.py, the rank process: the launcher then stops the others)
      if (bounded && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > opt.collective_timeout) {
        comm.poisoned = true;
        throw Error("rank " + std::to_string(comm.rank) + " of " + std::to_string(comm.nranks) + ": no result of a collective within " +
                    std::to_string((int)opt.collective_timeout) + " s (option collective_timeout): a peer rank has failed or never arrived");
      }
      // now and then: is the stream still alive?  An error ends the wait; a stream that has drained without the word
      // arriving (it cannot, short of a lost store) ends it after a bounded number of looks
      const hipError_t e = hipStreamQuery(stream);
      if (e == hipErrorNotReady) continue;
      if (e != hipSuccess) throw Error(std::string("stream failed while waiting for published scalars: ") + hipGetErrorString(e));
      if (*seq >= want) break;
      if (++idle_checks > 2000) throw Error("published scalars never arrived (sequence word not updated)");
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  void dots_host(std::initializer_list<std::pair<const double *, const double *>> pairs, double *out) {
    pub_arm = true;
    dots_dev(pairs);
    fetch_scalars();
    int j = 0;
    for (auto it = pairs.begin(); it != pairs.end(); ++it, ++j) out[j] = h_sc[S_SUM0 + j];
  }
  double norm_host(const double *x) {
    double s;
    dots_host({{x, x}}, &s);
    return std::sqrt(s);
  }
  void vec_copy(const double *x, double *y) { ProfScope ps(prof, KC_VEC); launch_copy(nred, x, y, stream); }
  void vec_set(double v, double *y) { ProfScope ps(prof, KC_VEC); launch_set(nred, v, y, stream); }
  void vec_waxpy(double *w, double a, const double *x, const double *y) {
    ProfScope ps(prof, KC_VEC);
    launch_waxpy(nred, w, sc.p, -1, 1.0, a, x, y, stream);
  }

  // =====================================================================================
  // thermal KSP path: SOEBaseStepDT_KSP (src/mpp/soe/SystemOfEquationsBaseType.F90:555-647) with
  // ThermalSOEComputeRHS / ComputeOperators (src/mpp/soe/SystemOfEquationsThermalType.F90:546-759)
  // =====================================================================================
  void thermal_upload_cells() {
    if (!th_cell_dirty) return;
    if ((int)th_dz.size() != n) throw Error("thermal soils are not set (mppgpu_thermal_set_soils)");
    PhaseTimer pt("thermal cell properties");
    // the caller's arrays go to the device as they are and are brought into the internal cell order there (a host-side permutation
    // touched a fresh gigabyte per property at 512^3: 2.6 s); properties nobody set are filled with their defaults on the device
    auto perm_d = [&](const std::vector<double> &v, DevBuf<double> &d, double dflt) {
      d.alloc(n);
      if (v.empty()) { launch_set(n, dflt, d.p, stream); return; }
      if (identity_perm) { d.upload(v, stream); HIP_CHECK(hipStreamSynchronize(stream)); return; }
      HIP_CHECK(hipMemcpyAsync(tmp_ext.p, v.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, stream));
      launch_gather(n, d_int2ext.p, tmp_ext.p, d.p, stream);
      HIP_CHECK(hipStreamSynchronize(stream));   // (tmp_ext is used again by the next property)
    };
    auto perm_c = [&](const std::vector<int> &v, DevBuf<signed char> &d, bool zero_is_one) {
      d.alloc(n);
      if (v.empty()) { HIP_CHECK(hipMemsetAsync(d.p, zero_is_one ? 1 : 0, (size_t)n, stream)); return; }
      int *stage = reinterpret_cast<int *>(tmp_ext.p);   // (n ints fit in n doubles)
      HIP_CHECK(hipMemcpyAsync(stage, v.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, stream));
      launch_gather_flags(n, d_int2ext.p, stage, d.p, zero_is_one ? 1 : 0, stream);
      HIP_CHECK(hipStreamSynchronize(stream));
    };
    perm_d(th_dz, d_th_dz, 0.0); perm_d(th_por, d_th_por, 0.0); perm_d(th_tkmg, d_th_tkmg, 0.0); perm_d(th_tkdry, d_th_tkdry, 0.0);
    perm_d(th_csol, d_th_csol, 0.0); perm_d(th_liq, d_th_liq, 0.0); perm_d(th_ice, d_th_ice, 0.0); perm_d(th_snow, d_th_snow, 0.0);
    perm_d(th_tuning, d_th_tuning, 1.0);
    perm_c(th_itype, d_th_itype, false); perm_c(th_shallow, d_th_shallow, false); perm_c(th_nsnow, d_th_nsnow0, true);
    HIP_CHECK(hipStreamSynchronize(stream));
    th_cell_dirty = false;
  }
  ThermalArgs thermal_args() const {
    ThermalArgs a;
    a.n = n; a.T = soln_prev.p;
    a.liq = d_th_liq.p; a.ice = d_th_ice.p; a.snow_water = d_th_snow.p; a.tuning = d_th_tuning.p; a.dz = d_th_dz.p; a.vol = d_vol.p;
    a.por = d_th_por.p; a.tkmg = d_th_tkmg.p; a.tkdry = d_th_tkdry.p; a.csol = d_th_csol.p;
    a.itype = d_th_itype.p; a.shallow = d_th_shallow.p; a.nsnow_zero = d_th_nsnow0.p;
    a.active = any_inactive ? d_active.p : nullptr;
    a.therm_cond = d_therm_cond.p; a.heat_cap = d_heat_cap.p;
    a.cup = d_cup.p; a.cdn = d_cdn.p; a.rc_ptr = d_rc_ptr.p; a.rc_conn = d_rc_conn.p; a.rc_slot = d_rc_slot.p;
    a.diag_slot = d_diag.p; a.rowptr = d_rowptr.p;
    a.cdu = d_cdu.p; a.cdd = d_cdd.p; a.carea = d_carea.p;
    a.src = has_src ? d_src.p : nullptr;
    a.rhs = F.p; a.val = d_val.p; a.dt = dtime; a.cnfac = cnfac;
    return a;
  }
  void thermal_assemble(int mode) {  // 0 rhs, 1 operator, 2 both
    if (!thermal) throw Error("problem_type is not thermal");
    need_setup(true);
    thermal_upload_cells();
    if (dtime <= 0.0) throw Error("time step not set");
    halo_exchange(soln_prev.p);
    ProfScope ps(prof, mode == 0 ? KC_RESIDUAL : KC_JACOBIAN);
    ThermalArgs a = thermal_args();
    launch_thermal_aux(a, stream);
    launch_thermal_rows(mode, a, stream);
    ThermalBcArgs b;
    b.t = a; b.ncond_cells = ncond_cells; b.cc_cell = d_cc_cell.p; b.cc_ptr = d_cc_ptr.p; b.type = d_bc_type.p;
    b.du = d_bc_du.p; b.dd = d_bc_dd.p; b.area = d_bc_rawarea.p; b.value = d_bc_value.p; b.frac = d_bc_frac.p;
    b.dhsdT = d_bc_dhsdT.p; b.bc_active = d_bc_act.p;
    b.reference_defects = opt.thermal_reference_defects;
    if (b.reference_defects) b.area_dirichlet_rhs = d_bc_stalearea.p;
    launch_thermal_bc(mode, b, stream);
    if (mode != 0) pc_ready = false;
  }
  void thermal_step_dt(double dt, int *converged, int *its) {
    dtime = dt;
    thermal_assemble(2);
    int it = 0;
    const int reason = ksp_solve(F.p, soln.p, &it);
    *converged = reason >= 0;
    if (its) *its = it;
    cum_linear += it;
    vec_copy(soln.p, soln_prev.p);
    HIP_CHECK(hipStreamSynchronize(stream));
  }

  // =====================================================================================
  // Krylov solvers (PETSc v3.16.2 KSPSolve_BCGS / KSPSolve_GMRES restated), zero initial guess, left PC,
  // preconditioned residual norm, KSPConvergedDefault.  reasons: 2 RTOL, 3 ATOL, 4 ITS; -3 ITS, -4 DTOL, -5 BREAKDOWN, -9 NAN
  // =====================================================================================
  struct KspConv { double rtol, atol, dtol, ttol = 0, rnorm0 = 0; };
  static int ksp_converged(KspConv &c, int it, double rnorm) {
    if (it == 0) { c.rnorm0 = rnorm; c.ttol = std::max(c.rtol * rnorm, c.atol); }
    if (!(rnorm == rnorm) || std::isinf(rnorm)) return -9;
    if (rnorm <= c.ttol) return rnorm < c.atol ? 3 : 2;
    if (rnorm >= c.dtol * c.rnorm0) return -4;
    return 0;
  }

  int ksp_bcgs(const double *b, double *x, int *its) {
    // First iteration specialised (same arithmetic, fewer passes over memory): with a zero initial guess
    // r0 = rp = M^{-1} b, p0 = r0 (beta*p and omega*beta*v vanish) and rho0 = (r0,r0) is the squared norm
    // already reduced for the convergence test, so r0 is written once into RP and aliased as R and P.
    KspConv cv{opt.ksp_rtol, opt.ksp_atol, opt.ksp_dtol};
    // Across ranks the initial (rp,rp) is not needed (rho, the iteration-0 test) before the first (v,rp) is: both travel in
    // one all-reduce, (rp,rp) parked in sc[S_SUM0+2] meanwhile.
    const bool merge0 = comm.active() && fused_pc_dots() && opt.defer_test0;
    // rp = M^{-1} b by back substitution of the right-hand side the assembly pass forward-eliminated AND v = M^{-1} A rp in one
    // kernel (k_tri_solve_lds<..., 2>): (v,rp), (v,v), (rp,rp) come out together, one scalar step (62) runs both recurrences
    const bool first_fused = opt.fused_first && fused_spmv_pc() && b == fwd_rhs && opt.defer_test0;
    // The line search's F . (J Y) without a product pass of its own (round 5, option ls_fold): a solve that ends with its first iteration
    // returns Y = alpha p + omega s, and the two fused kernels of that iteration hold the rows of J p and J s in registers -- they
    // reduce F . (J p) and F . (J s) on the way (one more 8-byte stream each), the update kernel reduces Y . Y, and the host forms
    // alpha F.(J p) + omega F.(J s): equal to PETSc's MatMult + VecDot up to the rounding of the recombination.  Only where the search's
    // reductions would have been issued behind this iteration anyway (ls_ahead) and ||X|| is known.
    const bool fold_th = th && th_fused_spmv_tri() && opt.ls_fold && !comm.active() && ls_ahead.armed && ls_ahead.X == nullptr && b == ls_ahead.F;
    const bool fold_ls = (first_fused && opt.ls_fold && opt.fused_waxpy && ls_ahead.armed && ls_ahead.X == nullptr && b == ls_ahead.F) || fold_th;
    if (first_fused) {
      const int nv1 = fold_ls ? 4 : 3;
      {
        ProfScope ps(prof, KC_SPMV_PC);
        const Fold fd = make_fold(tri_solve_lds_blocks(ncols), nv1, 0, 62);
        const int nbk = launch_back_spmv_tri_lds(tri_args(), fwd_elim.p, RP.p, V.p, col_partials.p, tri_scratch.p, stream, fd,
                                                 fold_ls ? ls_ahead.F : nullptr);
        if (!fd.counter) finalize(nbk, nv1, col_partials.p, 0, 62);
      }
      if (comm.active()) {
        comm.allreduce_sum(sc.p + S_SUM0, nv1, stream);
        launch_bcgs_scalars(62, sc.p, stream);
        pub_valid = false;
      }
    } else if (merge0) op_pc_apply_dots(b, RP.p, nullptr, 2, 0);
    else op_pc_apply_dots(b, RP.p, nullptr, 0, -1, 10);   // (rp,rp) in sc[S_SUM0]; scalar steps 0 and 1 (rho = (rp,rp), beta)
    *its = 0;
    // The convergence test of iteration 0 needs ||rp|| on the host.  Rather than stall the queue for it, the first
    // iteration is issued at once and the test is made when its scalars come back: had it fired, the answer is the zero
    // initial guess (restored below) and nothing else of the speculative iteration is visible.
    bool test0_pending = true;
    double dp = 0.0;
    int reason = 0;
    if (!opt.defer_test0) {
      fetch_scalars();
      test0_pending = false;
      reason = ksp_converged(cv, 0, h_sc[S_DP0]);
      if (reason) { vec_set(0.0, x); return reason; }
    }
    const double *Rcur = RP.p, *Pcur = RP.p;
    int i = 0;
    bool lazy_r = false;
    do {
      // rho = (r,rp): the initial (rp,rp) at i == 0; a separate reduction at i == 1 (the first update does not read rp:
      // most solves of the column-structured problems end there); afterwards the update kernel reduces it on the way
      // (the scalar step that follows, rho and beta, rides with the reduction that produces rho: see the end of the loop for i >= 2)
      if (i == 1) dots_dev({{Rcur, RP.p}}, 1);
      if (i > 0) {
        ProfScope ps(prof, KC_VEC);
        launch_bcgs_p(nred, Pv.p, Pcur, Rcur, V.p, sc.p, stream);
        Pcur = Pv.p;
      }
      // v = M^{-1} A p and t = M^{-1} A s: SpMV + column solve, one kernel each where the grid allows it (op_BA_dots)
      if (first_fused && i == 0) {
        // v, alpha: done by the solve's first kernel
      } else if (merge0 && i == 0) {
        op_BA_dots(Pcur, V.p, RP.p, 0, 3);   // (v,rp), (v,v) and the parked (rp,rp) in one all-reduce
        launch_bcgs_scalars(6, sc.p, stream);
        launch_bcgs_scalars(2, sc.p, stream);
        pub_valid = false;
      } else op_BA_dots(Pcur, V.p, RP.p, 0, -1, 2, (fold_th && i == 0) ? ls_ahead.F : nullptr);   // (v,rp) in sc[S_SUM0]  [(v,v) in S_SUM0+1 is not used]; alpha
      int end_step = 4;
      const bool fold_now = fold_ls && i == 0;
      if (opt.fused_waxpy && fused_spmv_pc()) {   // s = r - alpha v, t = M^{-1} A s, (t,s), (t,t): one kernel
        const int nv2 = fold_now ? 3 : 2;
        {
          ProfScope ps(prof, KC_SPMV_PC);
          const Fold fd = make_fold(tri_solve_lds_blocks(ncols), nv2, 0, 3);
          const int nbk = launch_waxpy_spmv_tri_lds(tri_args(), Rcur, V.p, sc.p, S.p, T.p, col_partials.p, tri_scratch.p, stream, fd,
                                                    fold_now ? ls_ahead.F : nullptr);
          if (!fd.counter) finalize(nbk, nv2, col_partials.p, 0, 3);
        }
        if (comm.active()) comm.allreduce_sum(sc.p + S_SUM0, nv2, stream);
        bcgs_step(3);
      } else {
        { ProfScope ps(prof, KC_VEC); launch_waxpy(nred, S.p, sc.p, S_ALPHA, -1.0, 0.0, V.p, Rcur, stream); }
        op_BA_dots(S.p, T.p, S.p, 0, -1, 3, (fold_th && i == 0) ? ls_ahead.F : nullptr);    // (t,s), (t,t) in sc[S_SUM0], sc[S_SUM0+1]; omega
      }
      {
        ProfScope ps(prof, KC_REDUCE);
        const bool wide = opt.wide_xr && nred >= (1 << 22);
        double *pbuf = partials.p;
        if (wide) {
          const size_t nb = (size_t)bcgs_xr_blocks(nred, true), need = 2 * (nb + nb / 2048 + 1);
          if (xr_partials.n < need) xr_partials.alloc(need);
          pbuf = xr_partials.p;
        }
        // r = s - omega t is written only if somebody will read it: where the solve is expected to end with this iteration (the
        // line search's reductions are already queued behind it) only its norm is formed; a solve that goes on forms r then
        lazy_r = opt.lazy_r && ls_ahead.armed && i == 0;
        const int nbk = launch_bcgs_xr(nred, i == 0, wide, x, lazy_r ? nullptr : R.p, Pcur, S.p, T.p, RP.p, sc.p, pbuf, stream, fold_now);
        // end of the iteration: dp, rhoold, omegaold (step 4) and, from the second iteration on, the next one's rho = (r,rp)
        // and beta (step 5) -- harmless when the test below ends the solve
        end_step = i == 0 ? 4 : 45;
        // the last writer of the scalars before the read-back publishes them (folded line-search sums without a speculative trial
        // point: this reduction is the last one)
        pub_arm = !(ls_ahead.armed && i == 0) || (fold_now && !ls_ahead.spec);
        finalize(nbk, (i == 0 && !fold_now) ? 1 : 2, pbuf, 0, end_step);
        if (!(ls_ahead.armed && i == 0) && comm.active()) comm.allreduce_sum(sc.p + S_SUM0, i == 0 ? 1 : 2, stream);
      }
      const bool ahead = ls_ahead.armed && i == 0;
      int ahead_nv = 0;
      const bool spec = ahead && ls_ahead.spec;
      if (ahead && fold_now) {   // ||r||^2 and Y . Y are in S_SUM0, S_SUM0 + 1; F . (J p), F . (J s) in S_FJP, S_FJS (scalar steps 62, 3)
        ahead_nv = 1;
        if (!spec && comm.active()) comm.allreduce_sum(sc.p + S_SUM0, 2, stream);
      } else if (ahead) {   // x is final if we stop here
        const int nv = ls_ahead.X ? 3 : 2;
        pub_arm = !spec;
        ahead_nv = op_spmv_dots_launch(x, ls_ahead.F, ls_ahead.X, 1, spec ? 0 : 1 + nv);
      }
      if (spec) {    // ... and so is the line search's first trial point X - x
        ls_point = Trial{ls_ahead.Xv, x, -1.0};
        ls_ahead.spec_fused = opt.fused_assembly && opt.fused_trials > 0 && !(opt.fused_predict && predict_backtrack());
        asm_slot = 1 + ahead_nv; asm_defer = true;
        if (ls_ahead.spec_fused) op_residual_jacobian(ls_ahead.Wv, ls_ahead.Gv); else op_residual(ls_ahead.Wv, ls_ahead.Gv);
        asm_slot = 0; asm_defer = false;
        ls_ahead.spec_issued = true;
        if (comm.active()) comm.allreduce_sum(sc.p + S_SUM0, 1 + ahead_nv + 2, stream);   // ||r||^2, the search's sums, the trial point's
      }
      Rcur = R.p;
      if (comm.active()) pub_arm = true;
      bcgs_step(end_step);
      fetch_scalars();
      auto leave = [&](int r) { spec_void(); return r; };   // every exit but the one that hands the trial point to the line search
      if (test0_pending) {
        test0_pending = false;
        reason = ksp_converged(cv, 0, h_sc[S_DP0]);
        if (reason) { vec_set(0.0, x); return leave(reason); }
      }
      const double rho = h_sc[S_RHOOLD], d1 = h_sc[S_D1];   // this iteration's rho (step 4 has just moved it there)
      if (d1 == 0.0) return leave(-5);
      if (h_sc[S_FLAG] != 0.0) {  // t == 0: x += alpha p was applied (omega forced to 0)
        const double ss = std::sqrt(h_sc[S_SUM0]);  // r == s here
        ++*its;
        return leave(ss != 0.0 ? -5 : 2);
      }
      dp = h_sc[S_DP];
      ++*its;
      reason = ksp_converged(cv, i + 1, dp);
      if (reason > 0 && ahead) {
        ls_ahead.done = true;
        for (int j = 0; j < ahead_nv; ++j) ls_ahead.r3[j] = h_sc[S_SUM0 + 1 + j];
        if (fold_now) { ls_ahead.r3[1] = h_sc[S_ALPHA] * h_sc[S_FJP] + h_sc[S_OMEGA] * h_sc[S_FJS]; ls_ahead.r3[2] = 0.0; ++ls_folds; }
        if (ls_ahead.spec_issued) {
          ls_ahead.spec_issued = false;
          ls_ahead.spec_done = true;
          ls_ahead.g2 = h_sc[S_SUM0 + 1 + ahead_nv]; ls_ahead.w2 = h_sc[S_SUM0 + 2 + ahead_nv];
        }
        return reason;
      }
      spec_void();   // the solve goes on (or has failed): the trial point is void and the matrix is needed again
      if (reason) return reason;
      if (rho == 0.0) return -5;
      if (lazy_r) { ProfScope ps(prof, KC_VEC); launch_waxpy(nred, R.p, sc.p, S_OMEGA, -1.0, 0.0, T.p, S.p, stream); }   // r = s - omega t after all
      ++i;
    } while (i < opt.ksp_max_it);
    return -3;
  }

  // a speculative trial point that is not going to be used: its function evaluation does not count, and the Jacobian, the
  // column factors and L^{-1} F it replaced are assembled again at X
  void spec_void() {
    if (!ls_ahead.spec_issued) return;
    ls_ahead.spec_issued = false;
    --nfuncs; --cum_funcs; ++spec_misses;
    asm_sums = false;
    if (ls_ahead.spec_fused) op_jacobian(ls_ahead.Xv, const_cast<double *>(ls_ahead.F));
  }

  static constexpr int kNormSlot = 44;   // sc[S_SUM0 + 44]: beyond the <= 31 Gram-Schmidt coefficients
  int ksp_gmres(const double *b, double *x, int *its) {
    const int m = opt.gmres_restart;
    if ((int)vv.size() < m + 1) {
      vv.resize(m + 1);
      for (auto &v : vv) v.alloc(nrow);
    }
    KspConv cv{opt.ksp_rtol, opt.ksp_atol, opt.ksp_dtol};
    std::vector<double> H((size_t)(m + 1) * m, 0.0), cs(m), sn(m), rs(m + 1), y(m), hh(m + 1);
    bool x_is_zero = true;   // zero initial guess: x is first written by the update of the first cycle
    *its = 0;
    int reason = 0;
    bool first = true;
    while (!reason) {
      // r0 = M^{-1}(b - A x) and its norm; with the column solver the norm comes out of the solve (and the first solve of
      // a Newton iteration starts from the right-hand side the assembly pass already forward-eliminated)
      pub_arm = true;
      if (first) op_pc_apply_dots(b, vv[0].p, nullptr);
      else {
        op_spmv(x, Wk.p);
        vec_waxpy(Wk.p, -1.0, Wk.p, b);
        op_pc_apply_dots(Wk.p, vv[0].p, nullptr);
      }
      fetch_scalars();
      double res = std::sqrt(h_sc[S_SUM0]);
      reason = ksp_converged(cv, first ? 0 : *its, res);
      first = false;
      if (reason) break;
      if (res == 0.0) { reason = 3; break; }
      scale_host(vv[0].p, res);
      rs[0] = res;
      int it = 0;
      while (it < m && *its < opt.ksp_max_it) {
        double *vnew = vv[it + 1].p;
        const bool dots_in_solve = it == 0 && fused_pc_dots();   // (w, v0) reduced by the column solve that produces w
        if (dots_in_solve) op_BA_dots(vv[0].p, vnew, vv[0].p);
        else op_BA(vv[it].p, vnew);
        // Restart lengths above 31 (PETSc takes any, -ksp_gmres_restart): the Gram-Schmidt kernels take up to kMaxMultiDot vectors per
        // launch, so the basis is walked in chunks of that many -- ALL the dot products first (classical Gram-Schmidt reads the
        // unmodified w), parked in gs_h, then the updates in the same ascending order a single launch would apply them
        const int nvt = it + 1;
        const bool chunked = nvt > kMaxMultiDot;
        const size_t nbm = (size_t)maxpy_blocks(nred), needm = nbm + nbm / 2048 + 1;
        if (xr_partials.n < needm) xr_partials.alloc(needm);
        if (chunked) {
          if (gs_h.n < (size_t)m + 2) gs_h.alloc((size_t)m + 2);
          for (int c0 = 0; c0 < nvt; c0 += kMaxMultiDot) {
            ProfScope ps(prof, KC_REDUCE);
            DotArgs a;
            a.n = nred; a.nvec = std::min(kMaxMultiDot, nvt - c0);
            for (int j = 0; j < a.nvec; ++j) { a.x[j] = vnew; a.y[j] = vv[c0 + j].p; }
            const int nbk = launch_dots(a, partials.p, stream);
            finalize(nbk, a.nvec, partials.p);
            if (comm.active()) comm.allreduce_sum(sc.p + S_SUM0, a.nvec, stream);
            HIP_CHECK(hipMemcpyAsync(gs_h.p + c0, sc.p + S_SUM0, (size_t)a.nvec * sizeof(double), hipMemcpyDeviceToDevice, stream));
          }
          for (int c0 = 0; c0 < nvt; c0 += kMaxMultiDot) {
            ProfScope ps(prof, KC_VEC);
            MaxpyArgs ma;
            ma.n = nred; ma.nvec = std::min(kMaxMultiDot, nvt - c0);
            for (int j = 0; j < ma.nvec; ++j) ma.v[j] = vv[c0 + j].p;
            const bool last = c0 + kMaxMultiDot >= nvt;
            const int nbk = launch_maxpy_neg(ma, gs_h.p + c0, vnew, last ? xr_partials.p : nullptr, stream);   // (the parked block of this chunk)
            if (last) {
              pub_arm = true;
              finalize(nbk, 1, xr_partials.p, kNormSlot);
              if (comm.active()) comm.allreduce_sum(sc.p + S_SUM0 + kNormSlot, 1, stream);
            }
          }
          HIP_CHECK(hipMemcpyAsync(hh.data(), gs_h.p, (size_t)nvt * sizeof(double), hipMemcpyDeviceToHost, stream));   // (complete when the
          fetch_scalars();                                                                                            // published scalars are)
          HIP_CHECK(hipStreamSynchronize(stream));
        } else {
        if (!dots_in_solve) {  // classical Gram-Schmidt: all dots, then one fused update
          ProfScope ps(prof, KC_REDUCE);
          DotArgs a;
          a.n = nred; a.nvec = it + 1;
          for (int j = 0; j <= it; ++j) { a.x[j] = vnew; a.y[j] = vv[j].p; }
          const int nbk = launch_dots(a, partials.p, stream);
          finalize(nbk, a.nvec, partials.p);
          if (comm.active()) comm.allreduce_sum(sc.p + S_SUM0, a.nvec, stream);
        }
        {
          ProfScope ps(prof, KC_VEC);
          MaxpyArgs ma;
          ma.n = nred; ma.nvec = it + 1;
          for (int j = 0; j <= it; ++j) ma.v[j] = vv[j].p;
          // the update also reduces ||vnew||^2 (slot kNormSlot, clear of the h_j it reads): one pass and one read-back fewer
          const int nbk = launch_maxpy_neg(ma, sc.p + S_SUM0, vnew, xr_partials.p, stream);
          pub_arm = true;
          finalize(nbk, 1, xr_partials.p, kNormSlot);
          if (comm.active()) comm.allreduce_sum(sc.p + S_SUM0 + kNormSlot, 1, stream);
        }
        fetch_scalars();
        for (int j = 0; j <= it; ++j) hh[j] = h_sc[S_SUM0 + j];
        }
        const double tt = std::sqrt(h_sc[S_SUM0 + kNormSlot]);
        const bool hapend = tt == 0.0;   // v_{it+1} = vnew / tt is formed below, and only if another step will use it
        double *hcol = H.data() + (size_t)it * (m + 1);
        for (int j = 0; j <= it; ++j) hcol[j] = hh[j];
        hcol[it + 1] = tt;
        for (int j = 0; j < it; ++j) {
          const double t1 = hcol[j], t2 = hcol[j + 1];
          hcol[j] = cs[j] * t1 + sn[j] * t2;
          hcol[j + 1] = -sn[j] * t1 + cs[j] * t2;
        }
        const double hyp = std::sqrt(hcol[it] * hcol[it] + hcol[it + 1] * hcol[it + 1]);
        if (hyp == 0.0) { reason = -5; break; }
        cs[it] = hcol[it] / hyp; sn[it] = hcol[it + 1] / hyp;
        rs[it + 1] = -sn[it] * rs[it];
        rs[it] = cs[it] * rs[it];
        hcol[it] = cs[it] * hcol[it] + sn[it] * hcol[it + 1];
        hcol[it + 1] = 0.0;
        res = std::fabs(rs[it + 1]);
        ++it; ++*its;
        reason = ksp_converged(cv, *its, res);
        if (reason || hapend) break;
        if (it < m && *its < opt.ksp_max_it) scale_host(vnew, tt);
      }
      for (int i = it - 1; i >= 0; --i) {
        double s = rs[i];
        for (int j = i + 1; j < it; ++j) s -= H[(size_t)j * (m + 1) + i] * y[j];
        y[i] = s / H[(size_t)i * (m + 1) + i];
      }
      for (int c0 = 0; c0 < it; c0 += kMaxMultiDot) {   // x += sum_j y_j v_j, kMaxMultiDot vectors per launch, ascending j
        ProfScope ps(prof, KC_VEC);
        MaxpyValArgs ma;
        ma.n = nred; ma.nvec = std::min(kMaxMultiDot, it - c0);
        for (int j = 0; j < ma.nvec; ++j) { ma.v[j] = vv[c0 + j].p; ma.c[j] = y[c0 + j]; }
        ma.zero_x = x_is_zero;
        launch_maxpy_val(ma, x, stream);
        x_is_zero = false;
      }
      if (!reason && *its >= opt.ksp_max_it) reason = -3;
    }
    if (x_is_zero) vec_set(0.0, x);   // converged before the first step: the answer is the initial guess
    return reason;
  }
  void scale_host(double *x, double divisor) {  // x <- x / divisor
    ProfScope ps(prof, KC_VEC);
    HIP_CHECK(hipMemcpyAsync(sc.p + S_D2, &divisor, sizeof(double), hipMemcpyHostToDevice, stream));
    pub_valid = false;
    launch_scale(nred, x, sc.p, S_D2, stream);
  }

  int ksp_solve(const double *b, double *x, int *its) {
    op_pc_setup();
    switch (opt.ksp_type) {
      case 0: op_pc_apply(b, x); *its = 1; return 4;
      case 2: return ksp_bcgs(b, x, its);
      default: return ksp_gmres(b, x, its);
    }
  }

  // =====================================================================================
  // SNES newtonls + bt line search (PETSc v3.16.2), then the reference's StepDT loop
  // =====================================================================================
  int snes_converged(int it, double xnorm, double snorm, double fnorm, double &ttol, double &rnorm0) const {
    if (!it) { ttol = fnorm * opt.snes_rtol; rnorm0 = fnorm; }
    if (fnorm != fnorm) return -4;
    if (fnorm < opt.snes_atol) return 2;
    if (nfuncs >= opt.snes_max_funcs) return -2;
    if (it) {
      if (fnorm <= ttol) return 3;
      if (snorm < opt.snes_stol * xnorm) return 4;
      if (opt.snes_divtol > 0 && fnorm > opt.snes_divtol * rnorm0) return -9;
    }
    return 0;
  }

  // X (= soln), F updated in place on success; returns 0 ok / 1 failed.  Same branch structure as the oracle.
  int linesearch_bt(double *&X, double *&Fv, double *Yv, double *&Wv, double *&Gv, double &fnorm, double &xnorm,
                    double &ynorm, bool &jac_valid, bool xnorm_known) {
    const double alpha = 1e-4, maxstep = 1e8, minlambda = 1e-12;
    const int max_its = 40;
    double lambda = 1.0, lambdatemp, lambdaprev, gprev, g, gnorm = 0.0, wnorm2 = 0.0;
    double r3[3];   // initslope = F . (J Y), ||Y||, ||X||
    // ||X|| was reduced together with ||F(X)|| when X was the accepted trial point of the previous line search
    if (ls_ahead.done) {            // reduced by the Krylov solver behind its last update
      ls_ahead.done = false;
      r3[0] = ls_ahead.r3[0]; r3[1] = ls_ahead.r3[1]; r3[2] = ls_ahead.r3[2];
    } else if (spmv_dots_fused()) {
      op_spmv_dots(Yv, Fv, xnorm_known ? nullptr : X, r3);
    } else {
      op_spmv(Yv, Wv);
      if (xnorm_known) dots_host({{Yv, Yv}, {Fv, Wv}}, r3);
      else dots_host({{Yv, Yv}, {Fv, Wv}, {X, X}}, r3);
    }
    if (!xnorm_known) xnorm = std::sqrt(r3[2]);
    ynorm = std::sqrt(r3[0]);
    double initslope = r3[1];
    // the full step's trial point was evaluated behind the Krylov solve (LsAhead::spec); it is void where PETSc would not
    // have evaluated it, or not there (the evaluation is then not counted; J(X) is dead by now either way)
    bool spec = ls_ahead.spec_done;
    ls_ahead.spec_done = false;
    if (spec && (ynorm == 0.0 || ynorm > maxstep || nfuncs - 1 >= opt.snes_max_funcs)) { spec = false; --nfuncs; --cum_funcs; }
    if (ynorm == 0.0) return 1;
    if (ynorm > maxstep) {
      vec_waxpy(Yv, maxstep / ynorm - 1.0, Yv, Yv);  // Y *= maxstep/ynorm
      initslope *= maxstep / ynorm;
      ynorm = maxstep;
    }
    const double f = fnorm * fnorm;
    if (initslope > 0.0) initslope = -initslope;
    if (initslope == 0.0) initslope = -1.0;
    // the first trial point is usually accepted: evaluate F and J there together; later (backtracking) trials
    // evaluate F only.  J(X) is dead once initslope is known, so the trial may overwrite the matrix values.
    int trial = 0;
    bool last_fused = false;
    jac_valid = false;
    auto eval = [&](double lam) {
      if (spec) {   // trial 0, lambda = 1: already evaluated (into Wv, Gv), its sums came with the Krylov solver's read-back
        spec = false;
        last_fused = ls_ahead.spec_fused;
        ls_first_fused = last_fused;
        ++trial;
        gnorm = std::sqrt(ls_ahead.g2); wnorm2 = ls_ahead.w2;
        return gnorm * gnorm;
      }
      // W = X - lam Y: formed inside the assembly pass when the column march finishes the rows itself
      if (rows_complete()) ls_point = Trial{X, Yv, -lam}; else vec_waxpy(Wv, -lam, Yv, X);
      // a trial point's Jacobian is assembled together with its residual when the point is likely to be accepted:
      // the full step unless the previous line search of this solve had to backtrack, and always the second trial
      last_fused = opt.fused_assembly && trial < opt.fused_trials && !(trial == 0 && opt.fused_predict && predict_backtrack());
      if (trial == 0) ls_first_fused = last_fused;
      ++trial;
      if (last_fused) op_residual_jacobian(Wv, Gv); else op_residual(Wv, Gv);
      double r2[2];
      residual_sums(Wv, Gv, r2);
      gnorm = std::sqrt(r2[0]); wnorm2 = r2[1];
      return gnorm * gnorm;
    };
    for (;;) {
      if (!spec && nfuncs >= opt.snes_max_funcs) return 1;
      g = eval(lambda);
      if (std::isfinite(g)) break;
      if (lambda <= minlambda) return 1;
      lambda = .5 * lambda;
    }
    if (!(.5 * g <= .5 * f + lambda * alpha * initslope)) {
      if (opt.snes_stol * xnorm > ynorm) return 0;  // tiny step, inadequate reduction: leave X, F untouched
      lambdatemp = -initslope / (g - f - 2.0 * lambda * initslope);
      lambdaprev = lambda; gprev = g;
      if (lambdatemp > .5 * lambda) lambdatemp = .5 * lambda;
      if (lambdatemp <= .1 * lambda) lambda = .1 * lambda; else lambda = lambdatemp;
      if (nfuncs >= opt.snes_max_funcs) return 1;
      g = eval(lambda);
      if (!std::isfinite(g)) return 1;
      if (!(.5 * g < .5 * f + lambda * alpha * initslope)) {
        int count;
        for (count = 0; count < max_its; ++count) {
          if (lambda <= minlambda) return 1;
          const double t1 = .5 * (g - f) - lambda * initslope, t2 = .5 * (gprev - f) - lambdaprev * initslope;
          const double a = (t1 / (lambda * lambda) - t2 / (lambdaprev * lambdaprev)) / (lambda - lambdaprev);
          const double b = (-lambdaprev * t1 / (lambda * lambda) + lambda * t2 / (lambdaprev * lambdaprev)) / (lambda - lambdaprev);
          double d = b * b - 3 * a * initslope;
          if (d < 0.0) d = 0.0;
          if (a == 0.0) lambdatemp = -initslope / (2.0 * b); else lambdatemp = (-b + std::sqrt(d)) / (3.0 * a);
          lambdaprev = lambda; gprev = g;
          if (lambdatemp > .5 * lambda) lambdatemp = .5 * lambda;
          if (lambdatemp <= .1 * lambda) lambda = .1 * lambda; else lambda = lambdatemp;
          if (nfuncs >= opt.snes_max_funcs) return 1;
          g = eval(lambda);
          if (!std::isfinite(g)) return 1;
          if (.5 * g < .5 * f + lambda * alpha * initslope) break;
        }
        if (count == max_its) return 1;
      }
    }
    ls_backtracked_last = trial > 1;
    bt_hist_cur.push_back(trial > 1);
    if (opt.ls_trace) std::fprintf(stderr, "[mppgpu ls] newton %d trials %d lambda %.6g first_fused %d\n", ls_trace_it, trial, lambda, (int)ls_first_fused);
    std::swap(X, Wv);   // X <- W, F <- G by pointer exchange (no copies)
    std::swap(Fv, Gv);
    jac_valid = last_fused;
    xnorm = std::sqrt(wnorm2);
    fnorm = gnorm;
    return 0;
  }

  // ---- the other line searches SOEBaseStepDT_SNES can select (SystemOfEquationsBaseType.F90:432-475): PETSc v3.16.2
  // SNESLineSearchApply_Basic / _L2 / _CP restated from the published algorithms (defaults: damping 1, steptol 1e-12,
  // maxstep 1e8, rtol 1e-8, atol 1e-15, max_its 1, cp order linear); same statement order as the oracle's.  They always move
  // X; l2 and cp report a failure when the step length ends at or below steptol.  X, F change by pointer exchange.
  double trial_norm2(double *Wv, double *Gv) {   // ||F(W)||^2 the way PETSc forms it: VecNorm, then squared
    op_residual(Wv, Gv);
    double r2[2];
    residual_sums(Wv, Gv, r2);
    const double nr = std::sqrt(r2[0]);
    return nr * nr;
  }
  // X <- W, F <- F(X) (evaluated again, as PETSc does), SNESLineSearchComputeNorms
  void ls_finish(double *&X, double *&Fv, double *Yv, double *&Wv, double &fnorm, double &xnorm, double &ynorm, bool &jac_valid) {
    std::swap(X, Wv);
    if (opt.fused_assembly) { op_residual_jacobian(X, Fv); jac_valid = true; } else { op_residual(X, Fv); jac_valid = false; }
    double r2[2], yy;
    residual_sums(X, Fv, r2);
    dots_host({{Yv, Yv}}, &yy);
    fnorm = std::sqrt(r2[0]); xnorm = std::sqrt(r2[1]); ynorm = std::sqrt(yy);
    ls_ynorm = ynorm;
  }
  int linesearch_basic(double *&X, double *&Fv, double *Yv, double *&Wv, double *&Gv, double &fnorm, double &xnorm,
                       double &ynorm, bool &jac_valid) {
    vec_waxpy(Wv, -1.0, Yv, X);
    if (opt.fused_assembly) { op_residual_jacobian(Wv, Gv); jac_valid = true; } else { op_residual(Wv, Gv); jac_valid = false; }
    double r2[2], yy;
    residual_sums(Wv, Gv, r2);
    dots_host({{Yv, Yv}}, &yy);
    fnorm = std::sqrt(r2[0]); xnorm = std::sqrt(r2[1]); ynorm = std::sqrt(yy);
    ls_ynorm = ynorm;
    std::swap(X, Wv);
    std::swap(Fv, Gv);
    return 0;
  }
  int linesearch_l2(double *&X, double *&Fv, double *Yv, double *&Wv, double *&Gv, double &fnorm, double &xnorm,
                    double &ynorm, bool &jac_valid) {
    const double steptol = 1e-12;
    double maxstep = 1e8;
    const int max_its = 1;
    double lambda = 1.0, lambda_old = 0.0, lambda_mid, lambda_update, delLambda;
    double fnrm = 0.0, fnrm_old = fnorm * fnorm, fnrm_mid = 0.0, delFnrm, delFnrm_old, del2Fnrm;
    lambda_mid = 0.5 * (lambda + lambda_old);
    jac_valid = false;
    for (int i = 0; i < max_its; ++i) {
      for (;;) {
        vec_waxpy(Wv, -lambda_mid, Yv, X);
        fnrm_mid = trial_norm2(Wv, Gv);
        vec_waxpy(Wv, -lambda, Yv, X);
        fnrm = trial_norm2(Wv, Gv);
        if (std::isfinite(fnrm)) break;
        if (lambda <= steptol) {
          double r2[2];
          dots_host({{X, X}, {Yv, Yv}}, r2);
          xnorm = std::sqrt(r2[0]); ynorm = std::sqrt(r2[1]);
          return 1;
        }
        maxstep = .95 * lambda;
        lambda = .5 * (lambda + lambda_old);
        lambda_mid = .5 * (lambda + lambda_old);
      }
      delLambda = lambda - lambda_old;
      delFnrm = (3. * fnrm - 4. * fnrm_mid + 1. * fnrm_old) / delLambda;
      delFnrm_old = (-3. * fnrm_old + 4. * fnrm_mid - 1. * fnrm) / delLambda;
      del2Fnrm = (delFnrm - delFnrm_old) / delLambda;
      if (del2Fnrm > 0.) lambda_update = lambda - delFnrm / del2Fnrm;
      else if (del2Fnrm < 0.) lambda_update = lambda + delFnrm / del2Fnrm;
      else break;
      if (lambda_update < steptol) lambda_update = 0.5 * (lambda + lambda_old);
      if (!std::isfinite(lambda_update)) break;
      if (lambda_update > maxstep) break;
      lambda_old = lambda;
      lambda = lambda_update;
      fnrm_old = fnrm;
      lambda_mid = 0.5 * (lambda + lambda_old);
    }
    vec_waxpy(Wv, -lambda, Yv, X);
    ls_finish(X, Fv, Yv, Wv, fnorm, xnorm, ynorm, jac_valid);
    return lambda <= steptol ? 1 : 0;
  }
  int linesearch_cp(double *&X, double *&Fv, double *Yv, double *&Wv, double *&Gv, double &fnorm, double &xnorm,
                    double &ynorm, bool &jac_valid, int &snes_reason) {
    const double steptol = 1e-12, maxstep = 1e8, rtol = 1e-8, atol = 1e-15;
    const int max_its = 1;
    double lambda = 1.0, lambda_old = 0.0, lambda_update, delLambda, fty, fty_init, fty_old, s;
    const double ynorm_in = ls_ynorm;
    jac_valid = false;
    dots_host({{Fv, Yv}}, &fty_old);
    if (std::fabs(fty_old) < atol * ynorm_in) {
      snes_reason = 2;
      double xx;
      dots_host({{X, X}}, &xx);
      xnorm = std::sqrt(xx); ynorm = ynorm_in;
      return 0;
    }
    fty_init = fty_old;
    for (int i = 0; i < max_its; ++i) {
      vec_waxpy(Wv, -lambda, Yv, X);
      op_residual(Wv, Gv);
      dots_host({{Gv, Yv}}, &fty);
      delLambda = lambda - lambda_old;
      if (std::fabs(delLambda) < steptol * lambda) break;
      if (std::fabs(fty) / std::fabs(fty_init) < rtol) break;
      if (std::fabs(fty) < atol * ynorm_in && i > 0) break;
      s = (fty - fty_old) / delLambda;
      if (s > 0.) s = -s;
      if (s == 0.0) break;
      lambda_update = lambda - fty / s;
      if (lambda_update < steptol) lambda_update = lambda + fty / s;
      if (!std::isfinite(lambda_update)) break;
      if (lambda_update > maxstep) break;
      lambda_old = lambda;
      lambda = lambda_update;
      fty_old = fty;
    }
    vec_waxpy(Wv, -lambda, Yv, X);
    ls_finish(X, Fv, Yv, Wv, fnorm, xnorm, ynorm, jac_valid);
    return lambda <= steptol ? 1 : 0;
  }

  int snes_solve(int *newton_its, int *linear_its) {
    // working pointers; `X` must end up in soln
    double *X = soln.p, *Fv = F.p, *Wv = W.p, *Gv = G.p, *Yv = Y.p;
    double ttol = 0, rnorm0 = 0, xnorm = 0, ynorm = 0;
    nfuncs = 0;
    ls_backtracked_last = false;
    bt_hist_cur.clear();
    *newton_its = 0; *linear_its = 0;
    bool jac_valid = false;
    if (opt.fused_assembly) { op_residual_jacobian(X, Fv); jac_valid = true; } else op_residual(X, Fv);
    double fnorm;
    bool xnorm0_known = false;   // ||X|| of the first iteration's line search from the assembly pass that also reduced ||F|| (option ls_fold)
    if (asm_sums) {
      double r2[2];
      residual_sums(X, Fv, r2);
      fnorm = std::sqrt(r2[0]);
      if (opt.ls_fold) { xnorm = std::sqrt(r2[1]); xnorm0_known = true; }
    } else fnorm = norm_host(Fv);
    int reason = snes_converged(0, 0.0, 0.0, fnorm, ttol, rnorm0);
    for (int i = 0; !reason && i < opt.snes_max_it; ++i) {
      if (!jac_valid) op_jacobian(X, Fv);
      int its = 0;
      ls_trace_it = i;
      ls_ahead.armed = opt.ls_ahead && ksp_one_shot && opt.ksp_type == 2 && spmv_dots_fused() && opt.linesearch_type == 2;
      ls_ahead.done = false; ls_ahead.spec_done = false; ls_ahead.spec_issued = false;
      ls_ahead.F = Fv; ls_ahead.X = (i > 0 || xnorm0_known) ? nullptr : X;
      ls_ahead.spec = ls_ahead.armed && opt.ls_spec && rows_complete();
      ls_ahead.Xv = X; ls_ahead.Wv = Wv; ls_ahead.Gv = Gv;
      const int kreason = ksp_solve(Fv, Yv, &its);
      ls_ahead.armed = false;
      ksp_one_shot = kreason > 0 && its == 1;
      *linear_its += its;
      if (kreason < 0) { reason = -3; break; }
      int lsfail, ls_reason = 0;
      ls_trace_it = i;
      switch (opt.linesearch_type) {
        case 1: lsfail = linesearch_basic(X, Fv, Yv, Wv, Gv, fnorm, xnorm, ynorm, jac_valid); break;
        case 3: lsfail = linesearch_l2(X, Fv, Yv, Wv, Gv, fnorm, xnorm, ynorm, jac_valid); break;
        case 4: lsfail = linesearch_cp(X, Fv, Yv, Wv, Gv, fnorm, xnorm, ynorm, jac_valid, ls_reason); break;
        default: lsfail = linesearch_bt(X, Fv, Yv, Wv, Gv, fnorm, xnorm, ynorm, jac_valid, i > 0 || xnorm0_known); ls_ynorm = ynorm; break;
      }
      if (ls_reason) { reason = ls_reason; break; }                 // if (snes->reason) break;
      if (!std::isfinite(fnorm)) { reason = -4; break; }            // SNESCheckFunctionNorm
      if (lsfail) {
        reason = (opt.snes_stol * xnorm > ynorm) ? 4 : -6;
        break;
      }
      *newton_its = i + 1;
      reason = snes_converged(i + 1, xnorm, ynorm, fnorm, ttol, rnorm0);
    }
    if (!reason) reason = -5;
    if (reason > 0) bt_hist_prev = bt_hist_cur;   // (a converged solve: what the next one is predicted from)
    if (X != soln.p) {  // an odd number of pointer swaps: move the iterate back into soln
      vec_copy(X, soln.p);
    }
    return reason;
  }

  // SetDataInSOEAuxVar in PostSolve (GE:1150-1166, 1338-1342): mass that crossed the horizontal internal connections and
  // the boundary connections during the sub-step, from the fluxes at the accepted state
  void accumulate_mass_exchange(double dt_sub) {
    if (lat_mass_exc.empty()) {
      lat_mass_exc.assign(n, 0.0); bnd_mass_exc.assign(nb, 0.0);
      conn_horizontal.assign(nc, 0);
      for (int c = 0; c < nc; ++c) {
        const double ux = cuv[3 * (size_t)c], uy = cuv[3 * (size_t)c + 1], uz = cuv[3 * (size_t)c + 2];
        conn_horizontal[c] = !(std::fabs(uz) > 0.5 * std::sqrt(ux * ux + uy * uy + uz * uz));
      }
    }
    if (nb > 0) {
      launch_bc(4, bc_args(soln.p, F.p), stream);
      std::vector<double> fl(nb);
      HIP_CHECK(hipMemcpyAsync(fl.data(), d_bc_flux.p, nb * sizeof(double), hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      for (int g = 0; g < nb; ++g) bnd_mass_exc[g] = bnd_mass_exc[g] + fl[bc_pos_of_g[g]] * dt_sub;
    }
    bool any_h = false;
    for (char h : conn_horizontal) any_h |= (h != 0);
    if (any_h) {
      DevBuf<double> fl;
      fl.alloc(nc_int);
      launch_conn_flux(cells(soln.p), conns(), d_cup.p, d_cdn.p, nc_int, fl.p, stream);
      std::vector<double> h(nc_int);
      HIP_CHECK(hipMemcpyAsync(h.data(), fl.p, (size_t)nc_int * sizeof(double), hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      for (int c = 0; c < nc; ++c)
        if (conn_horizontal[c]) {
          const double f = h[cext2int[c]] * dt_sub;
          lat_mass_exc[cup_e[c]] = lat_mass_exc[cup_e[c]] + f;
          lat_mass_exc[cdn_e[c]] = lat_mass_exc[cdn_e[c]] - f;
        }
    }
  }
  // soe%PreStepDT / PostStepDT of ELM's calling sequence (SystemOfEquationsVSFMType.F90:892-940; GE:2773-2785)
  void pre_step_dt() {
    need_setup(true);
    vec_copy(soln_prev_clm.p, soln_prev.p);
    vec_copy(soln_prev_clm.p, soln.p);
    std::fill(lat_mass_exc.begin(), lat_mass_exc.end(), 0.0);
    std::fill(bnd_mass_exc.begin(), bnd_mass_exc.end(), 0.0);
    HIP_CHECK(hipStreamSynchronize(stream));
  }
  void post_step_dt() {
    need_setup(true);
    vec_copy(soln_prev.p, soln_prev_clm.p);
    HIP_CHECK(hipStreamSynchronize(stream));
  }

  void step_dt(double dt, int *converged, int *converged_reason) {
    ensure_ready();
    const int max_num_time_cuts = 20;
    int num_time_cuts = 0, linesearch_iter = 0;
    const double target_time = dt;
    double dt_iter = dt;
    // the line searches tried, in this order, before the time step is cut (SystemOfEquationsBaseType.F90:432-453)
    const int ls_default = opt.linesearch_type;
    int ls_types[3] = {ls_default, 3, 2}, max_linesearch_iter = 3;   // basic or cp: then l2, then bt
    if (ls_default == 2 || ls_default == 3) { max_linesearch_iter = 2; ls_types[1] = ls_default == 2 ? 3 : 2; }
    struct Restore { int &t; int v; ~Restore() { t = v; } } restore{opt.linesearch_type, ls_default};   // :539-550
    time = 0.0;
    *converged = 0;
    last_substeps.clear();
    for (;;) {
      dtime = dt_iter;
      op_pre_solve();
      opt.linesearch_type = ls_types[linesearch_iter];
      int nit = 0, lit = 0;
      int reason = snes_solve(&nit, &lit);
      if (comm.active()) {  // every rank must take the same branch: worst reason wins
        double r = -(double)reason;
        HIP_CHECK(hipMemcpyAsync(sc.p + S_D2, &r, sizeof(double), hipMemcpyHostToDevice, stream));
        pub_valid = false;
        comm.allreduce_max(sc.p + S_D2, 1, stream);
        fetch_scalars();   // (a bounded wait when the communicator is RCCL)
        r = h_sc[S_D2];
        if (reason > 0 && r > 0) reason = -(int)r;
      }
      *converged_reason = reason;
      if (reason < 0) {
        ++linesearch_iter;
        if (opt.use_dynamic_linesearch && linesearch_iter < max_linesearch_iter) ++cum_ls_switches;   // another line search, same dt
        else {
          linesearch_iter = 0;
          ++num_time_cuts; ++cum_cuts;
          dt_iter = 0.5 * dt_iter;
        }
        vec_copy(soln_prev.p, soln.p);
      } else {
        *converged = 1;
        time += dt_iter;
        last_substeps.push_back(dt_iter);
        cum_newton += nit; cum_linear += lit;
        vec_copy(soln.p, soln_prev.p);  // VSFMSOEPostSolve (SystemOfEquationsVSFMType.F90:600)
        if (opt.track_mass_exchange && !th) accumulate_mass_exchange(dt_iter);
      }
      if (num_time_cuts > max_num_time_cuts) { *converged = 0; break; }
      if (time >= target_time) break;
    }
    HIP_CHECK(hipStreamSynchronize(stream));
  }

  // =====================================================================================
  // host <-> device in the caller's ordering
  // =====================================================================================
  void put_cells(const double *host_ext, double *dev_int) {
    if (identity_perm) {
      HIP_CHECK(hipMemcpyAsync(dev_int, host_ext, (size_t)n * sizeof(double), hipMemcpyHostToDevice, stream));
    } else {
      HIP_CHECK(hipMemcpyAsync(tmp_ext.p, host_ext, (size_t)n * sizeof(double), hipMemcpyHostToDevice, stream));
      launch_gather(n, d_int2ext.p, tmp_ext.p, dev_int, stream);
    }
    HIP_CHECK(hipStreamSynchronize(stream));  // the caller's buffer is free again on return
  }
  // whole unknown vectors: one field of n cells, or [P; T] for the coupled system
  void put_state(const double *host_ext, double *dev_int) {
    for (int f = 0; f * n < nrow; ++f) put_cells(host_ext + (size_t)f * n, dev_int + (size_t)f * n);
    ghost_zero(dev_int);
  }
  void get_state(const double *dev_int, double *host_ext) {
    for (int f = 0; f * n < nrow; ++f) get_cells(dev_int + (size_t)f * n, host_ext + (size_t)f * n, n);
  }
  void get_cells(const double *dev_int, double *host_ext, int count) {
    const double *src = dev_int;
    if (!identity_perm) { launch_gather(n, d_ext2int.p, dev_int, tmp_ext.p, stream); src = tmp_ext.p; }
    HIP_CHECK(hipMemcpyAsync(host_ext, src, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
  }

  // ThermalSOESetRDataFromCLM (src/mpp/soe/SystemOfEquationsThermalType.F90:234-289) + ThermKSPTempSoilAuxVarSetRValues
  void thermal_set_data(int auxvar_type, int var_type, int cond_id, int cnt, const double *data) {
    need_setup(true);
    if (!thermal) throw Error("problem_type is not thermal");
    if (auxvar_type == MPPGPU_AUXVAR_INTERNAL) {
      if (cnt > n) throw Error("set_data: size(data_1d) > nauxvar");
      std::vector<double> *dst = nullptr;
      switch (var_type) {
        case MPPGPU_VAR_LIQ_AREAL_DEN: dst = &th_liq; break;
        case MPPGPU_VAR_ICE_AREAL_DEN: dst = &th_ice; break;
        case MPPGPU_VAR_SNOW_WATER: dst = &th_snow; break;
        case MPPGPU_VAR_TUNING_FACTOR: dst = &th_tuning; break;
        case MPPGPU_VAR_NUM_SNOW_LYR:
          if (th_nsnow.empty()) th_nsnow.assign(n, 0);
          for (int i = 0; i < cnt; ++i) th_nsnow[i] = (int)data[i];
          th_cell_dirty = true;
          return;
        default: throw Error("set_data: unknown var_type " + std::to_string(var_type));
      }
      if (dst->empty()) dst->assign(n, dst == &th_tuning ? 1.0 : 0.0);
      for (int i = 0; i < cnt; ++i) (*dst)[i] = data[i];
      th_cell_dirty = true;
    } else if (auxvar_type == MPPGPU_AUXVAR_BC) {
      if (cond_id < 1 || cond_id > (int)bcs.size()) throw Error("set_data: bad boundary condition id");
      Condition &c = bcs[cond_id - 1];
      if (cnt > c.n) throw Error("set_data: size(data_1d) > nauxvar");
      for (int i = 0; i < cnt; ++i) {
        const int e = bc_pos_of_g[c.offset + i];
        switch (var_type) {
          case MPPGPU_VAR_BC_SS_CONDITION: c.value[i] = data[i]; h_bc_value[e] = data[i]; break;
          case MPPGPU_VAR_FRAC: h_bc_frac[e] = data[i]; break;
          case MPPGPU_VAR_DHS_DT: h_bc_dhsdT[e] = data[i]; break;
          case MPPGPU_VAR_ACTIVE: h_bc_act[e] = (signed char)(data[i] != 0.0); break;
          default: throw Error("set_data: unknown var_type " + std::to_string(var_type));
        }
      }
      d_bc_value.upload(h_bc_value, stream); d_bc_frac.upload(h_bc_frac, stream); d_bc_dhsdT.upload(h_bc_dhsdT, stream);
      d_bc_act.upload(h_bc_act.data(), h_bc_act.size(), stream);
      HIP_CHECK(hipStreamSynchronize(stream));
    } else if (auxvar_type == MPPGPU_AUXVAR_SS) {
      if (var_type != MPPGPU_VAR_BC_SS_CONDITION) throw Error("set_data: unknown var_type " + std::to_string(var_type));
      if (cond_id < 1 || cond_id > (int)sss.size()) throw Error("set_data: bad source-sink id");
      Condition &c = sss[cond_id - 1];
      if (cnt > c.n) throw Error("set_data: size(data_1d) > nauxvar");
      for (int i = 0; i < cnt; ++i) c.value[i] = data[i];
      std::vector<double> src(n, 0.0);   // dense heat-rate source [W]: sum over conditions (GE :950-962)
      for (auto &sc_ : sss)
        for (int i = 0; i < sc_.n; ++i)
          if (active[sc_.cell[i]]) src[ext2int[sc_.cell[i]]] += sc_.value[i];
      d_src.upload(src, stream);
      HIP_CHECK(hipStreamSynchronize(stream));
    } else throw Error("set_data: unknown auxvar_type " + std::to_string(auxvar_type));
  }

  void set_data(int auxvar_type, int var_type, int cond_id, int cnt, const double *data) {
    need_setup(true);
    if (thermal) { thermal_set_data(auxvar_type, var_type, cond_id, cnt, data); return; }
    if (th && auxvar_type == MPPGPU_AUXVAR_BC && (var_type == MPPGPU_VAR_PRESSURE || var_type == MPPGPU_VAR_TEMPERATURE)) {
      // the other field of a boundary auxvar: aux_vars_bc(:)%temperature of the mass equation, %pressure of the heat
      // equation -- the reference drivers assign them directly (mass_and_heat_model_problem.F90:583-640,
      // th_mms_problem.F90:846-875)
      if (cond_id < 1 || cond_id > (int)bcs.size()) throw Error("set_data: bad boundary condition id");
      Condition &c = bcs[cond_id - 1];
      if ((c.eqn == 0) != (var_type == MPPGPU_VAR_TEMPERATURE))
        throw Error("set_data: a mass-equation boundary takes VAR_TEMPERATURE, a heat-equation boundary VAR_PRESSURE");
      if (cnt > c.n) throw Error("set_data: size(data_1d) > nauxvar");
      for (int i = 0; i < cnt; ++i) { c.other[i] = data[i]; h_bc_other[bc_pos_of_g[c.offset + i]] = data[i]; }
      d_bc_other.upload(h_bc_other, stream);
      HIP_CHECK(hipStreamSynchronize(stream));
      return;
    }
    if (auxvar_type == MPPGPU_AUXVAR_INTERNAL && var_type == MPPGPU_VAR_FRAC_LIQ_SAT && !th) {
      // VSFMSOESetDataFromCLM + RichardsODEPressureGetFromSOEAuxVarsIntrn (GE:536-582): active cells only
      if (cnt > n) throw Error("set_data: size(data_1d) > nauxvar");
      if (h_frac.empty()) h_frac.assign(n, 1.0);
      for (int i = 0; i < cnt; ++i) if (active[i]) h_frac[i] = data[i];
      if (!d_frac.p) d_frac.alloc(n);
      put_cells(h_frac.data(), d_frac.p);
      return;
    }
    if (var_type != MPPGPU_VAR_BC_SS_CONDITION) throw Error("set_data: unknown var_type " + std::to_string(var_type));
    if (auxvar_type == MPPGPU_AUXVAR_BC) {
      if (cond_id < 1 || cond_id > (int)bcs.size()) throw Error("set_data: bad boundary condition id");
      Condition &c = bcs[cond_id - 1];
      if (cnt > c.n) throw Error("set_data: size(data_1d) > nauxvar");
      std::memcpy(c.value.data(), data, (size_t)cnt * sizeof(double));
      // The values go to the device through a pinned staging area (one region per condition) and are scattered there into the
      // cell-grouped order the boundary kernel reads: no host loop over the entries, no wait for the stream (a time step's
      // boundary values are typically set right before StepDT, which follows on the same stream).
      if (!h_bc_stage) {
        HIP_CHECK(hipHostMalloc(&h_bc_stage, std::max<size_t>(nb, 1) * sizeof(double), hipHostMallocDefault));
        d_bc_stage.alloc(std::max(nb, 1));
        d_bc_pos.upload(bc_pos_of_g, stream);
        HIP_CHECK(hipEventCreateWithFlags(&bc_stage_ev, hipEventDisableTiming));
        bc_stage_pending.assign(bcs.size(), 0);
      }
      if (bc_stage_pending[cond_id - 1]) {   // this region's previous copy may still be reading it
        HIP_CHECK(hipEventSynchronize(bc_stage_ev));
        std::fill(bc_stage_pending.begin(), bc_stage_pending.end(), 0);
      }
      std::memcpy(h_bc_stage + c.offset, data, (size_t)cnt * sizeof(double));
      HIP_CHECK(hipMemcpyAsync(d_bc_stage.p + c.offset, h_bc_stage + c.offset, (size_t)cnt * sizeof(double), hipMemcpyHostToDevice, stream));
      HIP_CHECK(hipEventRecord(bc_stage_ev, stream));
      bc_stage_pending[cond_id - 1] = 1;
      launch_scatter(cnt, d_bc_pos.p + c.offset, d_bc_stage.p + c.offset, d_bc_value.p, stream);
    } else if (auxvar_type == MPPGPU_AUXVAR_SS) {
      if (cond_id < 1 || cond_id > (int)sss.size()) throw Error("set_data: bad source-sink id");
      Condition &c = sss[cond_id - 1];
      if (cnt > c.n) throw Error("set_data: size(data_1d) > nauxvar");
      for (int i = 0; i < cnt; ++i) c.value[i] = data[i];
      rebuild_src();
    } else throw Error("set_data: unknown auxvar_type " + std::to_string(auxvar_type));
  }

  // dense per-cell source: sum over all mass-rate conditions of value/FMWH2O [kmol/s] (GE:1896); coupled TH: heat rates [W]
  void rebuild_src() {
    std::vector<double> src(nrow, 0.0);
    for (auto &s : sss)
      for (int i = 0; i < s.n; ++i) {
        if (!active[s.cell[i]]) continue;
        if (s.cond_type == MPPGPU_COND_DOWNREG_MASS_RATE_CAMPBELL || s.cond_type == MPPGPU_COND_DOWNREG_MASS_RATE_FETCH2) continue;
        if (s.eqn == 0) src[ext2int[s.cell[i]]] += s.value[i] / kFMWH2O;
        else src[n + ext2int[s.cell[i]]] += s.value[i];   // COND_HEAT_RATE [W] (GoveqnThermalEnthalpySoilType.F90:1481)
      }
    d_src.upload(src, stream);
    upload_sink_params();
    HIP_CHECK(hipStreamSynchronize(stream));
  }

  // VSFMComputeLateralFlux (SystemOfEquationsVSFMType.F90:848-889) + RichardsODEComputeLateralFlux (GE:2615-2743):
  // SetDtime, PreSolve, then the Darcy flux of every lateral connection at soln_prev, accumulated [kg/s] into the
  // mass-rate condition `cond_id` (the reference's 'Lateral_flux' source-sink: one entry per cell, in cell order)
  void compute_lateral_flux(double dt, int cond_id) {
    if (th || thermal) throw Error("lateral-flux operator split is a VSFM operation");
    if (cond_id < 1 || cond_id > (int)sss.size()) throw Error("RichardsODEComputeLateralFlux: Lateral source-sink not found.");
    Condition &lc = sss[cond_id - 1];
    if (lc.cond_type != MPPGPU_COND_MASS_RATE || lc.n != n) throw Error("the lateral-flux condition must be a COND_MASS_RATE over every cell");
    for (int i = 0; i < n; ++i) if (lc.cell[i] != i) throw Error("the lateral-flux condition must list the cells in order");
    if (!(dt > 0.0)) throw Error("dt must be positive");
    dtime = dt;
    ensure_ready();
    op_pre_solve();
    std::fill(lc.value.begin(), lc.value.end(), 0.0);
    if (nlat > 0) {
      launch_conn_flux(cells(soln_prev.p), ConnArrays{d_lat_Dq.p, d_lat_upw.p, d_lat_dgrav.p, d_lat_area.p}, d_lat_up.p, d_lat_dn.p,
                       nlat, d_lat_flux.p, stream);
      std::vector<double> fl(nlat);
      HIP_CHECK(hipMemcpyAsync(fl.data(), d_lat_flux.p, (size_t)nlat * sizeof(double), hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      for (int c = 0; c < nlat; ++c) {   // flux*FMWH2O [kg/s] (GE:2733-2737)
        lc.value[lat_up_e[c]] = lc.value[lat_up_e[c]] + fl[c];
        lc.value[lat_dn_e[c]] = lc.value[lat_dn_e[c]] - fl[c];
      }
    }
    rebuild_src();
  }
  void get_condition_values(int kind, int cond_id, int cnt, double *out) const {
    const std::vector<Condition> &list = kind == MPPGPU_COND_BC ? bcs : sss;
    if (cond_id < 1 || cond_id > (int)list.size()) throw Error("bad condition id");
    if (cnt > list[cond_id - 1].n) throw Error("size(data_1d) > nauxvar");
    for (int i = 0; i < cnt; ++i) out[i] = list[cond_id - 1].value[i];
  }

  void get_data(int auxvar_type, int var_type, int cond_id, int cnt, double *out) {
    ensure_ready();
    if (auxvar_type == MPPGPU_AUXVAR_INTERNAL) {
      if (cnt > n) throw Error("get_data: size(data_1d) > nauxvar");
      if (var_type == MPPGPU_VAR_PRESSURE) { get_cells(soln.p, out, cnt); return; }
      if (th && var_type == MPPGPU_VAR_TEMPERATURE) { get_cells(soln.p + n, out, cnt); return; }
      if (var_type == MPPGPU_VAR_FRAC_LIQ_SAT) { for (int i = 0; i < cnt; ++i) out[i] = h_frac.empty() ? 1.0 : h_frac[i]; return; }
      if (var_type == MPPGPU_VAR_LATERAL_MASS_EXCHANGED) {
        if (!opt.track_mass_exchange) throw Error("get_data: set option track_mass_exchange first");
        for (int i = 0; i < cnt; ++i) out[i] = lat_mass_exc.empty() ? 0.0 : lat_mass_exc[i];
        return;
      }
      if (th) throw Error("get_data: the coupled system exports VAR_PRESSURE and VAR_TEMPERATURE");
      if (var_type != MPPGPU_VAR_LIQ_SAT && var_type != MPPGPU_VAR_MASS && var_type != MPPGPU_VAR_SOIL_MATRIX_POT)
        throw Error("get_data: unknown var_type " + std::to_string(var_type));
      launch_cell_var(cells(soln.p), n, var_type, tmp_int.p, stream);
      get_cells(tmp_int.p, out, cnt);
      // inactive cells: RichardsODEPressureSetDataInSOEAuxVar copies nothing for them (GE:1170-1195), so the SoE auxvar keeps the
      // zeros of VSFMSOEAuxVarInit (SystemOfEquationsVSFMAuxType.F90:64-74)
      if (any_inactive) for (int i = 0; i < cnt; ++i) if (active[i] == 0) out[i] = 0.0;
    } else if (th) {
      throw Error("get_data: the coupled system exports AUXVAR_INTERNAL only");
    } else if (auxvar_type == MPPGPU_AUXVAR_BC) {
      if (cond_id < 1 || cond_id > (int)bcs.size()) throw Error("get_data: bad boundary condition id");
      if (var_type == MPPGPU_VAR_BC_MASS_EXCHANGED) {
        if (!opt.track_mass_exchange) throw Error("get_data: set option track_mass_exchange first");
        Condition &cm = bcs[cond_id - 1];
        if (cnt > cm.n) throw Error("get_data: size(data_1d) > nauxvar");
        for (int i = 0; i < cnt; ++i) out[i] = bnd_mass_exc.empty() ? 0.0 : bnd_mass_exc[cm.offset + i];
        return;
      }
      if (var_type != MPPGPU_VAR_MASS_FLUX) throw Error("get_data: unknown var_type");
      Condition &c = bcs[cond_id - 1];
      if (cnt > c.n) throw Error("get_data: size(data_1d) > nauxvar");
      launch_bc(4, bc_args(soln.p, F.p), stream);
      std::vector<double> fl(nb);
      HIP_CHECK(hipMemcpyAsync(fl.data(), d_bc_flux.p, nb * sizeof(double), hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      for (int i = 0; i < cnt; ++i) out[i] = fl[bc_pos_of_g[c.offset + i]];
    } else if (auxvar_type == MPPGPU_AUXVAR_SS) {
      if (cond_id < 1 || cond_id > (int)sss.size()) throw Error("get_data: bad source-sink id");
      if (var_type != MPPGPU_VAR_MASS_FLUX) throw Error("get_data: unknown var_type");
      Condition &c = sss[cond_id - 1];
      if (cnt > c.n) throw Error("get_data: size(data_1d) > nauxvar");
      std::vector<double> fl(ndr);
      if (ndr > 0) {   // down-regulated: ss_flux = value / factor (Campbell) or value * factor (FETCH2) (GE:1911, 1927)
        launch_sinks(4, sink_args(soln.p, F.p), stream);
        HIP_CHECK(hipMemcpyAsync(fl.data(), d_dr_flux.p, ndr * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
      }
      for (int i = 0; i < cnt; ++i) {
        const int e = dr_pos_of_g[c.offset + i];
        out[i] = e >= 0 ? fl[e] : c.value[i];  // ss_flux = cur_cond%value (GE:1898)
      }
    } else if (auxvar_type == MPPGPU_AUXVAR_CONN_INTERNAL) {
      if (var_type != MPPGPU_VAR_MASS_FLUX) throw Error("get_data: unknown var_type");
      if (cnt > nc) throw Error("get_data: size(data_1d) > nauxvar");
      DevBuf<double> fl;
      fl.alloc(nc_int);
      launch_conn_flux(cells(soln.p), conns(), d_cup.p, d_cdn.p, nc_int, fl.p, stream);
      std::vector<double> h(nc_int);
      HIP_CHECK(hipMemcpyAsync(h.data(), fl.p, (size_t)nc_int * sizeof(double), hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      for (int i = 0; i < cnt; ++i) out[i] = h[cext2int[i]];
    } else throw Error("get_data: unknown auxvar_type " + std::to_string(auxvar_type));
  }

  // PETSc's default PC is ILU(0).  On a grid whose connections are all vertical the matrix is block diagonal with
  // one tridiagonal block per soil column; ILU(0) in that ordering has no fill to drop, i.e. it IS the exact
  // per-column LU that the block-Jacobi kernels compute.  General (7-point) ILU(0) needs a level-scheduled
  // triangular solve that is not built yet.
  void check_ilu() {
    if (!opt.pc_is_ilu) return;
    opt.pc_type = (vertical_only && !th) ? 2 : 1;   // general meshes: level-scheduled ILU(0) in the caller's natural ordering
    pc_ready = false;
  }

  // Level sets of the ILU(0) dependency graphs.  Elimination order = the caller's numbering (PETSc factors in natural
  // ordering), i.e. ord[i] = external id of internal row i; level(i) = 1 + max level over the row's L (resp. U) part.
  // caller's (natural) position of an internal row and back; coupled TH: field-blocked [P(1..N); T(1..N)] both ways
  int row_int2ext(int r) const { return r < n ? int2ext[r] : n + int2ext[r - n]; }
  int row_ext2int(int pos) const { return pos < n ? ext2int[pos] : n + ext2int[pos - n]; }
  // Tiles (<= 256 consecutive positions of one level) and, per tile, the tiles that hold the dependencies of its rows.
  // skip_level0: rows of level 0 are final before the solve starts (unit lower triangle), so they get no tile and
  // nobody waits for them.  More than kChainDeps distinct dependency tiles anywhere: no chain (per-level launches).
  void build_chain(Chain &ch, const std::vector<int> &lptr, const std::vector<int> &col, int K, bool skip_level0) {
    ch.ok = false; ch.ntiles = 0;
    const int N = nrow, nlev = (int)lptr.size() - 1;
    if (nlev <= 0 || K <= 0) return;
    const int first = skip_level0 ? 1 : 0;
    std::vector<int> base(nlev + 1, 0);           // first tile of every level
    for (int l = 0; l < nlev; ++l) base[l + 1] = base[l] + (l < first ? 0 : (lptr[l + 1] - lptr[l] + 255) / 256);
    const int nt = base[nlev];
    if (nt <= 0) return;
    std::vector<int> tlo(nt), thi(nt), dep((size_t)kChainDeps * nt, -1);
    auto tile_of = [&](int pos, int hint_level) {   // dependencies sit in the previous level almost always
      int l = hint_level;
      if (!(pos >= lptr[l] && pos < lptr[l + 1])) l = (int)(std::upper_bound(lptr.begin(), lptr.end(), pos) - lptr.begin()) - 1;
      return l < first ? -1 : base[l] + (pos - lptr[l]) / 256;
    };
    for (int l = first; l < nlev; ++l)
      for (int lo = lptr[l], t = base[l]; lo < lptr[l + 1]; lo += 256, ++t) {
        const int hi = std::min(lo + 256, lptr[l + 1]);
        tlo[t] = lo; thi[t] = hi;
        int nd = 0;
        int *d = &dep[0];
        for (int k = 0; k < K; ++k)
          for (int p = lo; p < hi; ++p) {
            const int c = col[(size_t)k * N + p];
            if (c == p) continue;
            const int tt = tile_of(c, l > 0 ? l - 1 : 0);
            if (tt < 0) continue;
            if (tt >= t) return;                    // not a triangular dependency: never for a level schedule
            bool seen = false;
            for (int q = 0; q < nd; ++q) seen |= (d[(size_t)q * nt + t] == tt);
            if (seen) continue;
            if (nd == kChainDeps) return;
            d[(size_t)nd++ * nt + t] = tt;
          }
      }
    ch.tile_lo.upload(tlo, stream); ch.tile_hi.upload(thi, stream); ch.dep.upload(dep, stream);
    ch.flags.alloc(nt); ch.flags.zero(stream);
    ch.ticket.alloc(1); ch.ticket.zero(stream);
    ch.ntiles = nt;
    // one returning atomic per ticket: a single counter serves ~88 of them per microsecond (MI355X_MICROARCH.md, dequeue), so
    // long chains hand out several consecutive tiles per ticket
    ch.tiles_per_block = nt >= (1 << 18) ? 4 : (nt >= (1 << 16) ? 2 : 1);
    ch.nblocks = (nt + ch.tiles_per_block - 1) / ch.tiles_per_block;
    ch.launches = 0;
    ch.ok = true;
  }
  // one 8-byte read-back behind every chained application (the queue drains once per application: < 1 % of a 1-14 ms solve)
  bool chain_timed_out() {
    double e = 0.0;
    HIP_CHECK(hipMemcpyAsync(&e, sc.p + S_ERR, sizeof(double), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if (e == 0.0) return false;
    opt.ilu_chain = 0; opt.ilu_box_queue = false; opt.ilu_box_stream = false;   // (a streamed solve that gave up leaves its face buffers unusable)
    if (chain_fallbacks++ == 0)
      std::fprintf(stderr, "mppgpu: a flag-chained ILU(0) solve timed out waiting for a tile or box (spin budget %u); redone with one launch per "
                           "level, chained solves are off for this handle from now on\n", opt.ilu_chain_spins);
    launch_set(1, 0.0, sc.p + S_ERR, stream);
    return true;
  }
  // The chain pays per dependency with an uncached (agent-scope) load and the waiting tiles poll (every ~0.4 us: polling
  // faster floods the fabric once thousands of tiles wait); a per-level launch pays ~1.5-4.7 us per level.  Measured on
  // MI355X, per application: thermal 256^3 (22 K rows per level, 3 entries per row) 4.5 -> 3.6 ms, 384^3 8.2 -> 7.3 ms,
  // 512^3 (87 K) 14.5 -> 13.5 ms, 3-D VSFM 128x128x64 (3 K rows per level) 1.51 -> 1.10 ms; the TH columns (65 K rows per
  // level, 6 entries per row, 160 levels) 0.89 -> 0.98 ms: chain when rows-per-level x entries-per-row stays under 3x10^5.
  bool ilu_chained() const {
    if (!opt.ilu_chain || !chainL.ok || !chainU.ok) return false;
    if (opt.ilu_chain == 2) return true;
    const double nlev = (double)std::max<size_t>(h_lptr.size(), 2) - 1.0;
    return (double)nrow / nlev * std::max(ell_KL, ell_KU) < 3.0e5;
  }
  void ilu_solve_chained() {
    const unsigned ep = ++chain_epoch;
    ChainArgs a{nrow, ell_KL, d_Lcol.p, d_Lval.p, nullptr, chainL.tile_lo.p, chainL.tile_hi.p, chainL.dep.p, chainL.ntiles,
                kChainDeps, chainL.flags.p, ep, ilu_y.p, sc.p, opt.ilu_chain_spins, opt.ilu_chain_tickets ? chainL.ticket.p : nullptr,
                chainL.launches * (unsigned)chainL.nblocks, opt.ilu_chain_tickets ? chainL.tiles_per_block : 1};
    if (opt.ilu_chain_tickets) ++chainL.launches;
    launch_ell_chain(a, stream);
    launch_gather(nrow, d_l2u.p, ilu_y.p, ilu_x.p, stream);
    ChainArgs b{nrow, ell_KU, d_Ucol.p, d_Uval.p, d_Uidiag.p, chainU.tile_lo.p, chainU.tile_hi.p, chainU.dep.p, chainU.ntiles,
                kChainDeps, chainU.flags.p, ep, ilu_x.p, sc.p, opt.ilu_chain_spins, opt.ilu_chain_tickets ? chainU.ticket.p : nullptr,
                chainU.launches * (unsigned)chainU.nblocks, opt.ilu_chain_tickets ? chainU.tiles_per_block : 1};
    if (opt.ilu_chain_tickets) ++chainU.launches;
    launch_ell_chain(b, stream);
  }

  void ilu_analyse() {
    if (d_ord.p) return;
    PhaseTimer pt("ILU(0): level analysis");
    const int N = nrow;
    std::vector<int> lvl(N, 0), ulvl(N, 0);
    int nl = 0, nul = 0;
    for (int pos = 0; pos < N; ++pos) {
      const int i = row_ext2int(pos);
      int m = 0;
      for (int q = h_rowptr[i]; q < h_rowptr[i + 1]; ++q) {
        const int j = h_col[q];
        if (row_int2ext(j) < pos) m = std::max(m, lvl[j] + 1);
      }
      lvl[i] = m; nl = std::max(nl, m + 1);
    }
    for (int pos = N - 1; pos >= 0; --pos) {
      const int i = row_ext2int(pos);
      int m = 0;
      for (int q = h_rowptr[i]; q < h_rowptr[i + 1]; ++q) {
        const int j = h_col[q];
        if (row_int2ext(j) > pos) m = std::max(m, ulvl[j] + 1);
      }
      ulvl[i] = m; nul = std::max(nul, m + 1);
    }
    auto bucket = [&](const std::vector<int> &lv, int nlev_, std::vector<int> &ptr, std::vector<int> &rows) {
      ptr.assign(nlev_ + 1, 0);
      for (int i = 0; i < N; ++i) ptr[lv[i] + 1]++;
      for (int l = 0; l < nlev_; ++l) ptr[l + 1] += ptr[l];
      rows.resize(N);
      std::vector<int> fill(ptr.begin(), ptr.end() - 1);
      for (int pos = 0; pos < N; ++pos) { const int i = row_ext2int(pos); rows[fill[lv[i]]++] = i; }
    };
    std::vector<int> lrows, urows, ord(N);
    bucket(lvl, nl, h_lptr, lrows);
    bucket(ulvl, nul, h_uptr, urows);
    for (int r = 0; r < N; ++r) ord[r] = row_int2ext(r);
    d_lrows.upload(lrows, stream); d_urows.upload(urows, stream);
    d_ord.upload(ord, stream);
    d_lu.alloc(nnz); d_idiag.alloc(N);
    ilu_factor_program(lrows, ord);
    HIP_CHECK(hipStreamSynchronize(stream));
  }
  // level-ordered ELL maps of the general triangular solves: position of every row in the L ordering (lrows) and in the U
  // ordering (urows); built when the box solves do not apply
  void ilu_build_ell() {
    if (ell_built) return;
    ell_built = true;
    const int N = nrow;
    std::vector<int> lrows(N), urows(N), ord(N);
    HIP_CHECK(hipMemcpyAsync(lrows.data(), d_lrows.p, (size_t)N * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(urows.data(), d_urows.p, (size_t)N * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(ord.data(), d_ord.p, (size_t)N * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    std::vector<int> posL(N), posU(N);
    for (int p = 0; p < N; ++p) { posL[lrows[p]] = p; posU[urows[p]] = p; }
    ell_KL = ell_KU = 0;
    for (int i = 0; i < N; ++i) {
      int kl = 0, ku = 0;
      for (int q = h_rowptr[i]; q < h_rowptr[i + 1]; ++q) {
        const int oj = ord[h_col[q]];
        if (oj < ord[i]) ++kl; else if (oj > ord[i]) ++ku;
      }
      ell_KL = std::max(ell_KL, kl); ell_KU = std::max(ell_KU, ku);
    }
    std::vector<int> Lslot((size_t)ell_KL * N, -1), Lcol((size_t)ell_KL * N), Uslot((size_t)ell_KU * N, -1), Ucol((size_t)ell_KU * N);
    std::vector<int> l2u(N);
    std::vector<std::pair<int, int>> ents;   // (caller's column, CSR slot)
    for (int i = 0; i < N; ++i) {
      const int pl = posL[i], pu = posU[i];
      for (int k = 0; k < ell_KL; ++k) Lcol[(size_t)k * N + pl] = pl;
      for (int k = 0; k < ell_KU; ++k) Ucol[(size_t)k * N + pu] = pu;
      int kl = 0, ku = 0;
      ents.clear();
      for (int q = h_rowptr[i]; q < h_rowptr[i + 1]; ++q) ents.emplace_back(ord[h_col[q]], q);
      std::sort(ents.begin(), ents.end());   // a row's entries in the caller's column order, the order PETSc eliminates them in
      for (auto &en : ents) {
        const int q = en.second, j = h_col[q], oj = en.first;
        if (oj < ord[i]) { Lslot[(size_t)kl * N + pl] = q; Lcol[(size_t)kl * N + pl] = posL[j]; ++kl; }
        else if (oj > ord[i]) { Uslot[(size_t)ku * N + pu] = q; Ucol[(size_t)ku * N + pu] = posU[j]; ++ku; }
      }
      l2u[pu] = pl;
    }
    build_chain(chainL, h_lptr, Lcol, ell_KL, /*skip_level0=*/true);
    build_chain(chainU, h_uptr, Ucol, ell_KU, /*skip_level0=*/false);
    d_Lslot.upload(Lslot, stream); d_Lcol.upload(Lcol, stream); d_Uslot.upload(Uslot, stream); d_Ucol.upload(Ucol, stream);
    d_l2u.upload(l2u, stream); d_posU.upload(posU, stream);
    d_Lval.alloc((size_t)ell_KL * N); d_Uval.alloc((size_t)ell_KU * N); d_Uidiag.alloc(N);
    ilu_y.alloc(N); ilu_x.alloc(N);
    HIP_CHECK(hipStreamSynchronize(stream));
  }
  // elimination program of the numeric factorisation, rows in L level order (see k_ilu_factor_program)
  void ilu_factor_program(const std::vector<int> &lrows, const std::vector<int> &ord) {
    const int N = nrow;
    {
      std::vector<long long> fptr((size_t)N + 1, 0);
      std::vector<int> prog;
      prog.reserve((size_t)N * 16);
      std::vector<std::pair<int, int>> lpart;   // (ord of k, slot of (i,k))
      for (int p = 0; p < N; ++p) {
        const int i = lrows[p];
        fptr[p] = (long long)prog.size();
        lpart.clear();
        for (int q = h_rowptr[i]; q < h_rowptr[i + 1]; ++q)
          if (ord[h_col[q]] < ord[i]) lpart.emplace_back(ord[h_col[q]], q);
        std::sort(lpart.begin(), lpart.end());   // pivots in increasing elimination order
        for (auto &lk : lpart) {
          const int k = h_col[lk.second];
          prog.push_back(lk.second); prog.push_back(k);
          const size_t mpos = prog.size();
          prog.push_back(0);
          int m = 0;
          for (int q = h_rowptr[k]; q < h_rowptr[k + 1]; ++q) {   // U part of row k, in its CSR order
            const int j = h_col[q];
            if (ord[j] <= lk.first) continue;
            const int s_ij = find_slot(i, j);
            if (s_ij >= 0) { prog.push_back(q); prog.push_back(s_ij); ++m; }
          }
          prog[mpos] = m;
        }
      }
      fptr[N] = (long long)prog.size();
      d_fptr.upload(fptr, stream); d_fprog.upload(prog, stream);
      HIP_CHECK(hipStreamSynchronize(stream));
    }
  }
  // Is the pattern a structured 7-point (5-, 3-point) grid in the caller's numbering?  Then the triangular solves run box by box
  // (BoxGrid): extents from the distinct positive column distances {1, n1, n1*n2}, every row checked against the grid.
  void ilu_box_analyse() {
    if (box_tried || opt.ilu_box == 0) return;   // (ilu_box = 0 leaves the test for a later set_option("ilu_box", ...))
    box_tried = true; box_ok = false;
    if (th || nrow != n || nred < 2) return;
    PhaseTimer pt("ILU(0): structured-grid test");
    // With ghost cells (last in the numbering, identity layout, identity rows, zero right-hand side in the preconditioner) the
    // local ILU(0) is the ILU(0) of the block of the owned cells: the test and the boxes cover that block, ghost columns are skipped.
    const int N = nred;
    int offs[3], no = 0;
    for (int r = 0; r < N; ++r) {
      const int e = int2ext[r];
      for (int q = h_rowptr[r]; q < h_rowptr[r + 1]; ++q) {
        if (h_col[q] >= N) continue;
        const int d = int2ext[h_col[q]] - e;
        if (d <= 0) continue;
        bool seen = false;
        for (int k = 0; k < no; ++k) seen = seen || offs[k] == d;
        if (seen) continue;
        if (no == 3) return;
        offs[no++] = d;
      }
    }
    std::sort(offs, offs + no);
    if (no == 0 || offs[0] != 1) return;
    const long long n1 = no >= 2 ? offs[1] : N, s3 = no == 3 ? offs[2] : N;
    if (N % n1 || s3 % n1 || N % s3) return;
    const int n2 = (int)(s3 / n1), n3 = (int)(N / s3);
    std::atomic<bool> grid{true};
    host_parallel_for(N, [&](long long r) {   // the diagonal and exactly the grid neighbours in every row
      if (!grid) return;
      const int e = int2ext[r], ci = (int)(e % n1), cj = (int)((e / n1) % n2), ck = (int)(e / s3);
      const int expect = (ci > 0) + (ci < n1 - 1) + (cj > 0) + (cj < n2 - 1) + (ck > 0) + (ck < n3 - 1);
      int cnt = 0;
      bool diag = false;
      for (int q = h_rowptr[r]; q < h_rowptr[r + 1]; ++q) {
        if (h_col[q] >= N) continue;
        const long long d = (long long)int2ext[h_col[q]] - e;
        if (d == 0) diag = true;
        else if ((d == -1 && ci > 0) || (d == 1 && ci < n1 - 1) || (n2 > 1 && ((d == -n1 && cj > 0) || (d == n1 && cj < n2 - 1))) ||
                 (n3 > 1 && ((d == -s3 && ck > 0) || (d == s3 && ck < n3 - 1)))) ++cnt;
        else { grid = false; return; }
      }
      if (!diag || cnt != expect) grid = false;
    });
    if (!grid) return;
    BoxGrid g{};
    g.n1 = (int)n1; g.n2 = n2; g.n3 = n3;
    g.B3 = 1;
    while (g.B3 < 8 && g.B3 < n3) g.B3 *= 2;
    g.B2 = 64 / g.B3;
    g.lg2 = 0;
    while ((1 << g.lg2) < g.B2) ++g.lg2;
    g.B1 = opt.ilu_box_len > 0 ? opt.ilu_box_len : (N >= (1 << 24) ? 64 : 32);   // longer boxes: less padding of the skewed layout
                                                                                  // (256^3: 1.18 / 1.09 / 1.06 ms at 32 / 64 / 128; 512^3: 5.09 / 4.98 at 64 / 128 -- profiles/r04/thermal_box_length_streams.json)
    g.B1 = std::max(1, std::min(std::min(g.B1, g.n1), kBoxMaxLen));
    g.NB1 = (g.n1 + g.B1 - 1) / g.B1; g.NB2 = (g.n2 + g.B2 - 1) / g.B2; g.NB3 = (g.n3 + g.B3 - 1) / g.B3;
    g.S = (g.B1 + g.B2 + g.B3 - 2 + 7) / 8 * 8;   // step rows per box: a multiple of 8 (rows go in pairs; the streamed solves hand faces over 8 steps at a time)
    const long long nboxes = (long long)g.NB1 * g.NB2 * g.NB3, slots = nboxes * g.S * 64;
    if (slots >= (1ll << 31)) return;
    Scratch<int> cellmap((size_t)slots, -1);   // (filled and first touched by all host threads)
    BigVec<int> cellpos(N);
    std::vector<int> order((size_t)nboxes);
    host_parallel_for(N, [&](long long e) {
      const int i = (int)(e % n1), j = (int)((e / n1) % n2), k = (int)(e / s3);
      const int I = i / g.B1, J = j / g.B2, K = k / g.B3, ip = i % g.B1, jp = j % g.B2, kp = k % g.B3;
      const long long b = I + (long long)g.NB1 * (J + (long long)g.NB2 * K);
      const long long t = (long long)box_slot(g, b, ip + jp + kp, jp + g.B2 * kp);
      cellmap[(size_t)t] = ext2int[e];
      cellpos[ext2int[e]] = (int)t;
    });
    const int nl = g.NB1 + g.NB2 + g.NB3 - 2;
    box_lptr.assign(nl + 1, 0);
    for (int K = 0; K < g.NB3; ++K) for (int J = 0; J < g.NB2; ++J) for (int I = 0; I < g.NB1; ++I) box_lptr[I + J + K + 1]++;
    for (int l = 0; l < nl; ++l) box_lptr[l + 1] += box_lptr[l];
    std::vector<int> fill(box_lptr.begin(), box_lptr.end() - 1);
    for (int K = 0; K < g.NB3; ++K) for (int J = 0; J < g.NB2; ++J) for (int I = 0; I < g.NB1; ++I)
      order[fill[I + J + K]++] = I + g.NB1 * (J + g.NB2 * K);
    box_cellmap.upload(cellmap.p, cellmap.n, stream); box_cellpos.upload(cellpos, stream); box_order.upload(order, stream);
    {
      std::vector<int> rev(order.rbegin(), order.rend());
      box_order_rev.upload(rev, stream);
      box_flags.alloc(2 * (size_t)nboxes); box_flags.zero(stream);
      box_ticket.alloc(4); box_ticket.zero(stream);   // [0, 1]: the queued solves' counters, [2, 3]: the streamed solves'
      box_epoch = 0; box_ticket_base[0] = box_ticket_base[1] = 0;
    }
    for (int k = 0; k < 3; ++k) { box_L[k].alloc((size_t)slots); box_U[k].alloc((size_t)slots); }
    box_Di.alloc((size_t)slots); box_v.alloc((size_t)slots); box_scratch.alloc(128);
    HIP_CHECK(hipStreamSynchronize(stream));
    box = g; box_slots = slots; box_ok = true;
  }

  // current matrix values in the caller's ordering, columns sorted
  void export_csr(int *rowptr, int *col, double *val) {
    std::vector<double> v(nnz);
    HIP_CHECK(hipMemcpyAsync(v.data(), d_val.p, (size_t)nnz * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    int pos = 0;
    std::vector<std::pair<int, double>> row;
    for (int e = 0; e < nrow; ++e) {
      rowptr[e] = pos;
      const int i = row_ext2int(e);
      row.clear();
      for (int k = h_rowptr[i]; k < h_rowptr[i + 1]; ++k) row.emplace_back(row_int2ext(h_col[k]), v[k]);
      std::sort(row.begin(), row.end());
      for (auto &pr : row) { col[pos] = pr.first; val[pos] = pr.second; ++pos; }
    }
    rowptr[nrow] = pos;
  }

  void set_option(const std::string &key, const std::string &val) {
    auto num = [&]() {
      char *end = nullptr;
      const double v = std::strtod(val.c_str(), &end);
      if (end == val.c_str() || *end != '\0') throw Error("option '" + key + "': '" + val + "' is not a number");
      return v;
    };
    if (key == "ksp_type") {
      if (val == "preonly") opt.ksp_type = 0; else if (val == "gmres") opt.ksp_type = 1; else if (val == "bcgs") opt.ksp_type = 2;
      else throw Error("unknown ksp_type '" + val + "'");
    } else if (key == "pc_type") {
      if (val == "none") opt.pc_type = 0; else if (val == "bjacobi") opt.pc_type = 2; else if (val == "jacobi") opt.pc_type = 3;
      else if (val == "ilu") { opt.pc_type = 2; opt.pc_is_ilu = true; if (setup_done) check_ilu(); }
      else throw Error("unknown pc_type '" + val + "' (GPU preconditioners: bjacobi, ilu [vertical-only grids], jacobi, none)");
      if (val != "ilu") opt.pc_is_ilu = false;
      pc_ready = false;
    } else if (key == "ksp_rtol") opt.ksp_rtol = num();
    else if (key == "ksp_atol") opt.ksp_atol = num();
    else if (key == "ksp_max_it") opt.ksp_max_it = (int)num();
    else if (key == "ksp_gmres_restart") {
      const int m = (int)num();
      if (m < 1 || m > 4096) throw Error("ksp_gmres_restart must lie in [1, 4096]");   // (m + 1 basis vectors are held on the device)
      opt.gmres_restart = m; vv.clear();
    }
    else if (key == "debug_fail_at_fetch") debug_fail_at_fetch = (int)num();
    else if (key == "snes_rtol") opt.snes_rtol = num();
    else if (key == "snes_atol") opt.snes_atol = num();
    else if (key == "snes_stol") opt.snes_stol = num();
    else if (key == "snes_max_it") opt.snes_max_it = (int)num();
    else if (key == "snes_linesearch_type") {
      if (val == "basic") opt.linesearch_type = 1; else if (val == "bt") opt.linesearch_type = 2; else if (val == "l2") opt.linesearch_type = 3;
      else if (val == "cp") opt.linesearch_type = 4;
      else throw Error("Unknown default linesearch option '" + val + "' (basic | bt | l2 | cp)");   // SystemOfEquationsBaseType.F90:424-430
    } else if (key == "use_dynamic_linesearch") opt.use_dynamic_linesearch = num() != 0;
    else if (key == "spmv_variant") {
      const int v = (int)num();
      if (!(v == -1 || v == 0 || v == 1 || v == 17 || (v >= 20 && v <= 23)))
        throw Error("spmv_variant: -1 auto, 0 lane per row, 1 block CSR-stream, 17 pipelined wave CSR-stream, 20 / 21 compact index, 22 / 23 windowed compact index");
      opt.spmv_variant = v;
    }
    else if (key == "assembly") {
      if (val == "auto") opt.assembly = -1; else if (val == "general") opt.assembly = 0; else if (val == "columns") opt.assembly = 1;
      else throw Error("unknown assembly '" + val + "'");
    } else if (key == "nseg") opt.nseg = (int)num();
    else if (key == "fused_assembly") opt.fused_assembly = num() != 0;
    else if (key == "ilu_graph") opt.ilu_graph = num() != 0;
    else if (key == "ilu_graph_nodes") {   // kernel nodes per replayed graph (the sequences are cut into chunks of this size)
      opt.ilu_graph_nodes = std::max(1, (int)num());
      ilu_graph.destroy(); box_graph.destroy(); box_fgraph.destroy();
    }
    else if (key == "th_fused_dots") opt.th_fused_dots = num() != 0;
    else if (key == "th_spmv_cols") opt.th_spmv_cols = num() != 0;
    else if (key == "spmv_cols_dots") opt.spmv_cols_dots = num() != 0;   // line-search product on soil columns without an index
    else if (key == "lazy_tri_l") opt.lazy_tri_l = num() != 0;   // column LU: elimination factors not stored by the march where the fused solves redo them
    else if (key == "th_spmv_tri") opt.th_spmv_tri = num() != 0;   // coupled system on columns: M^{-1} A v + dots in one kernel
    else if (key == "th_stage") opt.th_stage = num() != 0;
    else if (key == "th_lazy_factor") opt.th_lazy_factor = num() != 0;   // coupled system: block factorisation inside the first solve's sweep
    else if (key == "th_inv_fly") opt.th_inv_fly = num() != 0;           // ... and the fused products redo the pivots' inverses instead of reading them
    else if (key == "setup_device") { need_setup(false); opt.setup_device = num() != 0; }   // incidence lists built on the device (0: on the host)
    else if (key == "setup_verify") { need_setup(false); opt.setup_verify = num() != 0; }   // ... both, compared (an error if they differ)
    else if (key == "ilu_box") { opt.ilu_box = (int)num(); pc_ready = false; }   // the factors are repacked for the other solve path
    else if (key == "ilu_box_factor") { opt.ilu_box_factor = num() != 0; pc_ready = false; }
    else if (key == "ilu_box_img_out") opt.ilu_box_img_out = num() != 0;   // the solution leaves the box layout through an LDS image (k_box_gather_out)
    else if (key == "ilu_box_stream") opt.ilu_box_stream = num() != 0;     // box solves as streams: pencils by ticket, faces handed over 8 steps at a time
    else if (key == "ilu_box_fused_io") opt.ilu_box_fused_io = num() != 0; // ... reading r and writing z through the slot map themselves
    else if (key == "ilu_box_queue") opt.ilu_box_queue = num() != 0;       // box solves: one launch per triangle (tickets + flags)
    else if (key == "ilu_box_workers") opt.ilu_box_workers = std::max(1, (int)num());   // ... on this many resident waves
    else if (key == "ilu_box_len") {
      const int v = (int)num();
      if (v < 0 || v > kBoxMaxLen) throw Error("ilu_box_len must be between 0 and " + std::to_string(kBoxMaxLen));
      if (box_tried) throw Error("ilu_box_len must be set before the first preconditioner set-up");
      opt.ilu_box_len = v;
    }
    else if (key == "ilu_chain") opt.ilu_chain = (int)num();   // 0 per-level launches, 1 chained where it pays, 2 always chained
    else if (key == "ilu_chain_tickets") opt.ilu_chain_tickets = num() != 0;
    else if (key == "ilu_chain_spins") opt.ilu_chain_spins = (unsigned)num();   // polls before a chained wait gives up (tests: 0)
    else if (key == "fused_trials") opt.fused_trials = (int)num();
    else if (key == "fused_predict") opt.fused_predict = num() != 0;
    else if (key == "fused_ls_dots") opt.fused_ls_dots = num() != 0;
    else if (key == "ell_assembly") opt.ell_assembly = num() != 0;
    else if (key == "ls_ahead") opt.ls_ahead = num() != 0;
    else if (key == "wide_xr") opt.wide_xr = num() != 0;
    else if (key == "host_poll") opt.host_poll = num() != 0;
    else if (key == "collective_timeout") opt.collective_timeout = num();
    else if (key == "ls_spec") opt.ls_spec = num() != 0;
    else if (key == "ls_fold") opt.ls_fold = num() != 0;
    else if (key == "cols3d") opt.cols3d = num() != 0;
    else if (key == "cols3d_rows") opt.cols3d_rows = num() != 0;
    else if (key == "ls_trace") opt.ls_trace = num() != 0;   // one line per bt line search on stderr (development aid)
    else if (key == "fused_spmv_pc") opt.fused_spmv_pc = num() != 0;
    else if (key == "fused_first") opt.fused_first = num() != 0;
    else if (key == "lazy_r") opt.lazy_r = num() != 0;
    else if (key == "fused_waxpy") opt.fused_waxpy = num() != 0;
    else if (key == "bc_inline") opt.bc_inline = num() != 0;
    else if (key == "fold_finalize") opt.fold_finalize = num() != 0;
    else if (key == "thermal_reference_defects") opt.thermal_reference_defects = num() != 0;
    else if (key == "spmv_tpw") opt.spmv_tpw = (int)num();
    else if (key == "soil_classes") { opt.soil_classes = num() != 0; soil_dirty = true; }
    else if (key == "level_geometry") { opt.level_geometry = num() != 0; soil_dirty = true; }
    else if (key == "tri_lds") opt.tri_lds = (int)num();
    else if (key == "tri_group") opt.tri_group = (int)num();
    else if (key == "defer_test0") opt.defer_test0 = num() != 0;
    else if (key == "fused_rows") opt.fused_rows = num() != 0;
    else if (key == "track_mass_exchange") opt.track_mass_exchange = num() != 0;
    else if (key == "temperature") temperature = num();
    else if (key == "problem_type") {
      if (setup_done || !bcs.empty() || !sss.empty()) throw Error("problem_type must be set before conditions are added");
      if (val == "thermal") { thermal = true; th = false; }
      else if (val == "vsfm") { thermal = false; th = false; }
      else if (val == "th") { thermal = false; th = true; if (!opt.pc_is_ilu && opt.pc_type == 2) { opt.pc_is_ilu = true; } }   // PETSc's default PC
      else throw Error("unknown problem_type '" + val + "' (vsfm | thermal | th)");
    } else if (key == "int_energy_enthalpy_type") {
      const int t = (int)num();
      if (t != 1 && t != 2) throw Error("int_energy_enthalpy_type must be 1 (constant) or 2 (IFC67)");
      ie_type = t;
    } else if (key == "cnfac") cnfac = num();
    else if (key == "profile") {   // 0 off, 1 every kernel class, 2 the roofline kernels only: the CSR SpMV launches and the SpMV fused with
      prof.flush();                // the column solve (with ls_fold the only product launches a Newton iteration on soil columns has left)
      const int m = (int)num();
      prof.enabled = m != 0;
      prof.mask = m == 2 ? ((1u << KC_SPMV) | (1u << KC_SPMV_PC)) : ~0u;
    }
    else if (key == "profile_stride") { prof.flush(); prof.stride = std::max(1, (int)num()); }
    else throw Error("unknown option '" + key + "'");
  }

  double algorithmic_bytes(int cls) const {
    const double N = n, NNZ = nnz;
    switch (cls) {
      case KC_SPMV: return 12.0 * NNZ + 20.0 * N + 4.0;
      case KC_PC_APPLY: return 40.0 * N;                   // 3 diagonals + rhs in, x out
      case KC_SPMV_PC: return 12.0 * NNZ + 20.0 * N + 4.0 + 40.0 * N;   // the two it replaces, as SURVEY counts them
      case KC_PC_SETUP: return 8.0 * NNZ + 24.0 * N;
      case KC_RESIDUAL: return 96.0 * N;                   // P, 5 soil arrays, accum_prev, 4 geometry arrays in; F out
      case KC_JACOBIAN: return 84.0 * N + 8.0 * NNZ;       // same reads + rowptr, CSR values out
      default: return 0.0;
    }
  }
};

}  // namespace mpp

// ===========================================================================================
// C ABI
// ===========================================================================================
using mpp::Solver;

struct mppgpu_handle_s {
  Solver s;
};

static thread_local std::string g_create_error;

#define API_BEGIN(h)                    \
  if (!(h)) return 1;                   \
  Solver &S = (h)->s;                   \
  try {                                 \
    (void)hipSetDevice(S.device);       \
    mpp::poison_stream() = S.stream;
#define API_END()                       \
  }                                     \
  catch (const std::exception &e) {     \
    S.err = e.what();                   \
    return 2;                           \
  }                                     \
  return 0;
// entry points that take part in collectives: a failure on this rank releases the in-process peers that are (or will be)
// waiting for it, each with an error of its own, instead of leaving them blocked
#define API_END_COLLECTIVE()            \
  }                                     \
  catch (const std::exception &e) {     \
    S.err = e.what();                   \
    S.comm.abort_peers();               \
    return 2;                           \
  }                                     \
  return 0;

extern "C" {

int mppgpu_create(int device_id, mppgpu_handle *out) {
  if (!out) return 1;
  *out = nullptr;
  try {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
      throw mpp::Error(std::string("no usable HIP device (this library has no CPU path): ") + hipGetErrorString(e));
    if (device_id < 0 || device_id >= count) throw mpp::Error("device_id out of range");
    HIP_CHECK(hipSetDevice(device_id));
    auto *h = new mppgpu_handle_s();
    h->s.device = device_id;
    HIP_CHECK(hipStreamCreateWithFlags(&h->s.stream, hipStreamNonBlocking));
    h->s.prof.stream = h->s.stream;
    *out = h;
  } catch (const std::exception &e) {
    g_create_error = e.what();
    return 2;
  }
  return 0;
}

int mppgpu_destroy(mppgpu_handle h) {
  if (!h) return 1;
  (void)hipSetDevice(h->s.device);
  if (h->s.stream) (void)hipStreamSynchronize(h->s.stream);
  delete h;
  return 0;
}

const char *mppgpu_last_error(mppgpu_handle h) { return h ? h->s.err.c_str() : g_create_error.c_str(); }

int mppgpu_synchronize(mppgpu_handle h) {
  API_BEGIN(h)
  HIP_CHECK(hipStreamSynchronize(S.stream));
  API_END()
}

int mppgpu_set_mesh(mppgpu_handle h, int ncells, int nlev, const double *vol, const int *is_active) {
  API_BEGIN(h)
  S.set_mesh(ncells, nlev, vol, is_active);
  API_END()
}
int mppgpu_set_connections(mppgpu_handle h, int nconn, const int *id_up, const int *id_dn, const double *dist_up,
                           const double *dist_dn, const double *area, const double *unit_vec) {
  API_BEGIN(h)
  S.set_connections(nconn, id_up, id_dn, dist_up, dist_dn, area, unit_vec);
  API_END()
}
int mppgpu_add_condition(mppgpu_handle h, int kind, int cond_type, int nconn, const int *id_dn, const double *dist_up,
                         const double *dist_dn, const double *area, const double *unit_vec, int swap_order,
                         int *cond_id) {
  API_BEGIN(h)
  const int id = S.add_condition(kind, cond_type, nconn, id_dn, dist_up, dist_dn, area, unit_vec, swap_order);
  if (cond_id) *cond_id = id;
  API_END()
}
int mppgpu_pre_step_dt(mppgpu_handle h) {
  API_BEGIN(h)
  S.pre_step_dt();
  API_END()
}
int mppgpu_post_step_dt(mppgpu_handle h) {
  API_BEGIN(h)
  S.post_step_dt();
  API_END()
}
int mppgpu_set_ghost_cells(mppgpu_handle h, int nghost) {
  API_BEGIN(h)
  S.set_ghost_cells(nghost);
  API_END()
}
int mppgpu_set_halo(mppgpu_handle h, int nneigh, const int *peer_rank, const int *send_ptr, const int *send_cells,
                    const int *recv_count) {
  API_BEGIN(h)
  S.set_halo(nneigh, peer_rank, send_ptr, send_cells, recv_count);
  API_END()
}
int mppgpu_set_lateral_connections(mppgpu_handle h, int nconn, const int *id_up, const int *id_dn, const double *dist_up,
                                   const double *dist_dn, const double *area, const double *unit_vec) {
  API_BEGIN(h)
  S.set_lateral_connections(nconn, id_up, id_dn, dist_up, dist_dn, area, unit_vec);
  API_END()
}
int mppgpu_compute_lateral_flux(mppgpu_handle h, double dt, int ss_cond_id) {
  API_BEGIN(h)
  S.need_setup(true);
  S.compute_lateral_flux(dt, ss_cond_id);
  API_END()
}
int mppgpu_get_condition_values(mppgpu_handle h, int kind, int cond_id, int n, double *values) {
  API_BEGIN(h)
  S.get_condition_values(kind, cond_id, n, values);
  API_END()
}
int mppgpu_set_source_sink_auxvar(mppgpu_handle h, int var_type, int n, const double *values) {
  API_BEGIN(h)
  S.set_ss_auxvar(var_type, n, values);
  API_END()
}
int mppgpu_th_add_condition(mppgpu_handle h, int igoveqn, int kind, int cond_type, int nconn, const int *id_dn,
                            const double *dist_up, const double *dist_dn, const double *area, const double *unit_vec,
                            int swap_order, int *cond_id) {
  API_BEGIN(h)
  if (!S.th) throw mpp::Error("problem_type is not th");
  const int id = S.add_condition(kind, cond_type, nconn, id_dn, dist_up, dist_dn, area, unit_vec, swap_order, igoveqn);
  if (cond_id) *cond_id = id;
  API_END()
}
int mppgpu_th_set_soils(mppgpu_handle h, const double *heat_cap, const double *tcw, const double *tcd, const double *talpha,
                        const double *den_soil, const double *px, const double *py, const double *pz) {
  API_BEGIN(h)
  if (!S.th) throw mpp::Error("problem_type is not th");
  const int n = S.n;
  if (!n) throw mpp::Error("mesh not set");
  S.the_heat_cap.assign(heat_cap, heat_cap + n); S.the_tcw.assign(tcw, tcw + n); S.the_tcd.assign(tcd, tcd + n);
  S.the_talpha.assign(talpha, talpha + n); S.the_dsoil.assign(den_soil, den_soil + n);
  if (px && py && pz) { S.the_permx.assign(px, px + n); S.the_permy.assign(py, py + n); S.the_permz.assign(pz, pz + n); }
  else { S.the_permx.clear(); S.the_permy.clear(); S.the_permz.clear(); }
  S.soil_dirty = true;
  API_END()
}
int mppgpu_set_soil_permeability(mppgpu_handle h, const double *px, const double *py, const double *pz) {
  API_BEGIN(h)
  S.set_perm(px, py, pz);
  API_END()
}
int mppgpu_set_soil_porosity(mppgpu_handle h, const double *por) {
  API_BEGIN(h)
  S.set_por(por);
  API_END()
}
int mppgpu_set_soil_porosity_linear(mppgpu_handle h, const double *por_base, const double *pressure_reference, const double *slope) {
  API_BEGIN(h)
  S.set_por_linear(por_base, pressure_reference, slope);
  API_END()
}
int mppgpu_set_saturation_function(mppgpu_handle h, const int *satfunc_type, const double *alpha, const double *lambda,
                                   const double *sat_res) {
  API_BEGIN(h)
  S.set_satfunc(satfunc_type, alpha, lambda, sat_res);
  API_END()
}
int mppgpu_set_density_type(mppgpu_handle h, int density_type) {
  API_BEGIN(h)
  if (density_type < 1 || density_type > 3) throw mpp::Error("unknown density type");
  S.density_type = density_type;
  API_END()
}
int mppgpu_set_option(mppgpu_handle h, const char *key, const char *value) {
  API_BEGIN(h)
  S.set_option(key ? key : "", value ? value : "");
  API_END()
}
int mppgpu_setup(mppgpu_handle h) {
  API_BEGIN(h)
  S.setup();
  API_END()
}
int mppgpu_restart(mppgpu_handle h, int n, const double *pressure) {
  API_BEGIN(h)
  S.need_setup(true);
  if (n != S.nrow) throw mpp::Error("restart: size(data_1d) /= number of unknowns (ncells_local, or 2 ncells_local [P; T] for the coupled system)");
  S.put_state(pressure, S.soln.p);
  S.vec_copy(S.soln.p, S.soln_prev.p);
  S.vec_copy(S.soln.p, S.soln_prev_clm.p);   // the drivers VecCopy soln into soln_prev_clm after Restart
  HIP_CHECK(hipStreamSynchronize(S.stream));
  API_END()
}
int mppgpu_set_data(mppgpu_handle h, int auxvar_type, int var_type, int cond_id, int n, const double *data) {
  API_BEGIN(h)
  S.set_data(auxvar_type, var_type, cond_id, n, data);
  API_END()
}
int mppgpu_step_dt(mppgpu_handle h, double dt, int nstep, int *converged, int *converged_reason) {
  (void)nstep;
  API_BEGIN(h)
  int c = 0, r = 0;
  S.step_dt(dt, &c, &r);
  if (converged) *converged = c;
  if (converged_reason) *converged_reason = r;
  API_END_COLLECTIVE()
}
int mppgpu_get_data(mppgpu_handle h, int auxvar_type, int var_type, int cond_id, int n, double *data) {
  API_BEGIN(h)
  S.get_data(auxvar_type, var_type, cond_id, n, data);
  API_END()
}
int mppgpu_get_counters(mppgpu_handle h, long long *newton_its, long long *linear_its, long long *time_cuts,
                        long long *function_evals) {
  API_BEGIN(h)
  if (newton_its) *newton_its = S.cum_newton;
  if (linear_its) *linear_its = S.cum_linear;
  if (time_cuts) *time_cuts = S.cum_cuts;
  if (function_evals) *function_evals = S.cum_funcs;
  API_END()
}

int mppgpu_get_step_history(mppgpu_handle h, int max_n, double *dt_sub, int *n) {
  API_BEGIN(h)
  const int have = (int)S.last_substeps.size();
  if (n) *n = have;
  for (int i = 0; i < have && i < max_n; ++i) dt_sub[i] = S.last_substeps[i];
  API_END()
}

int mppgpu_set_dtime(mppgpu_handle h, double dt) {
  API_BEGIN(h)
  if (!(dt > 0.0)) throw mpp::Error("dt must be positive");
  S.dtime = dt;
  API_END()
}
int mppgpu_pre_solve(mppgpu_handle h) {
  API_BEGIN(h)
  S.ensure_ready();
  if (S.dtime <= 0.0) throw mpp::Error("time step not set");
  S.op_pre_solve();
  HIP_CHECK(hipStreamSynchronize(S.stream));
  API_END_COLLECTIVE()
}
int mppgpu_residual(mppgpu_handle h, const double *x, double *f) {
  API_BEGIN(h)
  S.ensure_ready();
  S.put_state(x, S.W.p);
  S.op_residual(S.W.p, S.G.p);
  S.get_state(S.G.p, f);
  API_END_COLLECTIVE()
}
int mppgpu_jacobian_nnz(mppgpu_handle h, int *nnz) {
  API_BEGIN(h)
  S.need_setup(true);
  if (nnz) *nnz = S.nnz;
  API_END()
}
int mppgpu_jacobian(mppgpu_handle h, const double *x, int *rowptr, int *col, double *val) {
  API_BEGIN(h)
  S.ensure_ready();
  S.put_state(x, S.W.p);
  S.op_jacobian(S.W.p);
  S.export_csr(rowptr, col, val);
  API_END_COLLECTIVE()
}
int mppgpu_spmv(mppgpu_handle h, const double *x, double *y) {
  API_BEGIN(h)
  S.ensure_ready();
  S.put_state(x, S.W.p);
  S.op_spmv(S.W.p, S.G.p);
  S.get_state(S.G.p, y);
  API_END_COLLECTIVE()
}
int mppgpu_pc_apply(mppgpu_handle h, const double *r, double *z) {
  API_BEGIN(h)
  S.ensure_ready();
  S.put_state(r, S.W.p);
  S.op_pc_setup();
  S.op_pc_apply(S.W.p, S.G.p);
  S.get_state(S.G.p, z);
  API_END_COLLECTIVE()
}
int mppgpu_ksp_solve(mppgpu_handle h, const double *b, double *x, int *its, int *reason) {
  API_BEGIN(h)
  S.ensure_ready();
  S.put_state(b, S.G.p);
  int it = 0;
  const int r = S.ksp_solve(S.G.p, S.Y.p, &it);
  S.get_state(S.Y.p, x);
  if (its) *its = it;
  if (reason) *reason = r;
  API_END_COLLECTIVE()
}

/* ---- thermal KSP path ---- */
int mppgpu_thermal_set_soils(mppgpu_handle h, const int *itype, const int *is_soil_shallow, const double *watsat,
                             const double *csol, const double *tkmg, const double *tkdry, const double *dz) {
  API_BEGIN(h)
  if (!S.n) throw mpp::Error("mesh not set");
  if (!S.thermal) throw mpp::Error("problem_type is not thermal");
  const int n = S.n;
  for (int i = 0; i < n; ++i)
    if (itype[i] < 1 || itype[i] > 3) throw mpp::Error("thermal_set_soils: itype must be 1 (soil/crop), 2 (wetland) or 3 (land ice)");
  S.th_itype.assign(itype, itype + n); S.th_shallow.assign(is_soil_shallow, is_soil_shallow + n);
  S.th_por.assign(watsat, watsat + n); S.th_csol.assign(csol, csol + n); S.th_tkmg.assign(tkmg, tkmg + n);
  S.th_tkdry.assign(tkdry, tkdry + n); S.th_dz.assign(dz, dz + n);
  S.th_cell_dirty = true;
  API_END()
}
int mppgpu_thermal_set_soln_prev_clm(mppgpu_handle h, int n, const double *temperature) {
  API_BEGIN(h)
  S.need_setup(true);
  if (!S.thermal) throw mpp::Error("problem_type is not thermal");
  if (n != S.n) throw mpp::Error("size(data_1d) /= ncells_local");
  S.put_cells(temperature, S.soln_prev_clm.p);
  API_END()
}
int mppgpu_thermal_pre_step_dt(mppgpu_handle h) {
  API_BEGIN(h)
  S.need_setup(true);
  if (!S.thermal) throw mpp::Error("problem_type is not thermal");
  S.vec_copy(S.soln_prev_clm.p, S.soln_prev.p);
  S.vec_copy(S.soln_prev_clm.p, S.soln.p);
  HIP_CHECK(hipStreamSynchronize(S.stream));
  API_END()
}
int mppgpu_thermal_step_dt(mppgpu_handle h, double dt, int *converged, int *linear_its) {
  API_BEGIN(h)
  int c = 0, it = 0;
  S.thermal_step_dt(dt, &c, &it);
  if (converged) *converged = c;
  if (linear_its) *linear_its = it;
  API_END_COLLECTIVE()
}
int mppgpu_thermal_get_soln(mppgpu_handle h, int n, double *temperature) {
  API_BEGIN(h)
  S.need_setup(true);
  if (n > S.n) throw mpp::Error("size(data_1d) > ncells_local");
  S.get_cells(S.soln.p, temperature, n);
  API_END()
}
int mppgpu_thermal_compute_rhs(mppgpu_handle h, double dt, double *b) {
  API_BEGIN(h)
  S.dtime = dt;
  S.thermal_assemble(0);
  S.get_cells(S.F.p, b, S.n);
  API_END()
}
int mppgpu_thermal_compute_operators(mppgpu_handle h, double dt, int *rowptr, int *col, double *val) {
  API_BEGIN(h)
  S.dtime = dt;
  S.thermal_assemble(1);
  S.export_csr(rowptr, col, val);
  API_END()
}

int mppgpu_kernel_stats(mppgpu_handle h, const char *name, long long *launches, double *total_ms,
                        double *algorithmic_bytes_per_launch) {
  API_BEGIN(h)
  S.prof.flush();
  int cls = -1;
  for (int k = 0; k < mpp::KC_COUNT; ++k)
    if (name && std::strcmp(name, mpp::kclass_name(k)) == 0) cls = k;
  if (cls < 0) throw mpp::Error(std::string("unknown kernel class '") + (name ? name : "") + "'");
  if (launches) *launches = S.prof.launches[cls];
  if (total_ms) *total_ms = S.prof.total_ms[cls];
  if (algorithmic_bytes_per_launch) *algorithmic_bytes_per_launch = S.algorithmic_bytes(cls);
  API_END()
}
int mppgpu_kernel_calls(mppgpu_handle h, const char *name, long long *calls, int *stride) {
  API_BEGIN(h)
  int cls = -1;
  for (int k = 0; k < mpp::KC_COUNT; ++k)
    if (name && std::strcmp(name, mpp::kclass_name(k)) == 0) cls = k;
  if (cls < 0) throw mpp::Error(std::string("unknown kernel class '") + (name ? name : "") + "'");
  if (calls) *calls = S.prof.calls[cls];
  if (stride) *stride = S.prof.stride;
  API_END()
}
int mppgpu_reset_stats(mppgpu_handle h) {
  API_BEGIN(h)
  S.prof.reset();
  S.comm.n_allreduce = 0; S.comm.n_exchange = 0;
  S.n_fetch = 0;
  API_END()
}
int mppgpu_get_comm_stats(mppgpu_handle h, long long *allreduces, long long *halo_exchanges, long long *chain_fallbacks,
                          long long *host_readbacks) {
  API_BEGIN(h)
  if (host_readbacks) *host_readbacks = S.n_fetch;
  if (allreduces) *allreduces = S.comm.n_allreduce;
  if (halo_exchanges) *halo_exchanges = S.comm.n_exchange;
  if (chain_fallbacks) *chain_fallbacks = S.chain_fallbacks;
  API_END()
}

int mppgpu_get_halo_stats(mppgpu_handle h, long long *neighbours, long long *cells_sent, long long *cells_received,
                          long long *bytes_received_per_exchange, long long *max_cells_from_one_neighbour) {
  API_BEGIN(h)
  const auto &hp = S.halo;
  long long nb = 0, sent = 0, recv = 0, mx = 0;
  if (S.nghost > 0 && hp.set) {
    nb = (long long)hp.peer.size();
    sent = (long long)hp.send_cells.size();
    recv = hp.recv_ptr.empty() ? 0 : hp.recv_ptr.back();
    for (size_t k = 0; k + 1 < hp.recv_ptr.size(); ++k) mx = std::max<long long>(mx, hp.recv_ptr[k + 1] - hp.recv_ptr[k]);
  }
  if (neighbours) *neighbours = nb;
  if (cells_sent) *cells_sent = sent;
  if (cells_received) *cells_received = recv;
  if (bytes_received_per_exchange) *bytes_received_per_exchange = 8 * recv * (S.th ? 2 : 1);
  if (max_cells_from_one_neighbour) *max_cells_from_one_neighbour = mx;
  API_END()
}

int mppgpu_get_assembly_info(mppgpu_handle h, int n, int *info) {
  API_BEGIN(h)
  S.ensure_ready();
  const bool cols = S.use_cols() && !S.th && !S.thermal;
  const int v[14] = {cols ? S.ncls : 0, cols && S.lev_geo ? 1 : 0, cols && S.lev_pv ? 1 : 0, cols && S.rows_complete() ? 1 : 0,
                     cols ? S.pick_nseg() : 0, S.spmv_variant(), (S.opt.pc_type == 2 && !S.th && S.tri_lds()) ? 1 : 0,
                     (int)std::min<long long>(S.spec_misses, 2147483647LL), S.opt.ilu_graph ? 1 : 0,
                     (int)(S.box_graph.execs.size() + S.ilu_graph.execs.size()), S.setup_verified,
                     (S.opt.fused_ls_dots && (S.cols_index_free() || S.th_index_free())) ? 1 : 0,
                     (int)std::min<long long>(S.ls_folds, 2147483647LL), S.use_cols3d() ? 1 : 0};
  for (int i = 0; i < n && i < 14; ++i) info[i] = v[i];
  API_END()
}
int mppgpu_get_pc_info(mppgpu_handle h, int *path, int *launches, int *box_extents) {
  API_BEGIN(h)
  int p = 0, n = 0, ext[3] = {0, 0, 0};
  if (S.opt.pc_type == 1 && S.pc_ready) {
    if (S.box_active()) {
      p = 3; n = (S.box_stream() || S.opt.ilu_box_queue) ? 4 : 2 * ((int)S.box_lptr.size() - 1) + 2;
      ext[0] = S.box.B1; ext[1] = S.box.B2; ext[2] = S.box.B3;
    } else if (S.ilu_chained()) { p = 2; n = 5; }
    else { p = 1; n = (int)S.h_lptr.size() - 2 + (int)S.h_uptr.size() - 1 + 3; }
  }
  if (path) *path = p;
  if (launches) *launches = n;
  if (box_extents) for (int k = 0; k < 3; ++k) box_extents[k] = ext[k];
  API_END()
}

int mppgpu_bench_kernel(mppgpu_handle h, const char *name, int variant, int reps, double *avg_ms,
                        double *algorithmic_bytes) {
  API_BEGIN(h)
  S.ensure_ready();
  if (reps < 1) reps = 1;
  const std::string k = name ? name : "";
  if (S.dtime <= 0.0) S.dtime = 3600.0;
  hipEvent_t a, b;
  HIP_CHECK(hipEventCreate(&a));
  HIP_CHECK(hipEventCreate(&b));
  int cls;
  auto once = [&]() {
    if (k == "spmv") {
      cls = mpp::KC_SPMV;
      if (variant == 22 || variant == 23) {
        if (!S.multibase_index) throw mpp::Error("no multi-base index for this pattern");
        mpp::launch_spmv_multibase(S.opt.spmv_tpw > 0 ? S.opt.spmv_tpw : (variant == 22 ? 2 : 4), S.nrow, S.max_row_nnz, S.d_tileptr.p, S.d_roff.p, S.d_mcol.p, S.d_tbase.p, S.d_val.p, S.soln.p, S.G.p, S.stream);
      } else if (variant == 20 || variant == 21) {
        if (!S.compact_index) throw mpp::Error("no compact index for this pattern");
        mpp::launch_spmv_compact(S.opt.spmv_tpw > 0 ? S.opt.spmv_tpw : (variant == 20 ? 2 : 4), S.nrow, S.max_row_nnz, S.compact_split, S.d_tileptr.p, S.d_roff.p, S.d_dcol.p, S.d_val.p, S.soln.p, S.G.p, S.stream);
      } else mpp::launch_spmv(variant, S.nrow, S.max_row_nnz, S.d_rowptr.p, S.d_col.p, S.d_val.p, S.soln.p, S.G.p, S.stream);
    }
    else if (k == "pc_apply") { cls = mpp::KC_PC_APPLY; S.op_pc_setup(); S.op_pc_apply(S.F.p, S.G.p); }
    else if (k == "tri_back") {   // back substitution of a forward-eliminated right-hand side
      cls = mpp::KC_PC_APPLY; S.op_pc_setup();
      if (!S.fwd_elim.p) S.fwd_elim.alloc(S.nrow);
      mpp::launch_tri_back_dots(S.tri_args(), S.fwd_elim.p, S.G.p, S.col_partials.p, S.stream);
    }
    else if (k == "spmv_dots") {   // the line search's SpMV with its three reductions; variant > 0: tiles per wave
      cls = mpp::KC_REDUCE;
      if (!S.spmv_dots_fused()) throw mpp::Error("the fused SpMV + reductions kernel is not in use for this problem");
      const int keep = S.opt.spmv_tpw;
      if (variant > 0) S.opt.spmv_tpw = variant;
      S.op_spmv_dots_launch(S.Y.p, S.F.p, S.soln.p, 0, 0);
      S.opt.spmv_tpw = keep;
    }
    else if (k == "pc_apply_dots") {   // the column solve with its dot products; variant: -1 as configured, else tri_lds levels
      cls = mpp::KC_PC_APPLY; S.op_pc_setup();
      const int keep = S.opt.tri_lds;
      if (variant >= 0) S.opt.tri_lds = variant;       // 0: sweeps through z (k_tri_solve_dots), 1: LDS
      S.op_pc_apply_dots(S.F.p, S.G.p, S.W.p, 0, 0);
      S.opt.tri_lds = keep;
    }
    else if (k == "pc_setup") { cls = mpp::KC_PC_SETUP; S.pc_ready = false; S.op_pc_setup(); S.ensure_th_factor(true); }
    else if (k == "residual") { cls = mpp::KC_RESIDUAL; S.op_residual(S.soln.p, S.G.p); }
    else if (k == "jacobian") { cls = mpp::KC_JACOBIAN; S.op_jacobian(S.soln.p); }
    else if (k == "assemble_both") { cls = mpp::KC_JACOBIAN; S.op_residual_jacobian(S.soln.p, S.G.p); }
    else if (k == "bcgs_xr") { cls = mpp::KC_VEC; mpp::launch_bcgs_xr(S.n, false, false, S.Y.p, S.R.p, S.Pv.p, S.S.p, S.T.p, S.RP.p, S.sc.p, S.partials.p, S.stream); }
    else if (k == "waxpy") { cls = mpp::KC_VEC; mpp::launch_waxpy(S.n, S.W.p, S.sc.p, -1, 1.0, 0.5, S.soln.p, S.G.p, S.stream); }
    else if (k == "dots3") { cls = mpp::KC_REDUCE; S.dots_dev({{S.Y.p, S.Y.p}, {S.soln.p, S.soln.p}, {S.F.p, S.W.p}}); }
    else if (k == "copy") { cls = mpp::KC_VEC; mpp::launch_copy(S.n, S.soln.p, S.G.p, S.stream); }
    else if (k == "dot") { cls = mpp::KC_REDUCE; S.dots_dev({{S.soln.p, S.G.p}}); }
    else throw mpp::Error("unknown kernel '" + k + "'");
  };
  const bool prof_was = S.prof.enabled;
  S.prof.enabled = false;
  const int saved_funcs = S.nfuncs;
  const long long saved_cum = S.cum_funcs;
  once();  // warm-up
  HIP_CHECK(hipEventRecord(a, S.stream));
  for (int r = 0; r < reps; ++r) once();
  HIP_CHECK(hipEventRecord(b, S.stream));
  HIP_CHECK(hipEventSynchronize(b));
  float ms = 0.f;
  HIP_CHECK(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  S.prof.enabled = prof_was;
  S.nfuncs = saved_funcs; S.cum_funcs = saved_cum;
  if (avg_ms) *avg_ms = ms / reps;
  if (algorithmic_bytes) {
    const double N = S.n;
    *algorithmic_bytes = (k == "copy" || k == "dot") ? 16.0 * N : (k == "waxpy") ? 24.0 * N : (k == "bcgs_xr") ? 48.0 * N
                         : (k == "dots3") ? 32.0 * N : (k == "pc_apply_dots") ? 48.0 * N : (k == "tri_back") ? 32.0 * N : (k == "spmv_dots") ? 12.0 * S.nnz + 28.0 * N : (k == "assemble_both") ? 92.0 * N + 8.0 * S.nnz : S.algorithmic_bytes(cls);
  }
  API_END()
}

// ---- the constitutive laws on the host, for drivers that need them to build their inputs ----
#define HOST_LAW_BEGIN try {
#define HOST_LAW_END                    \
    return 0;                           \
  } catch (const std::exception &e) {   \
    g_create_error = e.what();          \
    return 1;                           \
  }
int mppgpu_host_density(int density_type, int n, const double *p, const double *t_K, double *den, double *dden_dp, double *dden_dt) {
  HOST_LAW_BEGIN
  if (density_type < mpp::kDenConstant || density_type > mpp::kDenIFC67) throw mpp::Error("unknown density type " + std::to_string(density_type));
  for (int i = 0; i < n; ++i) {
    double d, dp, dt;
    mpp::density_pt<true>(density_type, p[i], t_K[i], d, dp, dt, mpp::PowTab{nullptr, nullptr});
    if (den) den[i] = d;
    if (dden_dp) dden_dp[i] = dp;
    if (dden_dt) dden_dt[i] = dt;
  }
  HOST_LAW_END
}
int mppgpu_host_internal_energy_enthalpy(int ie_type, int n, const double *p, const double *t_K, const double *den,
                                         const double *dden_dt, const double *dden_dp, double *u, double *h, double *du_dt,
                                         double *dh_dt, double *du_dp, double *dh_dp) {
  HOST_LAW_BEGIN
  if (ie_type != mpp::kIEConstant && ie_type != mpp::kIEIFC67) throw mpp::Error("unknown internal-energy type " + std::to_string(ie_type));
  for (int i = 0; i < n; ++i) {
    const mpp::IEOut o = mpp::internal_energy_enthalpy<true>(ie_type, p[i], t_K[i], den[i], dden_dt[i], dden_dp[i], mpp::PowTab{nullptr, nullptr});
    if (u) u[i] = o.U;
    if (h) h[i] = o.H;
    if (du_dt) du_dt[i] = o.dU_dT;
    if (dh_dt) dh_dt[i] = o.dH_dT;
    if (du_dp) du_dp[i] = o.dU_dP;
    if (dh_dp) dh_dp[i] = o.dH_dP;
  }
  HOST_LAW_END
}
int mppgpu_host_sat_relperm(int n, const int *satfunc_type, const double *p, const double *sat_res, const double *alpha,
                            const double *lambda, double *sat, double *dsat_dp, double *kr, double *dkr_dp) {
  HOST_LAW_BEGIN
  for (int i = 0; i < n; ++i) {
    const int sf = mpp::map_sat_func(satfunc_type[i]);
    if (sf < 0) throw mpp::Error("unsupported saturation function " + std::to_string(satfunc_type[i]));
    mpp::Soil s{};
    s.sat_res = sat_res[i]; s.alpha = alpha[i]; s.m = lambda[i]; s.sf = sf;
    if (sf == mpp::kSatSBC) mpp::Solver::sbc_coefficients(satfunc_type[i], alpha[i], lambda[i], s.pu, s.ps, s.b2, s.b3);
    double a, da, k, dk;
    mpp::sat_relperm<true>(sf, s, p[i], 1.0, a, da, k, dk, mpp::PowTab{nullptr, nullptr});
    if (sat) sat[i] = a;
    if (dsat_dp) dsat_dp[i] = da;
    if (kr) kr[i] = k;
    if (dkr_dp) dkr_dp[i] = dk;
  }
  HOST_LAW_END
}

int mppgpu_comm_runtime_version(int *version) {
  try {
    const int v = mpp::Comm::runtime_version();
    if (version) *version = v;
    return 0;
  } catch (const std::exception &e) {
    g_create_error = e.what();
    return 1;
  }
}
int mppgpu_comm_get_unique_id(void *id128) {
  try {
    mpp::Comm::get_unique_id(id128);
  } catch (const std::exception &e) {
    g_create_error = e.what();
    return 2;
  }
  return 0;
}
int mppgpu_comm_init(mppgpu_handle h, int nranks, int rank, const void *id128) {
  API_BEGIN(h)
  S.comm.init(nranks, rank, id128);
  API_END()
}
int mppgpu_comm_init_local(mppgpu_handle *handles, int nranks) {
  if (!handles || nranks < 1) return 1;
  for (int r = 0; r < nranks; ++r) if (!handles[r]) return 1;
  try {
    auto g = std::make_shared<mpp::LocalGroup>(nranks);
    for (int r = 0; r < nranks; ++r) {
      (void)hipSetDevice(handles[r]->s.device);
      mpp::poison_stream() = handles[r]->s.stream;
      handles[r]->s.comm.init_local(g, r);
    }
  } catch (const std::exception &e) {
    handles[0]->s.err = e.what();
    return 2;
  }
  return 0;
}

}  // extern "C"
