/*
 * thermal_oracle.c -- CPU ORACLE (test infrastructure only) for MPP's soil heat-diffusion KSP path
 * (SURVEY.md section 8 f-1).  Plain-C restatement, every function cites the reference lines it follows:
 *   auxvars        src/mpp/auxvar/ThermalKSPTemperatureSoilAuxType.F90:84-168
 *   right-hand side src/mpp/ge/GoveqnThermalKSPTemperatureSoilType.F90:646-972
 *   operator        src/mpp/ge/GoveqnThermalKSPTemperatureSoilType.F90:975-1229
 *   time step       src/mpp/soe/SystemOfEquationsBaseType.F90:555-647 (SOEBaseStepDT_KSP),
 *                   src/mpp/soe/SystemOfEquationsThermalType.F90:391-481, 546-649
 * The linear solve is PETSc's (GMRES(30)+ILU(0) by default): the restatement in vsfm_oracle.c is reused.
 * Pinned by the reference's regression baselines regression_tests/thermal/thermal_mms{,_2d,_3d}.regression.baseline.
 *
 * Two reference defects (both invisible on the reference's own test problems) are NOT replicated unless
 * to_set_reference_defects(p, 1) asks for them:
 *   - the COND_DIRICHLET branch of the RHS uses the stale local `area` (GoveqnThermalKSPTemperatureSoilType.F90:906-908):
 *     the area of the last internal connection between two active cells (:787), or of the last COND_HEAT_FLUX boundary
 *     connection met before it in the loop over conditions (:911); by default the boundary connection's own area is used
 *     (a mesh without any such earlier connection reads an undefined variable in the reference: 0 here);
 *   - the COND_HEAT_FLUX operator term is written `dhsdT**area` (:1215); by default it is dhsdT*area.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "vsfm_oracle.h"

/* mpp_varcon.F90:14-28 */
#define CPLIQ 4.188e3
#define CPICE 2.11727e3
#define DENH2O 1.0e3
#define DENICE 0.917e3
#define TKICE 2.290
#define TKWAT 0.57
#define THK_BEDROCK 3.0
#define TFRZ 273.15

enum { TO_SOIL = 1, TO_WETLAND = 2, TO_ICE = 3 };
enum { TO_COND_DIRICHLET = 505, TO_COND_HEAT_FLUX = 507, TO_COND_HEAT_RATE = 511, TO_COND_BC = 501, TO_COND_SS = 502 };

typedef struct {
  double temperature, liq_areal_den, ice_areal_den, snow_water, tuning_factor;
  int num_snow_layer, itype, is_soil_shallow, is_active;
  double por, therm_cond_minerals, therm_cond_dry, heat_cap_minerals_puv;
  double therm_cond, heat_cap_pva;
  double condition_value, frac, dhsdT; /* BC auxvars */
} tauxvar;

typedef struct {
  int kind, cond_type, n;
  int *id_dn;
  double *dist_up, *dist_dn, *area, *value;
  tauxvar *aux;
} tcond;

typedef struct to_problem {
  int n, nc, nbc, nss;
  double *vol, *dz;
  tauxvar *aux;
  int *up, *dn;
  double *dist_up, *dist_dn, *area;
  tcond *bc, *ss;
  int *rowptr, *col, *diag;
  double *val, *rhs, *soln, *soln_prev, *soln_prev_clm;
  double dtime, cnfac;
  int reference_defects;
  int ksp_type, pc_type;
  double ksp_rtol;
  long cum_linear;
  vo_problem *lin; /* bare handle carrying the KSP options of vsfm_oracle.c */
} to_problem;

static void taux_init(tauxvar *a) { /* ThermKSPTempSoilAuxVarInit :41-63 */
  memset(a, 0, sizeof *a);
  a->tuning_factor = 1.0;
  a->itype = TO_SOIL;
  a->is_soil_shallow = 1;
  a->is_active = 1;
  a->frac = 1.0;
}

/* ThermKSPTempSoilAuxVarCompute :66-168 */
static void taux_compute(tauxvar *a, double dz) {
  if (a->itype == TO_SOIL) {
    if (a->is_soil_shallow) {
      double satw = (a->liq_areal_den / DENH2O + a->ice_areal_den / DENICE) / (dz * a->por);
      satw = fmin(1.0, satw);
      if (satw > .1e-6) {
        double dke = (a->temperature >= TFRZ) ? fmax(0.0, log10(satw) + 1.0) : satw;
        double fl = (a->liq_areal_den / (DENH2O * dz)) / (a->liq_areal_den / (DENH2O * dz) + a->ice_areal_den / (DENICE * dz));
        double dksat = a->therm_cond_minerals * pow(TKWAT, fl * a->por) * pow(TKICE, (1.0 - fl) * a->por);
        a->therm_cond = dke * dksat + (1.0 - dke) * a->therm_cond_dry;
      } else {
        a->therm_cond = a->therm_cond_dry;
      }
      a->heat_cap_pva = a->heat_cap_minerals_puv * (1.0 - a->por) * dz + a->ice_areal_den * CPICE + a->liq_areal_den * CPLIQ;
      if (a->num_snow_layer == 0) a->heat_cap_pva = a->heat_cap_pva + a->snow_water * CPICE;
    } else {
      a->therm_cond = THK_BEDROCK;
      a->heat_cap_pva = a->heat_cap_minerals_puv * (1.0 - a->por) * dz + a->ice_areal_den * CPICE + a->liq_areal_den * CPLIQ;
    }
    a->heat_cap_pva = a->heat_cap_pva / dz;
  } else if (a->itype == TO_WETLAND) {
    if (a->is_soil_shallow) {
      a->therm_cond = (a->temperature < TFRZ) ? TKICE : TKWAT;
      a->heat_cap_pva = a->ice_areal_den * CPICE + a->liq_areal_den * CPLIQ;
      if (a->num_snow_layer == 0) a->heat_cap_pva = a->heat_cap_pva + a->snow_water * CPICE;
      a->heat_cap_pva = a->heat_cap_pva / dz;
    } else {
      a->therm_cond = THK_BEDROCK;
      a->heat_cap_pva = a->heat_cap_minerals_puv;
    }
  } else { /* land ice */
    a->therm_cond = (a->temperature < TFRZ) ? TKICE : TKWAT;
    a->heat_cap_pva = a->ice_areal_den * CPICE + a->liq_areal_den * CPLIQ;
    if (a->num_snow_layer == 0) a->heat_cap_pva = a->heat_cap_pva + a->snow_water * CPICE;
    a->heat_cap_pva = a->heat_cap_pva / dz;
  }
}

to_problem *to_create(int n) {
  to_problem *p = (to_problem *)calloc(1, sizeof *p);
  p->n = n;
  p->vol = (double *)calloc(n, sizeof(double));
  p->dz = (double *)calloc(n, sizeof(double));
  p->aux = (tauxvar *)malloc(n * sizeof(tauxvar));
  for (int i = 0; i < n; i++) taux_init(&p->aux[i]);
  p->rhs = (double *)calloc(n, sizeof(double));
  p->soln = (double *)calloc(n, sizeof(double));
  p->soln_prev = (double *)calloc(n, sizeof(double));
  p->soln_prev_clm = (double *)calloc(n, sizeof(double));
  p->cnfac = 0.5; /* mpp_varcon.F90:28 */
  p->ksp_type = VO_KSP_GMRES;
  p->pc_type = VO_PC_ILU0;
  p->ksp_rtol = 1e-5;
  p->lin = vo_create(n);
  return p;
}

static void tcond_free(tcond *c) { free(c->id_dn); free(c->dist_up); free(c->dist_dn); free(c->area); free(c->value); free(c->aux); }

void to_destroy(to_problem *p) {
  if (!p) return;
  free(p->vol); free(p->dz); free(p->aux); free(p->up); free(p->dn); free(p->dist_up); free(p->dist_dn); free(p->area);
  for (int i = 0; i < p->nbc; i++) tcond_free(&p->bc[i]);
  for (int i = 0; i < p->nss; i++) tcond_free(&p->ss[i]);
  free(p->bc); free(p->ss); free(p->rowptr); free(p->col); free(p->diag); free(p->val);
  free(p->rhs); free(p->soln); free(p->soln_prev); free(p->soln_prev_clm);
  vo_destroy(p->lin);
  free(p);
}

int to_set_mesh(to_problem *p, const double *vol, const double *dz, const int *is_active) {
  memcpy(p->vol, vol, p->n * sizeof(double));
  memcpy(p->dz, dz, p->n * sizeof(double));
  for (int i = 0; i < p->n; i++) p->aux[i].is_active = is_active ? (is_active[i] != 0) : 1;
  return 0;
}

static int cmp_int2(const void *a, const void *b) { return (*(const int *)a > *(const int *)b) - (*(const int *)a < *(const int *)b); }
static int tfind(const to_problem *p, int row, int col) {
  for (int k = p->rowptr[row]; k < p->rowptr[row + 1]; k++) if (p->col[k] == col) return k;
  return -1;
}

int to_set_connections(to_problem *p, int nc, const int *up, const int *dn, const double *du, const double *dd, const double *area) {
  int n = p->n;
  p->nc = nc;
  p->up = (int *)malloc((nc + 1) * sizeof(int)); p->dn = (int *)malloc((nc + 1) * sizeof(int));
  p->dist_up = (double *)malloc((nc + 1) * sizeof(double)); p->dist_dn = (double *)malloc((nc + 1) * sizeof(double));
  p->area = (double *)malloc((nc + 1) * sizeof(double));
  for (int c = 0; c < nc; c++) {
    if (up[c] < 1 || up[c] > n || dn[c] < 1 || dn[c] > n) return -1;
    p->up[c] = up[c] - 1; p->dn[c] = dn[c] - 1; p->dist_up[c] = du[c]; p->dist_dn[c] = dd[c]; p->area[c] = area[c];
  }
  int *cnt = (int *)calloc(n + 1, sizeof(int));
  for (int i = 0; i < n; i++) cnt[i + 1] = 1;
  for (int c = 0; c < nc; c++) { cnt[p->up[c] + 1]++; cnt[p->dn[c] + 1]++; }
  for (int i = 0; i < n; i++) cnt[i + 1] += cnt[i];
  int *tmp = (int *)malloc(cnt[n] * sizeof(int)), *fill = (int *)malloc(n * sizeof(int));
  for (int i = 0; i < n; i++) { fill[i] = cnt[i]; tmp[fill[i]++] = i; }
  for (int c = 0; c < nc; c++) { tmp[fill[p->up[c]]++] = p->dn[c]; tmp[fill[p->dn[c]]++] = p->up[c]; }
  p->rowptr = (int *)malloc((n + 1) * sizeof(int));
  p->col = (int *)malloc(cnt[n] * sizeof(int));
  int nnz = 0;
  for (int i = 0; i < n; i++) {
    p->rowptr[i] = nnz;
    qsort(tmp + cnt[i], cnt[i + 1] - cnt[i], sizeof(int), cmp_int2);
    for (int k = cnt[i]; k < cnt[i + 1]; k++) if (k == cnt[i] || tmp[k] != tmp[k - 1]) p->col[nnz++] = tmp[k];
  }
  p->rowptr[n] = nnz;
  p->val = (double *)calloc(nnz, sizeof(double));
  p->diag = (int *)malloc(n * sizeof(int));
  for (int i = 0; i < n; i++) p->diag[i] = tfind(p, i, i);
  free(cnt); free(tmp); free(fill);
  return 0;
}

int to_add_condition(to_problem *p, int kind, int cond_type, int n, const int *id_dn, const double *du, const double *dd, const double *area) {
  tcond **list = kind == TO_COND_BC ? &p->bc : &p->ss;
  int *cnt = kind == TO_COND_BC ? &p->nbc : &p->nss;
  *list = (tcond *)realloc(*list, (*cnt + 1) * sizeof(tcond));
  tcond *c = &(*list)[*cnt];
  memset(c, 0, sizeof *c);
  c->kind = kind; c->cond_type = cond_type; c->n = n;
  c->id_dn = (int *)malloc((n + 1) * sizeof(int));
  c->dist_up = (double *)calloc(n + 1, sizeof(double)); c->dist_dn = (double *)calloc(n + 1, sizeof(double));
  c->area = (double *)calloc(n + 1, sizeof(double)); c->value = (double *)calloc(n + 1, sizeof(double));
  c->aux = (tauxvar *)malloc((n + 1) * sizeof(tauxvar));
  for (int i = 0; i < n; i++) {
    if (id_dn[i] < 1 || id_dn[i] > p->n) return -1;
    c->id_dn[i] = id_dn[i] - 1;
    taux_init(&c->aux[i]);
    /* a boundary auxvar keeps the is_soil_shallow = FALSE of ThermKSPTempSoilAuxVarInit (auxvar :65): MPPThermalSetSoils copies
       itype, por and the soil constants to it but not that flag (MultiPhysicsProbThermal.F90:186-204), so its conductivity is
       thk_bedrock for soil and wetland cells -- visible only when the boundary connection has dist_up > 0 */
    c->aux[i].is_soil_shallow = 0;
    if (kind == TO_COND_BC) { c->aux[i].is_active = 0; c->aux[i].frac = 0.0; }   /* ThermKSPTempBaseAuxVarInit (base auxvar :56-60) */
    if (kind == TO_COND_BC) { c->dist_up[i] = du[i]; c->dist_dn[i] = dd[i]; c->area[i] = area[i]; }
  }
  return ++(*cnt);
}

/* MPPThermalSetSoils (src/mpp/mpp/MultiPhysicsProbThermal.F90:76-208), per-cell arrays; BC auxvars copy their id_dn cell */
int to_set_soils(to_problem *p, const int *itype, const int *is_soil_shallow, const double *por, const double *tkmg,
                 const double *tkdry, const double *csol) {
  for (int i = 0; i < p->n; i++) {
    tauxvar *a = &p->aux[i];
    a->itype = itype[i]; a->is_soil_shallow = is_soil_shallow[i]; a->por = por[i];
    a->therm_cond_minerals = tkmg[i]; a->therm_cond_dry = tkdry[i]; a->heat_cap_minerals_puv = csol[i];
  }
  for (int b = 0; b < p->nbc; b++)
    for (int i = 0; i < p->bc[b].n; i++) {
      tauxvar *a = &p->bc[b].aux[i];
      const tauxvar *s = &p->aux[p->bc[b].id_dn[i]];
      a->itype = s->itype; a->por = s->por; a->therm_cond_minerals = s->therm_cond_minerals;
      a->therm_cond_dry = s->therm_cond_dry; a->heat_cap_minerals_puv = s->heat_cap_minerals_puv;
    }
  return 0;
}

/* SetRDataFromCLM(AUXVAR_INTERNAL, var, ...): 1 liq_areal_den, 2 ice_areal_den, 3 tuning_factor, 4 snow_water */
int to_set_cell_data(to_problem *p, int var, const double *v) {
  for (int i = 0; i < p->n; i++) {
    tauxvar *a = &p->aux[i];
    switch (var) {
      case 1: a->liq_areal_den = v[i]; break;
      case 2: a->ice_areal_den = v[i]; break;
      case 3: a->tuning_factor = v[i]; break;
      case 4: a->snow_water = v[i]; break;
      case 5: a->num_snow_layer = (int)v[i]; break;
      default: return -1;
    }
  }
  return 0;
}

int to_set_condition_values(to_problem *p, int kind, int id, int n, const double *v) {
  tcond *c = kind == TO_COND_BC ? &p->bc[id - 1] : &p->ss[id - 1];
  if (n > c->n) return -1;
  for (int i = 0; i < n; i++) { if (kind == TO_COND_BC) c->aux[i].condition_value = v[i]; c->value[i] = v[i]; }
  return 0;
}

/* sysofeqns_thermal_auxvar_type of a boundary condition: var 1 frac, 2 dhsdT, 3 is_active (SystemOfEquationsThermalAuxType.F90:23-32) */
int to_set_condition_aux(to_problem *p, int id, int var, int n, const double *v) {
  if (id < 1 || id > p->nbc) return -1;
  tcond *c = &p->bc[id - 1];
  if (n > c->n) return -1;
  for (int i = 0; i < n; i++) {
    if (var == 1) c->aux[i].frac = v[i];
    else if (var == 2) c->aux[i].dhsdT = v[i];
    else if (var == 3) c->aux[i].is_active = v[i] != 0.0;
    else return -1;
  }
  return 0;
}

int to_set_options(to_problem *p, double cnfac, int ksp_type, int pc_type, double ksp_rtol) {
  if (cnfac >= 0) p->cnfac = cnfac;
  p->ksp_type = ksp_type; p->pc_type = pc_type;
  if (ksp_rtol > 0) p->ksp_rtol = ksp_rtol;
  return 0;
}

int to_set_reference_defects(to_problem *p, int on) { p->reference_defects = on != 0; return 0; }
int to_set_soln_prev_clm(to_problem *p, const double *T) { memcpy(p->soln_prev_clm, T, p->n * sizeof(double)); return 0; }
/* ThermalSOEPreStepDT :391-409 */
int to_pre_step_dt(to_problem *p) {
  memcpy(p->soln_prev, p->soln_prev_clm, p->n * sizeof(double));
  memcpy(p->soln, p->soln_prev_clm, p->n * sizeof(double));
  return 0;
}

/* UpdateAuxVarsIntrn + UpdateAuxVarsBC (GE :566-642) */
static void update_auxvars(to_problem *p) {
  for (int i = 0; i < p->n; i++) { p->aux[i].temperature = p->soln_prev[i]; taux_compute(&p->aux[i], p->dz[i]); }
  for (int b = 0; b < p->nbc; b++) {
    tcond *c = &p->bc[b];
    for (int i = 0; i < c->n; i++) {
      int cell = c->id_dn[i];
      if (c->cond_type == TO_COND_DIRICHLET) c->aux[i].temperature = c->aux[i].condition_value;
      else c->aux[i].temperature = p->aux[cell].temperature;
      taux_compute(&c->aux[i], p->dz[cell]);
    }
  }
}

static double harmonic(double ku, double kd, double du, double dd) { return ku * kd * (du + dd) / (ku * dd + kd * du); }

/* ThermKSPTempSoilComputeRHS = Accum + Divergence (GE :646-972) */
int to_compute_rhs(to_problem *p, double *b) {
  double dt = p->dtime, cnfac = p->cnfac, stale_area = 0.0;
  update_auxvars(p);
  memset(b, 0, p->n * sizeof(double));
  for (int i = 0; i < p->n; i++) {
    const tauxvar *a = &p->aux[i];
    if (a->is_active) b[i] = a->heat_cap_pva * p->vol[i] / (dt * a->tuning_factor) * a->temperature;
  }
  for (int c = 0; c < p->nc; c++) {
    int u = p->up[c], d = p->dn[c];
    if (!p->aux[u].is_active || !p->aux[d].is_active) continue;
    double k = harmonic(p->aux[u].therm_cond, p->aux[d].therm_cond, p->dist_up[c], p->dist_dn[c]);
    double flux = -k * (p->aux[u].temperature - p->aux[d].temperature) / (p->dist_up[c] + p->dist_dn[c]); /* DiffHeatFlux :975-1003 */
    b[u] = b[u] + cnfac * flux * p->area[c] * 1.0;
    b[d] = b[d] - cnfac * flux * p->area[c] * 1.0;
    stale_area = p->area[c]; /* the local `area` of the reference routine (:787) */
  }
  for (int bb = 0; bb < p->nbc; bb++) {
    tcond *cd = &p->bc[bb];
    for (int i = 0; i < cd->n; i++) {
      int cell = cd->id_dn[i];
      if (!p->aux[cell].is_active) continue;
      if (cd->cond_type == TO_COND_DIRICHLET) {
        if (!cd->aux[i].is_active) continue;
        double du = cd->dist_up[i], dd = cd->dist_dn[i], dist = du + dd;
        double k = cd->aux[i].therm_cond * p->aux[cell].therm_cond * (du + dd) / (cd->aux[i].therm_cond * dd + p->aux[cell].therm_cond * du);
        b[cell] = b[cell] + k / dist * cd->aux[i].temperature * (p->reference_defects ? stale_area : cd->area[i]) * 1.0;
      } else if (cd->cond_type == TO_COND_HEAT_FLUX) {
        stale_area = cd->area[i]; /* :911 */
        /* cur_cond%value = H - dH/dT * T of the cell (ThermKSPTempSoilGetFromSOEAuxVarsBC :339-354; T = soln_prev, :157-187) */
        double value = cd->aux[i].condition_value - cd->aux[i].dhsdT * p->aux[cell].temperature;
        b[cell] = b[cell] + 1.0 * value * cd->aux[i].frac * cd->area[i];
      }
    }
  }
  for (int s = 0; s < p->nss; s++) {
    tcond *cd = &p->ss[s];
    for (int i = 0; i < cd->n; i++) {
      int cell = cd->id_dn[i];
      if (!p->aux[cell].is_active) continue;
      b[cell] = b[cell] + cd->value[i] * 1.0; /* COND_HEAT_RATE :950-962 */
    }
  }
  return 0;
}

/* ThermKSPTempSoilComputeOperatorsDiag (GE :1007-1229); auxvars as left by ComputeRHS */
int to_compute_operators(to_problem *p) {
  double dt = p->dtime, cnfac = p->cnfac;
  memset(p->val, 0, p->rowptr[p->n] * sizeof(double));
  for (int i = 0; i < p->n; i++) {
    const tauxvar *a = &p->aux[i];
    double value = a->is_active ? a->heat_cap_pva * p->vol[i] / (dt * a->tuning_factor) : 1.0;
    p->val[p->diag[i]] += value;
  }
  for (int c = 0; c < p->nc; c++) {
    int u = p->up[c], d = p->dn[c];
    if (!p->aux[u].is_active || !p->aux[d].is_active) continue;
    double du = p->dist_up[c], dd = p->dist_dn[c], dist = du + dd;
    double k = p->aux[u].therm_cond * p->aux[d].therm_cond * dist / (p->aux[u].therm_cond * dd + p->aux[d].therm_cond * du);
    double value = (1.0 - cnfac) * k / dist * p->area[c];
    p->val[p->diag[u]] += value * 1.0;
    p->val[tfind(p, u, d)] += -value * 1.0;
    p->val[tfind(p, d, u)] += -value * 1.0;
    p->val[p->diag[d]] += value * 1.0;
  }
  for (int bb = 0; bb < p->nbc; bb++) {
    tcond *cd = &p->bc[bb];
    for (int i = 0; i < cd->n; i++) {
      int cell = cd->id_dn[i];
      if (!p->aux[cell].is_active) continue;
      if (cd->cond_type == TO_COND_DIRICHLET) {
        if (!cd->aux[i].is_active) continue;
        double du = cd->dist_up[i], dd = cd->dist_dn[i], dist = du + dd;
        double k = cd->aux[i].therm_cond * p->aux[cell].therm_cond * dist / (cd->aux[i].therm_cond * dd + p->aux[cell].therm_cond * du);
        double value = cd->aux[i].frac * (1.0 - cnfac) * k / dist * cd->area[i] * 1.0;
        p->val[p->diag[cell]] += value;
      } else if (cd->cond_type == TO_COND_HEAT_FLUX) {
        if (p->reference_defects) p->val[p->diag[cell]] += -cd->aux[i].frac * pow(cd->aux[i].dhsdT, cd->area[i]) * 1.0; /* :1215 as written */
        else p->val[p->diag[cell]] += -cd->aux[i].frac * cd->aux[i].dhsdT * cd->area[i] * 1.0;
      }
    }
  }
  return 0;
}

int to_get_operator(to_problem *p, int *rowptr, int *col, double *val) {
  int nnz = p->rowptr[p->n];
  memcpy(rowptr, p->rowptr, (p->n + 1) * sizeof(int));
  memcpy(col, p->col, nnz * sizeof(int));
  memcpy(val, p->val, nnz * sizeof(double));
  return 0;
}
int to_nnz(to_problem *p) { return p->rowptr ? p->rowptr[p->n] : -1; }

/* SOEBaseStepDT_KSP :555-647 */
int to_step_dt(to_problem *p, double dt, int *converged) {
  p->dtime = dt;
  to_compute_rhs(p, p->rhs);
  to_compute_operators(p);
  int its = 0, reason = 0;
  vo_ksp_solve(p->lin, p->ksp_type, p->pc_type, p->ksp_rtol, p->rowptr, p->col, p->val, p->rhs, p->soln, &its, &reason);
  *converged = reason >= 0;
  p->cum_linear += its;
  memcpy(p->soln_prev, p->soln, p->n * sizeof(double)); /* post-solve */
  return 0;
}

int to_get_soln(to_problem *p, double *T) { memcpy(T, p->soln, p->n * sizeof(double)); return 0; }
int to_get_rhs(to_problem *p, double *b) { memcpy(b, p->rhs, p->n * sizeof(double)); return 0; }
long to_linear_iterations(to_problem *p) { return p->cum_linear; }
int to_set_dtime(to_problem *p, double dt) { p->dtime = dt; return 0; }
