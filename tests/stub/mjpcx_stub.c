/* A CPU stand-in for libmjpcx.so -- TEST INFRASTRUCTURE ONLY, never shipped, never on a product path.
 *
 * tests/test_distributed_gloo.py preloads it (LD_PRELOAD) under the C++ planners of mujoco_mpc_amd/host so that their SHARDING logic --
 * contiguous candidate ranges, candidate_offset, the exchange callbacks, top-k merge, elite moments, the robust planner's perturbed second stage -- runs on a machine without a
 * device at world size 1 and 2. It is not physics: a "rollout" is a fixed function of the candidate's spline nodes and the state, and a
 * candidate's noise is a hash of (seed, GLOBAL candidate index, iteration, entry) -- exactly the property the sharding contract rests
 * on (results independent of the number of ranks), with none of the numbers of the real kernels. Entry points the sampling /
 * cross-entropy / robust planners do not call return MJPCX_EUNSUPPORTED. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mjpcx.h"

struct mjpcx_ctx {
  int nq, nv, nu, nr, ntrace, N, H, P;
  double* ctrlrange; /* 2 nu */
  double state[64];
  double time;
  double* nodes;   /* N x P x nu */
  double* returns; /* N */
};

static uint64_t mix(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
static double unit(uint64_t h) { return ((double)(h >> 11) + 0.5) / 9007199254740992.0; }
static double normal(uint64_t seed, uint32_t cand, uint32_t entry, uint32_t iter) {
  const uint64_t h = mix(mix(seed ^ ((uint64_t)cand << 32 | entry)) + iter);
  return sqrt(-2.0 * log(unit(h))) * cos(6.283185307179586 * unit(mix(h)));
}
static double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
static double score(const mjpcx_ctx* c, const double* v) {
  double s = 0;
  for (int j = 0; j < c->P * c->nu; j++) { const double t = 0.3 * sin(j + c->state[0]) + 0.01 * c->time; s += (v[j] - t) * (v[j] - t); }
  return s;
}
static int reserve(mjpcx_ctx* c, int N, int H, int P) {
  free(c->nodes); free(c->returns);
  c->N = N; c->H = H; c->P = P;
  c->nodes = (double*)calloc((size_t)N * P * c->nu, sizeof(double));
  c->returns = (double*)calloc((size_t)N, sizeof(double));
  return c->nodes && c->returns ? MJPCX_OK : MJPCX_ENOMEM;
}

int mjpcx_create(const mjpcx_model* m, const mjpcx_task* t, int device, int precision, mjpcx_ctx** out) {
  (void)device; (void)precision;
  mjpcx_ctx* c = (mjpcx_ctx*)calloc(1, sizeof *c);
  c->nq = m->nq; c->nv = m->nv; c->nu = m->nu; c->nr = t->num_residual; c->ntrace = t->num_trace;
  c->ctrlrange = (double*)malloc(sizeof(double) * 2 * m->nu);
  memcpy(c->ctrlrange, m->actuator_ctrlrange, sizeof(double) * 2 * m->nu);
  *out = c;
  return MJPCX_OK;
}
void mjpcx_destroy(mjpcx_ctx* c) { if (c) { free(c->ctrlrange); free(c->nodes); free(c->returns); free(c); } }
const char* mjpcx_create_error(void) { return ""; }
const char* mjpcx_error_string(int code) { (void)code; return "stub"; }
const char* mjpcx_last_error(const mjpcx_ctx* c) { (void)c; return ""; }
const char* mjpcx_kernel_name(const mjpcx_ctx* c) { (void)c; return "cpu stub (tests/stub/mjpcx_stub.c): not a rollout"; }
int mjpcx_set_state(mjpcx_ctx* c, const double* state, double time, const double* mocap, const double* userdata) {
  (void)mocap; (void)userdata;
  for (int i = 0; i < c->nq + c->nv && i < 64; i++) c->state[i] = state[i];
  c->time = time;
  return MJPCX_OK;
}
int mjpcx_set_task_params(mjpcx_ctx* c, const double* w, const double* np, const double* p, double risk) { (void)c; (void)w; (void)np; (void)p; (void)risk; return MJPCX_OK; }
int mjpcx_set_residual_state(mjpcx_ctx* c, const int32_t* ri, const double* rr) { (void)c; (void)ri; (void)rr; return MJPCX_OK; }
int mjpcx_rollout_splines(mjpcx_ctx* c, int N, int H, int P, int interp, const double* times, const double* values) {
  (void)interp; (void)times;
  int rc = reserve(c, N, H, P);
  if (rc) return rc;
  memcpy(c->nodes, values, sizeof(double) * (size_t)N * P * c->nu);
  for (int i = 0; i < N; i++) c->returns[i] = score(c, c->nodes + (size_t)i * P * c->nu);
  return MJPCX_OK;
}
/* the perturbation of a "noisy rollout": a function of (seed, GLOBAL rollout index) on top of the spline's score */
int mjpcx_rollout_splines_noisy(mjpcx_ctx* c, int N, int H, int P, int interp, const double* t, const double* v, double s, double r, uint64_t seed, int off) {
  (void)interp; (void)t; (void)r;
  int rc = reserve(c, N, H, P);
  if (rc) return rc;
  memcpy(c->nodes, v, sizeof(double) * (size_t)N * P * c->nu);
  for (int i = 0; i < N; i++) {
    const double z = normal(seed, (uint32_t)(off + i), 0x7fffu, 0);
    c->returns[i] = score(c, c->nodes + (size_t)i * P * c->nu) * (1.0 + s * z * z);
  }
  return MJPCX_OK;
}
int mjpcx_rollout_noise(mjpcx_ctx* c, int N, int H, int P, int interp, const double* times, const double* nominal, const mjpcx_noise_spec* ns) {
  (void)interp; (void)times;
  int rc = reserve(c, N, H, P);
  if (rc) return rc;
  for (int i = 0; i < N; i++) {
    const int gi = ns->candidate_offset + i;
    double* v = c->nodes + (size_t)i * P * c->nu;
    for (int j = 0; j < P * c->nu; j++) {
      const int k = j % c->nu;
      const double lo = c->ctrlrange[2 * k], hi = c->ctrlrange[2 * k + 1];
      double sigma;
      if (ns->mode == MJPCX_NOISE_SAMPLING) sigma = 0.5 * (hi - lo) * ns->std0;
      else { const double fl = gi < ns->explore_count ? ns->std0 : ns->std1, s = sqrt(ns->param_variance[j]); sigma = s > fl ? s : fl; }
      v[j] = nominal[j];
      if (gi != ns->nominal_candidate) v[j] = clampd(v[j] + sigma * normal(ns->seed, (uint32_t)gi, (uint32_t)j, ns->iteration), lo, hi);
    }
    c->returns[i] = score(c, v);
  }
  return MJPCX_OK;
}
int mjpcx_sync(mjpcx_ctx* c) { (void)c; return MJPCX_OK; }
int mjpcx_get_returns(mjpcx_ctx* c, double* ret, int32_t* fail) {
  for (int i = 0; i < c->N; i++) { if (ret) ret[i] = c->returns[i]; if (fail) fail[i] = 0; }
  return MJPCX_OK;
}
int mjpcx_get_return_at(mjpcx_ctx* c, int i, double* ret, int32_t* fail) { if (ret) *ret = c->returns[i]; if (fail) *fail = 0; return MJPCX_OK; }
int mjpcx_topk(mjpcx_ctx* c, int k, int32_t* index, double* ret) {
  char* used = (char*)calloc((size_t)c->N, 1);
  for (int r = 0; r < k; r++) {
    int best = -1;
    for (int i = 0; i < c->N; i++) if (!used[i] && (best < 0 || c->returns[i] < c->returns[best])) best = i;
    used[best] = 1; index[r] = best; ret[r] = c->returns[best];
  }
  free(used);
  return MJPCX_OK;
}
int mjpcx_best(mjpcx_ctx* c, int ref, int32_t* index, double* best_return, double* ref_return, double* spline) {
  int b = 0;
  for (int i = 1; i < c->N; i++) if (c->returns[i] < c->returns[b]) b = i;
  *index = b; *best_return = c->returns[b];
  if (ref >= 0 && ref_return) *ref_return = c->returns[ref];
  if (spline) memcpy(spline, c->nodes + (size_t)b * c->P * c->nu, sizeof(double) * c->P * c->nu);
  return MJPCX_OK;
}
int mjpcx_elite_moments(mjpcx_ctx* c, int n, const int32_t* cand, const double* mean, double* out, double* sum_return) {
  const int np = c->P * c->nu;
  for (int j = 0; j < np; j++) out[j] = 0;
  double sr = 0;
  for (int e = 0; e < n; e++) {
    const double* v = c->nodes + (size_t)cand[e] * np;
    for (int j = 0; j < np; j++) out[j] += mean ? (v[j] - mean[j]) * (v[j] - mean[j]) : v[j];
    sr += c->returns[cand[e]];
  }
  if (sum_return) *sum_return = sr;
  return MJPCX_OK;
}
int mjpcx_fetch_trajectory(mjpcx_ctx* c, int i, mjpcx_traj_view* o) {
  const int H = c->H < o->horizon ? c->H : o->horizon, ds = c->nq + c->nv;
  o->horizon = H;
  if (o->states) memset(o->states, 0, sizeof(double) * (size_t)H * ds);
  if (o->actions) memset(o->actions, 0, sizeof(double) * (size_t)H * c->nu);
  if (o->times) for (int t = 0; t < H; t++) o->times[t] = c->time + 0.01 * t;
  if (o->residual) memset(o->residual, 0, sizeof(double) * (size_t)H * c->nr);
  if (o->costs) memset(o->costs, 0, sizeof(double) * (size_t)H);
  if (o->trace) memset(o->trace, 0, sizeof(double) * (size_t)H * 3 * c->ntrace);
  o->total_return = c->returns[i]; o->failure = 0;
  return MJPCX_OK;
}
int mjpcx_fetch_spline(mjpcx_ctx* c, int i, double* v) { memcpy(v, c->nodes + (size_t)i * c->P * c->nu, sizeof(double) * c->P * c->nu); return MJPCX_OK; }
int mjpcx_timing_reset(mjpcx_ctx* c) { (void)c; return MJPCX_OK; }
int mjpcx_timing_read(mjpcx_ctx* c, double* ms, int64_t* n) { (void)c; if (ms) *ms = 0; if (n) *n = 0; return MJPCX_OK; }
int mjpcx_timing_read_main(mjpcx_ctx* c, double* ms, int64_t* n) { (void)c; if (ms) *ms = 0; if (n) *n = 0; return MJPCX_OK; }
int mjpcx_quad_stats(mjpcx_ctx* c, int32_t* h) { (void)c; for (int k = 0; k < 8; k++) h[k] = 0; return MJPCX_OK; }
int64_t mjpcx_algorithmic_bytes(const mjpcx_ctx* c, int H, int P) { (void)c; (void)H; (void)P; return 0; }
int mjpcx_comm_unique_id(void* id) { (void)id; return MJPCX_EUNSUPPORTED; }
int mjpcx_comm_init(mjpcx_ctx* c, const void* id, int rank, int world) { (void)c; (void)id; (void)rank; (void)world; return MJPCX_EUNSUPPORTED; }
int mjpcx_comm_info(const mjpcx_ctx* c, int* rank, int* world) { (void)c; if (rank) *rank = 0; if (world) *world = 1; return MJPCX_OK; }
int mjpcx_exchange_best(mjpcx_ctx* c, int32_t* i, double* b, double* n, double* s, int k) { (void)c; (void)i; (void)b; (void)n; (void)s; (void)k; return MJPCX_OK; }
int mjpcx_merge_topk(mjpcx_ctx* c, int k, int64_t* i, double* r) { (void)c; (void)k; (void)i; (void)r; return MJPCX_OK; }
int mjpcx_elite_allreduce(mjpcx_ctx* c, double* v, int n) { (void)c; (void)v; (void)n; return MJPCX_OK; }
int mjpcx_comm_barrier(mjpcx_ctx* c) { (void)c; return MJPCX_OK; }
int mjpcx_comm_destroy(mjpcx_ctx* c) { (void)c; return MJPCX_OK; }
/* entry points the sampling / cross-entropy planners never reach: present so that no call can fall through to the device library with a
 * stub context */
int mjpcx_rollout_feedback(mjpcx_ctx* c, int n, int h, int mode, int rep, int us, int nh, const double* t, const double* s, const double* a,
                           const double* g, const double* imp, const double* al) {
  (void)c; (void)n; (void)h; (void)mode; (void)rep; (void)us; (void)nh; (void)t; (void)s; (void)a; (void)g; (void)imp; (void)al; return MJPCX_EUNSUPPORTED;
}
int mjpcx_transition_fd(mjpcx_ctx* c, int nh, const double* t, const double* s, const double* a, double eps, int cen, double* A, double* B, double* C, double* D) {
  (void)c; (void)nh; (void)t; (void)s; (void)a; (void)eps; (void)cen; (void)A; (void)B; (void)C; (void)D; return MJPCX_EUNSUPPORTED;
}
int mjpcx_cost_derivatives(mjpcx_ctx* c, int T, const double* r, const double* C, const double* D, double* cx, double* cu, double* cxx, double* cxu, double* cuu) {
  (void)c; (void)T; (void)r; (void)C; (void)D; (void)cx; (void)cu; (void)cxx; (void)cxu; (void)cuu; return MJPCX_EUNSUPPORTED;
}
int mjpcx_backward_pass(mjpcx_ctx* c, int n, int m, int T, double mu, int rt, int ul, const double* A, const double* B, const double* cx, const double* cu,
                        const double* cxx, const double* cxu, const double* cuu, const double* act, const double* lim, double* Vx, double* Vxx, double* K,
                        double* du, double* dV, int32_t* status, double* ms) {
  (void)c; (void)n; (void)m; (void)T; (void)mu; (void)rt; (void)ul; (void)A; (void)B; (void)cx; (void)cu; (void)cxx; (void)cxu; (void)cuu; (void)act; (void)lim;
  (void)Vx; (void)Vxx; (void)K; (void)du; (void)dV; (void)status; (void)ms; return MJPCX_EUNSUPPORTED;
}
int mjpcx_kinematics(mjpcx_ctx* c, double* xpos, double* xquat, double* xmat, double* xipos, double* site_xpos, double* com, double* linvel) {
  (void)c; (void)xpos; (void)xquat; (void)xmat; (void)xipos; (void)site_xpos; (void)com; (void)linvel; return MJPCX_EUNSUPPORTED;
}
int mjpcx_device_buffer(mjpcx_ctx* c, int which, void** ptr, size_t* bytes) { (void)c; (void)which; (void)ptr; (void)bytes; return MJPCX_EUNSUPPORTED; }
