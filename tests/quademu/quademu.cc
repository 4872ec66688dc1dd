// quademu.cc -- TEST INFRASTRUCTURE: a CPU lock-step emulator of the quad kernel (mujoco_mpc_amd/csrc/quad_step.h).
// The kernel's step function is a SIMT program for four lanes per candidate whose only cross-lane operations are the quad
// primitives qd_sum / qd_bcast / qd_rot / qd_or. Here each lane is a thread and a primitive is an exchange through a shared
// buffer behind a barrier, so the SAME source that hipcc compiles for gfx950 runs on the host and can be checked against the
// oracle without a GPU (tests/test_quad_emulator.py). Never part of the product: libmjpcx.so does not contain this file.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

struct QuadBarrier {
  std::atomic<int> count{0};
  std::atomic<int> sense{0};
  void wait(int& local_sense) {
    local_sense ^= 1;
    if (count.fetch_add(1, std::memory_order_acq_rel) == 3) {
      count.store(0, std::memory_order_relaxed);
      sense.store(local_sense, std::memory_order_release);
    } else {
      int spins = 0;
      while (sense.load(std::memory_order_acquire) != local_sense) if (++spins > 2000) { std::this_thread::yield(); }
    }
  }
};
struct QuadCtx {
  QuadBarrier bar;
  volatile double dbuf[2][4];
  volatile int ibuf[2][4];
};
static thread_local QuadCtx* g_ctx = nullptr;
static thread_local int g_lane = 0, g_phase = 0, g_sense = 0;

static inline double qd_sum(double x) {
  const int s = g_phase++ & 1, l = g_lane;
  g_ctx->dbuf[s][l] = x;
  g_ctx->bar.wait(g_sense);
  // the DPP butterfly of the device: (x_l + x_{l^1}) + (x_{l^2} + x_{l^3}), bit-identical in the four lanes
  return (g_ctx->dbuf[s][l] + g_ctx->dbuf[s][l ^ 1]) + (g_ctx->dbuf[s][l ^ 2] + g_ctx->dbuf[s][l ^ 3]);
}
template <int K> static inline double qd_bcast(double x) {
  const int s = g_phase++ & 1;
  g_ctx->dbuf[s][g_lane] = x;
  g_ctx->bar.wait(g_sense);
  return g_ctx->dbuf[s][K];
}
template <int D> static inline double qd_rot(double x) {  // lane (l + D) % 4's value
  const int s = g_phase++ & 1;
  g_ctx->dbuf[s][g_lane] = x;
  g_ctx->bar.wait(g_sense);
  return g_ctx->dbuf[s][(g_lane + D) & 3];
}
static inline double qd_rotv(double x, int d) {  // lane (l + d) % 4's value, d at run time
  const int s = g_phase++ & 1;
  g_ctx->dbuf[s][g_lane] = x;
  g_ctx->bar.wait(g_sense);
  return g_ctx->dbuf[s][(g_lane + d) & 3];
}
static inline double qd_partner(double x, int p) {  // lane (l xor p)'s value
  const int s = g_phase++ & 1;
  g_ctx->dbuf[s][g_lane] = x;
  g_ctx->bar.wait(g_sense);
  return g_ctx->dbuf[s][g_lane ^ p];
}
static inline int qd_or(int x);
static inline int qw_max(int v);
static inline bool qw_any(bool pred) { return qd_or(pred ? 1 : 0) != 0; }  // (the emulator's wavefront is one quad)
static inline int qd_or(int x) {
  const int s = g_phase++ & 1;
  g_ctx->ibuf[s][g_lane] = x;
  g_ctx->bar.wait(g_sense);
  return g_ctx->ibuf[s][0] | g_ctx->ibuf[s][1] | g_ctx->ibuf[s][2] | g_ctx->ibuf[s][3];
}

static inline int qw_max(int v) {  // the largest value in the wavefront (= the quad here), for v in 0..4
  const int m = qd_or(1 << v);
  return 31 - __builtin_clz(m);
}

#define QD static inline
#include "../../mujoco_mpc_amd/csrc/quad_model.h"
// the lane's contact store: a plain array here (LDS slots + a scratch overflow on the device, quad_kernel.h)
namespace mjpcx { namespace quad { struct QContact; } }
struct EmuProf {};
struct EmuStore { mjpcx::quad::QContact* p; };
static inline void qcs_load(const EmuStore& cs, int slot, mjpcx::quad::QContact& c);
static inline void qcs_store(EmuStore& cs, int slot, const mjpcx::quad::QContact& c);
static inline void qcs_store_jar(EmuStore& cs, int slot, const mjpcx::quad::QContact& c);
// the store of M (LDS on the device; the trunk block once per quad there, per lane here)
struct EmuM { double l[6], b[3][6], t[21]; };
static inline double qms_l(const EmuM& m, int i) { return m.l[i]; }
static inline double qms_b(const EmuM& m, int j, int k) { return m.b[j][k]; }
static inline double qms_t(const EmuM& m, int i) { return m.t[i]; }
static inline void qms_set_l(EmuM& m, int i, double v) { m.l[i] = v; }
static inline void qms_set_b(EmuM& m, int j, int k, double v) { m.b[j][k] = v; }
static inline void qms_set_t(EmuM& m, int i, double v) { m.t[i] = v; }
#include "../../mujoco_mpc_amd/csrc/quad_step.h"
static inline void qcs_load(const EmuStore& cs, int slot, mjpcx::quad::QContact& c) { c = cs.p[slot]; }
static inline void qcs_store(EmuStore& cs, int slot, const mjpcx::quad::QContact& c) { cs.p[slot] = c; }
static inline void qcs_store_jar(EmuStore& cs, int slot, const mjpcx::quad::QContact& c) { for (int k = 0; k < 6; k++) cs.p[slot].jar[k] = c.jar[k]; }

using namespace mjpcx;
using namespace mjpcx::quad;

namespace {
struct Built {
  QuadModel qm;
  QuadTables qt;
  std::vector<double> weight, norm_p, norm_q, param, re;
  std::vector<int> ri;
  QTask tk;
  QStaticPose sp[kQStatic];
};
std::string build(const mjpcx_model* model, const mjpcx_task* task, const double* mocap, Built& b) {
  std::string why = quad_build(model, task, &b.qm, &b.qt);
  if (!why.empty()) return why;
  b.weight.assign(task->weight, task->weight + task->num_term);
  b.norm_p.assign(task->num_term, 0.0); b.norm_q.assign(task->num_term, 0.0);
  for (int k = 0, shift = 0; k < task->num_term; k++) {
    const int np = task->num_norm_parameter[k];
    if (np > 0) b.norm_p[k] = task->norm_parameter[shift];
    if (np > 1) b.norm_q[k] = task->norm_parameter[shift + 1];
    shift += np;
  }
  b.param.assign(task->parameters, task->parameters + task->num_parameter);
  b.re.assign(task->residual_real, task->residual_real + task->num_residual_real);
  b.ri.assign(task->residual_int, task->residual_int + task->num_residual_int);
  b.tk.mocap = mocap; b.tk.weight = b.weight.data(); b.tk.norm_p = b.norm_p.data(); b.tk.norm_q = b.norm_q.data();
  b.tk.param = b.param.data(); b.tk.re = b.re.data(); b.tk.ri = b.ri.data(); b.tk.risk = task->risk;
  return "";
}
template <class F> void run_quad(F&& body) {
  QuadCtx ctx;
  std::thread th[4];
  for (int l = 0; l < 4; l++) th[l] = std::thread([&, l] { g_ctx = &ctx; g_lane = l; g_phase = 0; g_sense = 0; body(l); });
  for (auto& t : th) t.join();
}
}  // namespace

extern "C" {

const char* quademu_check(const mjpcx_model* model, const mjpcx_task* task) {
  static thread_local std::string msg;
  Built* b = new Built;
  msg = quad_build(model, task, &b->qm, &b->qt);
  delete b;
  return msg.c_str();
}

// the elimination plan of a leg-contact graph (quad_step.h make_plan): masks[k] bit x (1..3) = leg k touches leg k ^ x. out, per lane
// (4 x 9 ints): nslots, x[3], eslot[4], cyclic -- the four lanes must agree
void quademu_plan(const int* masks, int* out) {
  run_quad([&](int leg) {
    const QPlan p = make_plan(masks[leg], leg);
    int* o = out + 9 * leg;
    o[0] = p.nslots;
    for (int s = 0; s < 3; s++) o[1 + s] = p.x[s];
    for (int k = 0; k < 4; k++) o[4 + k] = p.eslot[k];
    o[8] = p.cyclic ? 1 : 0;
  });
}

// prints the self-collision tables of the model (bring-up aid)
void quademu_dump_pairs(const mjpcx_model* model, const mjpcx_task* task) {
  Built* b = new Built;
  if (!quad_build(model, task, &b->qm, &b->qt).empty()) { delete b; return; }
  for (int l = 0; l < kQLegs; l++) {
    const QuadLeg& L = b->qm.leg[l];
    std::printf("leg %d: %d pair geoms:", l, L.npg);
    for (int i = 0; i < L.npg; i++) { const QuadGeom& g = L.geom[L.pg_slot[i]]; std::printf(" [link %d type %d r %.3f h %.3f]", g.link, g.type, g.size[0], g.size[1]); }
    std::printf("\n");
    for (int o = 0; o <= kQLegs; o++) {
      if (o == l) continue;
      int n = 0;
      std::printf("  vs %d:", o);
      for (int i = 0; i < kQPairGeom; i++) for (int j = 0; j < kQPairGeom; j++) if (b->qt.mm[l][i][o][j].collide) { n++; std::printf(" (%d,%d)", i, j); }
      std::printf("  = %d\n", n);
    }
  }
  std::printf("trunk pair geoms: %d; pair_margin %g\n", b->qm.ntpg, b->qm.pair_margin);
  delete b;
}

// one mj_forward + residual at (state, ctrl): out = qacc[18] qfrc_smooth[18] qfrc_constraint[18] M[18*18] com[3] residual[42] cost
// ncon iters ; returns the flag bits
int quademu_forward(const mjpcx_model* model, const mjpcx_task* task, const double* state, double time, const double* mocap, const double* ctrl,
                    const double* warm /* 18 or NULL */, double* out) {
  Built* b = new Built;
  if (!build(model, task, mocap, *b).empty()) { delete b; return -1; }
  int flags_out[4] = {0, 0, 0, 0};
  double M[18][18];
  std::memset(M, 0, sizeof M);
  run_quad([&](int leg) {
    if (leg == 0) for (int s = 0; s < b->qm.nstatic; s++) static_pose(b->qm, mocap, s, b->sp[s]);
    (void)qd_or(0);
    QState S;
    for (int k = 0; k < 7; k++) S.tq[k] = state[k];
    for (int k = 0; k < 6; k++) { S.tv[k] = state[19 + k]; S.wt[k] = warm ? warm[k] : 0; }
    for (int j = 0; j < 3; j++) { S.lq[j] = state[7 + 3 * leg + j]; S.lv[j] = state[25 + 3 * leg + j]; S.wl[j] = warm ? warm[6 + 3 * leg + j] : 0; }
    S.time = time;
    QContact con[kQMaxCon];
    EmuStore cs{con};
    EmuM ms;
    QDyn D;
    QSense f;
    const double c3[3] = {ctrl[3 * leg], ctrl[3 * leg + 1], ctrl[3 * leg + 2]};
    EmuProf pf;
    int fl = forward_smooth(b->qm, b->qt, b->sp, leg, S, c3, cs, ms, D, f, pf);
    flags_out[leg] = fl;
    if (fl) return;
    QResidual r;
    const double cost = residual_cost(b->qm, b->tk, b->sp, leg, S, f, r);
    double al[3], at[6], fc_l[3], fc_t[6];
    int iters;
    if (((D.pmask >> 1) & 1) + ((D.pmask >> 2) & 1) + ((D.pmask >> 3) & 1) >= kQGeneralFrom)
      fl = constraint_newton<true>(b->qm, D.kin, ms, D.R, cs, D.ncon, D.nrel, leg, D.pmask, D.mymask, D.have_rel != 0, D.sl, D.st, S.wl, S.wt, warm != nullptr, al, at, fc_l, fc_t, iters, pf);
    else
      fl = constraint_newton<false>(b->qm, D.kin, ms, D.R, cs, D.ncon, D.nrel, leg, D.pmask, D.mymask, D.have_rel != 0, D.sl, D.st, S.wl, S.wt, warm != nullptr, al, at, fc_l, fc_t, iters, pf);
    flags_out[leg] = fl;
    if (fl) return;
    for (int j = 0; j < 3; j++) {
      out[6 + 3 * leg + j] = al[j]; out[18 + 6 + 3 * leg + j] = D.fs_l[j]; out[36 + 6 + 3 * leg + j] = fc_l[j];
      for (int i = 0; i < 3; i++) M[6 + 3 * leg + j][6 + 3 * leg + i] = ms.l[tri(j, i)];
      for (int k = 0; k < 6; k++) M[6 + 3 * leg + j][k] = M[k][6 + 3 * leg + j] = ms.b[j][k];
    }
    double* res = out + 54 + 324 + 3;
    res[7 + b->qm.leg[leg].foot_index] = r.gait;
    for (int j = 0; j < 3; j++) { res[13 + 3 * leg + j] = r.effort[j]; res[25 + 3 * leg + j] = r.posture[j]; }
    if (leg == 0) {
      for (int k = 0; k < 6; k++) { out[k] = at[k]; out[18 + k] = D.fs_t[k]; out[36 + k] = fc_t[k]; for (int i = 0; i < 6; i++) M[k][i] = ms.t[tri(k, i)]; }
      for (int k = 0; k < 3; k++) out[54 + 324 + k] = f.com[k];
      for (int i = 0; i < 7; i++) res[i] = r.shared[i];
      res[11] = r.shared[7]; res[12] = r.shared[8];
      for (int i = 0; i < 5; i++) res[37 + i] = r.shared[9 + i];
      res[42] = cost; res[43] = D.ncon; res[44] = iters;
    }
  });
  std::memcpy(out + 54, M, sizeof M);
  const int fl = flags_out[0] | flags_out[1] | flags_out[2] | flags_out[3];
  delete b;
  return fl;
}

// N rollouts. node_values: candidate-major [N][P][nu] or NULL (then `ns` generates them); outputs candidate-major [N][H][field];
// nodes_out [N][P][nu] (the candidates rolled out); flags[N]
static int run_rollouts(const mjpcx_model* model, const mjpcx_task* task, const double* state, double time, const double* mocap, int N, int H, int P,
                        int interp, const double* node_times, const double* node_values, const mjpcx_noise_spec* ns, const double* nominal,
                        double* states, double* actions, double* times, double* residual, double* costs, double* trace, double* total_return,
                        int* failure, double* nodes_out, int* flags, const QFeedback* fb) {
  Built* b = new Built;
  if (!build(model, task, mocap, *b).empty()) { delete b; return -1; }
  const int nu = model->nu;
  std::vector<double> nodes((size_t)P * nu * N, 0.0);  // [P][nu][N]
  if (node_values)
    for (int c = 0; c < N; c++) for (int j = 0; j < P * nu; j++) nodes[(size_t)j * N + c] = node_values[(size_t)c * P * nu + j];
  QArgs a{};
  a.N = N; a.H = H; a.P = P; a.interp = interp; a.node_times = node_times; a.nodes = nodes.data(); a.nominal = nominal;
  a.noise_mode = -1;
  if (!node_values && ns) {
    a.noise_mode = ns->mode; a.seed = ns->seed; a.iteration = ns->iteration; a.candidate_offset = ns->candidate_offset;
    a.nominal_candidate = ns->nominal_candidate; a.explore_count = ns->explore_count; a.std0 = ns->std0; a.std1 = ns->std1; a.param_variance = ns->param_variance;
  }
  a.states = states; a.actions = actions; a.times = times; a.residual = residual; a.costs = costs; a.trace = trace; a.total_return = total_return; a.failure = failure;
  for (int s = 0; s < b->qm.nstatic; s++) static_pose(b->qm, mocap, s, b->sp[s]);
  std::atomic<int> next{0};
  const unsigned hw = std::thread::hardware_concurrency();
  const int nquads = (int)std::max(1u, std::min(hw / 4, (unsigned)N));
  std::vector<std::thread> pool;
  for (int q = 0; q < nquads; q++)
    pool.emplace_back([&] {
      for (;;) {
        const int cand = next.fetch_add(1);
        if (cand >= N) break;
        run_quad([&](int leg) {
          QContact con[kQMaxCon];
          EmuStore cs{con};
          EmuProf pf;
          EmuM ms;
          const int fl = fb ? rollout<true>(b->qm, b->qt, b->sp, b->tk, state, time, a, *fb, cand, leg, cs, ms, pf)
                            : rollout<false>(b->qm, b->qt, b->sp, b->tk, state, time, a, QFeedback{}, cand, leg, cs, ms, pf);
          if (leg == 0 && flags) flags[cand] = fl;
        });
      }
    });
  for (auto& t : pool) t.join();
  if (nodes_out)
    for (int c = 0; c < N; c++) for (int j = 0; j < P * nu; j++) nodes_out[(size_t)c * P * nu + j] = nodes[(size_t)j * N + c];
  delete b;
  return 0;
}

int quademu_rollout(const mjpcx_model* model, const mjpcx_task* task, const double* state, double time, const double* mocap, int N, int H, int P,
                    int interp, const double* node_times, const double* node_values, const mjpcx_noise_spec* ns, const double* nominal,
                    double* states, double* actions, double* times, double* residual, double* costs, double* trace, double* total_return,
                    int* failure, double* nodes_out, int* flags) {
  return run_rollouts(model, task, state, time, mocap, N, H, P, interp, node_times, node_values, ns, nominal, states, actions, times, residual, costs, trace,
                      total_return, failure, nodes_out, flags, nullptr);
}

// N rollouts under the iLQG feedback policy (mjpcx_rollout_feedback's arguments: mode 0 the index policy, 1 iLQGPolicy::Action)
int quademu_rollout_feedback(const mjpcx_model* model, const mjpcx_task* task, const double* state, double time, const double* mocap, int N, int H, int mode,
                             int representation, int use_state, int Tn, const double* fb_times, const double* fb_states, const double* fb_actions,
                             const double* fb_gains, const double* fb_improvement, const double* alpha,
                             double* states, double* actions, double* times, double* residual, double* costs, double* trace, double* total_return,
                             int* failure, int* flags) {
  QFeedback fb{fb_times, fb_states, fb_actions, fb_gains, fb_improvement, alpha, Tn, mode, representation, use_state};
  const double t0[1] = {0.0};
  return run_rollouts(model, task, state, time, mocap, N, H, 1, 0, t0, nullptr, nullptr, nullptr, states, actions, times, residual, costs, trace,
                      total_return, failure, nullptr, flags, &fb);
}

}  // extern "C"
