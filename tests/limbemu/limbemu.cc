// limbemu.cc -- TEST INFRASTRUCTURE: a CPU lock-step emulator of the limb kernel (mujoco_mpc_amd/csrc/limb_step.h).
// The kernel's step function is a SIMT program for four lanes per candidate whose only cross-lane operations are the quad primitives
// qd_sum / qd_bcast / qd_or and the fence ld_sync around the quad's shared block. Here each lane is a thread and a primitive is an exchange
// through a shared buffer behind a barrier, so the SAME source that hipcc compiles for gfx950 runs on the host -- in double, against the
// oracle at 1e-9, and in float, the precision BASELINE configs[3] is quoted in (tests/test_limb_emulator.py). Never part of the product.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
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
  volatile float fbuf[2][4];
  volatile int ibuf[2][4];
};
static thread_local QuadCtx* g_ctx = nullptr;
static thread_local int g_lane = 0, g_phase = 0, g_sense = 0;

static inline double qd_sum(double x) {
  const int s = g_phase++ & 1, l = g_lane;
  g_ctx->dbuf[s][l] = x;
  g_ctx->bar.wait(g_sense);
  return (g_ctx->dbuf[s][l] + g_ctx->dbuf[s][l ^ 1]) + (g_ctx->dbuf[s][l ^ 2] + g_ctx->dbuf[s][l ^ 3]);  // the DPP butterfly's order
}
static inline float qd_sum(float x) {
  const int s = g_phase++ & 1, l = g_lane;
  g_ctx->fbuf[s][l] = x;
  g_ctx->bar.wait(g_sense);
  return (g_ctx->fbuf[s][l] + g_ctx->fbuf[s][l ^ 1]) + (g_ctx->fbuf[s][l ^ 2] + g_ctx->fbuf[s][l ^ 3]);
}
template <int K> static inline int qd_bcasti(int x) {
  const int s = g_phase++ & 1;
  g_ctx->ibuf[s][g_lane] = x;
  g_ctx->bar.wait(g_sense);
  return g_ctx->ibuf[s][K];
}
static inline int qd_or(int x) {
  const int s = g_phase++ & 1;
  g_ctx->ibuf[s][g_lane] = x;
  g_ctx->bar.wait(g_sense);
  return g_ctx->ibuf[s][0] | g_ctx->ibuf[s][1] | g_ctx->ibuf[s][2] | g_ctx->ibuf[s][3];
}
static inline bool qw_any(bool pred) { return qd_or(pred ? 1 : 0) != 0; }  // (the emulator's wavefront is one quad)
static inline void ld_sync() { (void)qd_or(0); }

#define LD static inline
#define LPOISON(x) std::memset((void*)&(x), 0xFF, sizeof(x))
#include "../../mujoco_mpc_amd/csrc/limb_model.h"
namespace mjpcx { namespace limb { template <typename R> struct LContact; template <typename R> struct LCross; } }
// the lane's stores: plain arrays here (LDS on the device, limb_kernel.h)
template <typename R> struct EmuCS { mjpcx::limb::LContact<R>* p; };
template <typename R> struct EmuMData { R l[21], b[mjpcx::limb::kLD][mjpcx::limb::kTD], t[45]; };
template <typename R> struct EmuM { EmuMData<R>* p; };  // (the stores are handles passed by value, as the device's LDS pointers are)
namespace mjpcx { namespace limb { template <typename R> struct LKin; } }
template <typename R> struct EmuKin { R (*cdof)[6]; R (*cdofT)[6]; };  // a view of an LKin held by the caller (the device keeps the axes in LDS)
template <typename R> static inline void lkin_store(EmuKin<R>& ks, const mjpcx::limb::LKin<R>& k);
template <typename R> struct EmuShared;
template <typename R> struct EmuSH { EmuShared<R>* p; };
template <typename R> static inline void lcs_load(const EmuCS<R>& cs, int i, mjpcx::limb::LContact<R>& c);
template <typename R> static inline void lcs_store(EmuCS<R>& cs, int i, const mjpcx::limb::LContact<R>& c);
template <typename R> static inline void lcs_store_jar(EmuCS<R>& cs, int i, const mjpcx::limb::LContact<R>& c);
template <typename R> static inline R lms_l(const EmuM<R>& m, int i) { return m.p->l[i]; }
template <typename R> static inline R lms_b(const EmuM<R>& m, int j, int k) { return m.p->b[j][k]; }
template <typename R> static inline R lms_t(const EmuM<R>& m, int i) { return m.p->t[i]; }
template <typename R> static inline void lms_set_l(EmuM<R>& m, int i, R v) { m.p->l[i] = v; }
template <typename R> static inline void lms_set_b(EmuM<R>& m, int j, int k, R v) { m.p->b[j][k] = v; }
template <typename R> static inline void lms_set_t(EmuM<R>& m, int i, R v) { m.p->t[i] = v; }
template <typename R> static inline void lsh_set_geom(EmuSH<R>& sh, int g, const R* pos, const R* axis);
template <typename R> static inline void lsh_get_geom(const EmuSH<R>& sh, int g, R* pos, R* axis);
template <typename R> static inline void lsh_set_cross(EmuSH<R>& sh, int r, const mjpcx::limb::LCross<R>& c);
template <typename R> static inline void lsh_get_cross(const EmuSH<R>& sh, int r, mjpcx::limb::LCross<R>& c);
template <typename R> static inline R lsh_xget(const EmuSH<R>& sh, int r, int f);
template <typename R> static inline void lsh_xset(EmuSH<R>& sh, int r, int f, R v);
#include "../../mujoco_mpc_amd/csrc/limb_step.h"
template <typename R> struct EmuShared { R gpos[mjpcx::limb::kNG][3], gax[mjpcx::limb::kNG][3]; mjpcx::limb::LCross<R> cross[mjpcx::limb::kMaxX]; R xrow[mjpcx::limb::kMaxX][2]; };
template <typename R> static inline R lsh_xget(const EmuSH<R>& sh, int r, int f) { return f == 6 ? sh.p->cross[r].D : sh.p->xrow[r][f - 10]; }
template <typename R> static inline void lsh_xset(EmuSH<R>& sh, int r, int f, R v) { sh.p->xrow[r][f - 10] = v; }
template <typename R> static inline void lkin_store(EmuKin<R>& ks, const mjpcx::limb::LKin<R>& k) {
  for (int j = 0; j < mjpcx::limb::kLD; j++) for (int c = 0; c < 6; c++) ks.cdof[j][c] = k.cdof[j][c];
  for (int j = 0; j < mjpcx::limb::kTD; j++) for (int c = 0; c < 6; c++) ks.cdofT[j][c] = k.cdofT[j][c];
}
template <typename R> static inline void lcs_load(const EmuCS<R>& cs, int i, mjpcx::limb::LContact<R>& c) { c = cs.p[i]; }
template <typename R> static inline void lcs_store(EmuCS<R>& cs, int i, const mjpcx::limb::LContact<R>& c) { cs.p[i] = c; }
template <typename R> static inline void lcs_store_jar(EmuCS<R>& cs, int i, const mjpcx::limb::LContact<R>& c) { for (int k = 0; k < 4; k++) cs.p[i].jar[k] = c.jar[k]; }
template <typename R> static inline void lsh_set_geom(EmuSH<R>& sh, int g, const R* pos, const R* axis) { for (int k = 0; k < 3; k++) { sh.p->gpos[g][k] = pos[k]; sh.p->gax[g][k] = axis[k]; } }
template <typename R> static inline void lsh_get_geom(const EmuSH<R>& sh, int g, R* pos, R* axis) { for (int k = 0; k < 3; k++) { pos[k] = sh.p->gpos[g][k]; axis[k] = sh.p->gax[g][k]; } }
template <typename R> static inline void lsh_set_cross(EmuSH<R>& sh, int r, const mjpcx::limb::LCross<R>& c) { sh.p->cross[r] = c; }
template <typename R> static inline void lsh_get_cross(const EmuSH<R>& sh, int r, mjpcx::limb::LCross<R>& c) { c = sh.p->cross[r]; }

using namespace mjpcx;
using namespace mjpcx::limb;

namespace {
template <typename R> struct Built {
  LimbModelT<R> lm;
  std::vector<R> weight, norm_p, norm_q, re, mocap, key;
  std::vector<int> ri;
  LTask<R> tk;
};
template <typename R> std::string build(const mjpcx_model* model, const mjpcx_task* task, const double* mocap, Built<R>& b) {
  LimbModelD* d = new LimbModelD;
  const std::string why = limb_build(model, task, d);
  if (!why.empty()) { delete d; return why; }
  limb_cast(*d, b.lm);
  delete d;
  b.weight.assign(task->weight, task->weight + task->num_term);
  b.norm_p.assign(task->num_term, R(0)); b.norm_q.assign(task->num_term, R(0));
  for (int k = 0, shift = 0; k < task->num_term; k++) {
    const int np = task->num_norm_parameter[k];
    if (np > 0) b.norm_p[k] = (R)task->norm_parameter[shift];
    if (np > 1) b.norm_q[k] = (R)task->norm_parameter[shift + 1];
    shift += np;
  }
  b.re.assign(task->residual_real, task->residual_real + task->num_residual_real);
  b.ri.assign(task->residual_int, task->residual_int + task->num_residual_int);
  if (mocap) b.mocap.assign(mocap, mocap + 7 * model->nmocap);
  const size_t nk = (size_t)model->nkey * model->nmocap * 3;
  b.key.resize(nk);
  for (size_t i = 0; i < nk; i++) b.key[i] = (R)model->key_mpos[i];
  b.tk.mocap = b.mocap.data(); b.tk.weight = b.weight.data(); b.tk.norm_p = b.norm_p.data(); b.tk.norm_q = b.norm_q.data();
  b.tk.re = b.re.data(); b.tk.ri = b.ri.data(); b.tk.risk = (R)task->risk; b.tk.key_mpos = b.key.data();
  return "";
}
template <class F> void run_quad(F&& body) {
  QuadCtx ctx;
  std::thread th[4];
  for (int l = 0; l < 4; l++) th[l] = std::thread([&, l] { g_ctx = &ctx; g_lane = l; g_phase = 0; g_sense = 0; body(l); });
  for (auto& t : th) t.join();
}

// one mj_forward + residual at (state, ctrl): out = qacc[nv] qfrc_smooth[nv] qfrc_constraint[nv] M[nv*nv] com[3] residual[nr] cost ncon nx iters
// qacc_smooth[nv]; returns the flag bits
template <typename R>
int forward_impl(const mjpcx_model* model, const mjpcx_task* task, const double* state, double time, const double* mocap, const double* ctrl_in,
                 const double* warm, double* out) {
  Built<R>* b = new Built<R>;
  if (!build(model, task, mocap, *b).empty()) { delete b; return -1; }
  const int nv = model->nv, nq = model->nq, nr = task->num_residual;
  int flags_out[4] = {0, 0, 0, 0};
  std::vector<double> M((size_t)nv * nv, 0.0);
  EmuShared<R> shared;
  std::memset(&shared, 0, sizeof shared);
  std::vector<R> res((size_t)nr, R(0));
  run_quad([&](int lane) {
    const LimbModelT<R>& m = b->lm;
    const LimbT<R>& L = m.limb[lane];
    LState<R> S;
    for (int k = 0; k < 7; k++) S.tq[k] = (R)state[k];
    for (int k = 0; k < 6; k++) { S.tv[k] = (R)state[nq + k]; S.wt[k] = warm ? (R)warm[k] : R(0); }
    R tctrl[3] = {0, 0, 0}, ctrl[kLD];
    for (int h = 0; h < 3; h++) {
      const LJointT<R>& J = m.tjnt[h];
      S.tq[7 + h] = J.on ? (R)state[J.qadr] : R(0); S.tv[6 + h] = J.on ? (R)state[nq + J.dof] : R(0); S.wt[6 + h] = (warm && J.on) ? (R)warm[J.dof] : R(0);
      if (J.on && J.act >= 0) tctrl[h] = (R)ctrl_in[J.act];
    }
    for (int j = 0; j < kLD; j++) {
      const LJointT<R>& J = L.jnt[j];
      S.lq[j] = J.on ? (R)state[J.qadr] : R(0); S.lv[j] = J.on ? (R)state[nq + J.dof] : R(0); S.wl[j] = (warm && J.on) ? (R)warm[J.dof] : R(0);
      ctrl[j] = (J.on && J.act >= 0) ? (R)ctrl_in[J.act] : R(0);
    }
    S.time = (R)time;
    LContact<R> con[kMaxPC];
    EmuCS<R> cs{con};
    EmuMData<R> msd;
    EmuM<R> ms{&msd};
    EmuSH<R> sh{&shared};
    LKin<R> kind;
    EmuKin<R> ks{kind.cdof, kind.cdofT};
    LDyn<R> D;
    LSense<R> f;
    int fl = forward_smooth(m, lane, &S, ctrl, tctrl, cs, ms, sh, ks, &D, &f, (long long*)nullptr);
    flags_out[lane] = fl;
    if (fl) return;
    const R cost = residual_cost(m, &b->tk, lane, &S, ctrl, tctrl, &f, res.data(), (long long*)nullptr);
    LNewtonOut<R> io;
    fl = newton(m, lane, ks, ms, &D, &S, cs, sh, warm != nullptr, &io, (long long*)nullptr);
    flags_out[lane] = fl;
    if (fl) return;
    const R* al = io.al; const R* at = io.at; const R* fc_l = io.fc_l; const R* fc_t = io.fc_t;
    const int iters = io.iters;
    for (int j = 0; j < kLD; j++) {
      const LJointT<R>& J = L.jnt[j];
      if (!J.on) continue;
      out[J.dof] = al[j]; out[nv + J.dof] = D.fs_l[j]; out[2 * nv + J.dof] = fc_l[j];
      out[3 * nv + (size_t)nv * nv + 3 + nr + 4 + J.dof] = D.sl[j];
      for (int i = 0; i < kLD; i++) if (L.jnt[i].on) M[(size_t)J.dof * nv + L.jnt[i].dof] = msd.l[tri(j, i)];
      for (int k = 0; k < kTD; k++) {
        const int dk = k < 6 ? k : (m.tjnt[k - 6].on ? m.tjnt[k - 6].dof : -1);
        if (dk >= 0) M[(size_t)J.dof * nv + dk] = M[(size_t)dk * nv + J.dof] = msd.b[j][k];
      }
    }
    if (lane == 0) {
      for (int k = 0; k < kTD; k++) {
        const int dk = k < 6 ? k : (m.tjnt[k - 6].on ? m.tjnt[k - 6].dof : -1);
        if (dk < 0) continue;
        out[dk] = at[k]; out[nv + dk] = D.fs_t[k]; out[2 * nv + dk] = fc_t[k];
        out[3 * nv + (size_t)nv * nv + 3 + nr + 4 + dk] = D.st[k];
        for (int i = 0; i < kTD; i++) { const int di = i < 6 ? i : (m.tjnt[i - 6].on ? m.tjnt[i - 6].dof : -1); if (di >= 0) M[(size_t)dk * nv + di] = msd.t[tri(k, i)]; }
      }
      double* o = out + 3 * nv + (size_t)nv * nv;
      for (int k = 0; k < 3; k++) o[k] = D.com[k];
      o[3 + nr] = cost; o[3 + nr + 1] = 0; o[3 + nr + 2] = D.nx; o[3 + nr + 3] = iters;
    }
    // (floor contacts of the candidate: summed by the caller from the lanes)
    g_ctx->ibuf[0][lane] = D.ncon;
  });
  std::memcpy(out + 3 * nv, M.data(), sizeof(double) * nv * nv);
  for (int i = 0; i < nr; i++) out[3 * nv + (size_t)nv * nv + 3 + i] = res[i];
  const int fl = flags_out[0] | flags_out[1] | flags_out[2] | flags_out[3];
  delete b;
  return fl;
}

template <typename R>
int rollout_impl(const mjpcx_model* model, const mjpcx_task* task, const double* state, double time, const double* mocap, int N, int H, int P,
                 int interp, const double* node_times, const double* node_values, const mjpcx_noise_spec* ns, const double* nominal,
                 double* states, double* actions, double* times, double* residual, double* costs, double* trace, double* total_return,
                 int* failure, double* nodes_out, int* flags, int* iters) {
  Built<R>* b = new Built<R>;
  if (!build(model, task, mocap, *b).empty()) { delete b; return -1; }
  const int nu = model->nu, ds = model->nq + model->nv, nr = task->num_residual, ntr = task->num_trace;
  std::vector<R> nodes((size_t)P * nu * N, R(0)), nt(P), nom((size_t)P * nu, R(0)), st0(ds);
  if (node_values)
    for (int c = 0; c < N; c++) for (int j = 0; j < P * nu; j++) nodes[(size_t)j * N + c] = (R)node_values[(size_t)c * P * nu + j];
  for (int p = 0; p < P; p++) nt[p] = (R)node_times[p];
  if (nominal) for (int j = 0; j < P * nu; j++) nom[j] = (R)nominal[j];
  for (int i = 0; i < ds; i++) st0[i] = (R)state[i];
  std::vector<R> o_states((size_t)N * H * ds), o_actions((size_t)N * H * nu), o_times((size_t)N * H), o_res((size_t)N * H * nr), o_costs((size_t)N * H),
      o_trace((size_t)N * H * 3 * (ntr > 0 ? ntr : 1));
  LArgs<R> a{};
  a.N = N; a.H = H; a.P = P; a.interp = interp; a.node_times = nt.data(); a.nodes = nodes.data(); a.nominal = nom.data();
  a.noise_mode = -1;
  if (!node_values && ns) {
    a.noise_mode = ns->mode; a.seed = ns->seed; a.iteration = ns->iteration; a.candidate_offset = ns->candidate_offset;
    a.nominal_candidate = ns->nominal_candidate; a.explore_count = ns->explore_count; a.std0 = ns->std0; a.std1 = ns->std1; a.param_variance = ns->param_variance;
  }
  a.states = o_states.data(); a.actions = o_actions.data(); a.times = o_times.data(); a.residual = o_res.data(); a.costs = o_costs.data(); a.trace = o_trace.data();
  a.total_return = total_return; a.failure = failure; a.iters = iters;
  std::atomic<int> next{0};
  const unsigned hw = std::thread::hardware_concurrency();
  const int nquads = (int)std::max(1u, std::min(hw / 4, (unsigned)N));
  std::vector<std::thread> pool;
  for (int q = 0; q < nquads; q++)
    pool.emplace_back([&] {
      for (;;) {
        const int cand = next.fetch_add(1);
        if (cand >= N) break;
        EmuShared<R> shared;
        std::memset(&shared, 0, sizeof shared);
        run_quad([&](int lane) {
          LContact<R> con[kMaxPC];
          EmuCS<R> cs{con};
          EmuMData<R> msd;
          EmuM<R> ms{&msd};
          EmuSH<R> sh{&shared};
          LKin<R> kind;
          EmuKin<R> ks{kind.cdof, kind.cdofT};
          const int fl = rollout(b->lm, b->tk, st0.data(), (R)time, a, cand, lane, cs, ms, sh, ks);
          if (lane == 0 && flags) flags[cand] = fl;
        });
      }
    });
  for (auto& t : pool) t.join();
  for (size_t i = 0; i < o_states.size(); i++) states[i] = o_states[i];
  for (size_t i = 0; i < o_actions.size(); i++) actions[i] = o_actions[i];
  for (size_t i = 0; i < o_times.size(); i++) times[i] = o_times[i];
  for (size_t i = 0; i < o_res.size(); i++) residual[i] = o_res[i];
  for (size_t i = 0; i < o_costs.size(); i++) costs[i] = o_costs[i];
  if (ntr > 0) for (size_t i = 0; i < o_trace.size(); i++) trace[i] = o_trace[i];
  if (nodes_out)
    for (int c = 0; c < N; c++) for (int j = 0; j < P * nu; j++) nodes_out[(size_t)c * P * nu + j] = nodes[(size_t)j * N + c];
  delete b;
  return 0;
}
}  // namespace

extern "C" {
const char* limbemu_check(const mjpcx_model* model, const mjpcx_task* task) {
  static thread_local std::string msg;
  LimbModelD* d = new LimbModelD;
  msg = limb_build(model, task, d);
  delete d;
  return msg.c_str();
}
int limbemu_forward(const mjpcx_model* model, const mjpcx_task* task, const double* state, double time, const double* mocap, const double* ctrl,
                    const double* warm, int precision, double* out) {
  return precision == 32 ? forward_impl<float>(model, task, state, time, mocap, ctrl, warm, out) : forward_impl<double>(model, task, state, time, mocap, ctrl, warm, out);
}
int limbemu_rollout(const mjpcx_model* model, const mjpcx_task* task, const double* state, double time, const double* mocap, int N, int H, int P,
                    int interp, const double* node_times, const double* node_values, const mjpcx_noise_spec* ns, const double* nominal, int precision,
                    double* states, double* actions, double* times, double* residual, double* costs, double* trace, double* total_return,
                    int* failure, double* nodes_out, int* flags, int* iters) {
  if (precision == 32)
    return rollout_impl<float>(model, task, state, time, mocap, N, H, P, interp, node_times, node_values, ns, nominal, states, actions, times, residual, costs, trace,
                               total_return, failure, nodes_out, flags, iters);
  return rollout_impl<double>(model, task, state, time, mocap, N, H, P, interp, node_times, node_values, ns, nominal, states, actions, times, residual, costs, trace,
                              total_return, failure, nodes_out, flags, iters);
}
}  // extern "C"
