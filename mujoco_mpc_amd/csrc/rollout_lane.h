// rollout_lane.h -- the "small model" rollout-and-evaluate kernel for gfx950.
//
// Replaces, for one batch of N candidate splines, the reference's fan-out of
//   Trajectory::Rollout  (mjpc/trajectory.cc:92-210)     N x [(H-1) x mj_step + mj_forward]
//   Trajectory::UpdateReturn (mjpc/trajectory.cc:312-326) N x H x Task::CostValue
//   SamplingPolicy::Action (mjpc/planners/sampling/policy.cc:52-59) per step
//   AddNoiseToPolicy (sampling/planner.cc:326-352, cross_entropy/planner.cc:351-385)
// across a ThreadPool (sampling/planner.cc:355-393) by ONE kernel launch.
//
// Mapping to CDNA4
//   * one candidate per LANE, 64 candidates per wavefront, one wavefront per workgroup:
//     a 2-dof model has no intra-candidate parallelism worth a cross-lane exchange, so
//     all 64 lanes do useful fp64 work and there is no divergence between lanes except
//     at (rare) active joint limits, which are predicated.
//   * the kinematic-tree TOPOLOGY is a compile-time template parameter (Topo): every
//     loop over bodies/dofs is fully unrolled and every per-candidate quantity
//     (xpos, xmat, cinert, cdof, M, ...) lives in VGPRs -- no scratch, no LDS round trips.
//   * the model's numeric constants (LaneModel) are read through a wave-uniform pointer
//     with constant offsets: s_load through the scalar cache into SGPRs.
//   * the candidate's spline nodes and the node times are staged once into LDS
//     ([node][actuator][lane], conflict-free) and touched only when the (wave-uniform) spline
//     segment changes: there is NO global load inside the time loop, so the per-step stores are
//     never waited for.
//   * per-step outputs are written in a [step][field][candidate] SoA layout so that each
//     store instruction of the wave writes 64 consecutive elements (512 B for fp64).
#pragma once
#include <utility>

#include "device_common.h"

namespace mjpcx {

// ------------------------------------------------------------------ static topology
// Packed nibbles/bits so that the descriptor is a handful of integer template arguments
// the host can also compute from a runtime model to select an instantiation.
template <int NB_, int NV_, int NU_, int NSITE_, int NMOCAP_, uint64_t PARENT_, uint64_t MOCAP_,
          uint64_t JTYPE_, uint64_t JBODY_, uint64_t JLIMITED_, uint64_t ACTJ_, uint64_t SITEB_>
struct Topo {
  static constexpr int NB = NB_, NV = NV_, NU = NU_, NSITE = NSITE_, NMOCAP = NMOCAP_;
  static constexpr uint64_t kParent = PARENT_, kMocap = MOCAP_, kJtype = JTYPE_, kJbody = JBODY_,
                            kJlimited = JLIMITED_, kActj = ACTJ_, kSiteb = SITEB_;
  __host__ __device__ static constexpr int parent(int b) { return (int)((PARENT_ >> (4 * b)) & 15); }
  __host__ __device__ static constexpr int mocap(int b) { return (int)((MOCAP_ >> (4 * b)) & 15); }  // 15: none
  __host__ __device__ static constexpr int jtype(int j) { return (int)((JTYPE_ >> (2 * j)) & 3); }
  __host__ __device__ static constexpr int jbody(int j) { return (int)((JBODY_ >> (4 * j)) & 15); }
  __host__ __device__ static constexpr bool jlimited(int j) { return (JLIMITED_ >> j) & 1; }
  __host__ __device__ static constexpr int actj(int u) { return (int)((ACTJ_ >> (4 * u)) & 15); }
  __host__ __device__ static constexpr int siteb(int s) { return (int)((SITEB_ >> (4 * s)) & 15); }
  __host__ __device__ static constexpr int root(int b) {
    while (b > 0 && parent(b) != 0) b = parent(b);
    return b;
  }
  __host__ __device__ static constexpr bool has_joint(int b) {
    for (int j = 0; j < NV_; j++) if (jbody(j) == b) return true;
    return false;
  }
  // does any dof move body b?
  __host__ __device__ static constexpr bool moves(int b) {
    while (b > 0) { if (has_joint(b)) return true; b = parent(b); }
    return false;
  }
  __host__ __device__ static constexpr int last_dof_of_body(int b, int before) {
    int r = -1;
    for (int j = 0; j < before; j++) if (jbody(j) == b) r = j;
    return r;
  }
  __host__ __device__ static constexpr int dof_parent(int j) {
    int r = last_dof_of_body(jbody(j), j);
    if (r >= 0) return r;
    int b = parent(jbody(j));
    while (b > 0) {
      r = last_dof_of_body(b, NV_);
      if (r >= 0) return r;
      b = parent(b);
    }
    return -1;
  }
  __host__ __device__ static constexpr bool dof_ancestor(int i, int j) {  // is j a strict ancestor of i
    int k = dof_parent(i);
    while (k >= 0) { if (k == j) return true; k = dof_parent(k); }
    return false;
  }
  // does dof i move body b (its joint sits on b or on an ancestor of b)
  __host__ __device__ static constexpr bool dof_moves_body(int i, int b) {
    while (b > 0) { if (jbody(i) == b) return true; b = parent(b); }
    return false;
  }
  __host__ __device__ static constexpr int num_limited() {
    int n = 0;
    for (int j = 0; j < NV_; j++) n += jlimited(j) ? 1 : 0;
    return n;
  }
};

template <int RID_, int NR_, int NTERM_, uint64_t TERMDIM_, int NTRACE_, uint64_t TRACESITE_>
struct TaskTopo {
  static constexpr int RID = RID_, NR = NR_, NTERM = NTERM_, NTRACE = NTRACE_;
  static constexpr uint64_t kTermDim = TERMDIM_, kTraceSite = TRACESITE_;
  __host__ __device__ static constexpr int term_dim(int k) { return (int)((TERMDIM_ >> (4 * k)) & 15); }
  __host__ __device__ static constexpr int term_off(int k) {
    int o = 0;
    for (int i = 0; i < k; i++) o += term_dim(i);
    return o;
  }
  __host__ __device__ static constexpr int trace_site(int k) { return (int)((TRACESITE_ >> (4 * k)) & 15); }
};

struct NoiseArgs {
  int mode;  // -1: candidates given in `nodes`; else MJPCX_NOISE_*
  uint64_t seed;
  uint32_t iteration;
  int candidate_offset, nominal_candidate, explore_count;
  double std0, std1;
  const double* param_variance;  // device, P*NU (cross-entropy)
};

template <typename T>
struct RolloutArgs {
  int N, H, P, interp;
  const T* node_times;  // P
  T* nodes;             // [P][NU][N]
  const T* nominal;     // [P][NU]
  NoiseArgs noise;
  T *states, *actions, *times, *residual, *costs, *trace;
  double* total_return;
  int* failure;
  // Trajectory::NoisyRollout (wavefront-per-candidate kernels): Ornstein-Uhlenbeck xfrc_applied noise, decay = exp(-dt / rate),
  // scale = std sqrt(1 - decay^2); scale = 0: plain Rollout. Normals: Philox keyed on (seed, global candidate, step, entry).
  double xfrc_decay, xfrc_scale;
  uint64_t xfrc_seed;
  // generic Jacobian-free kernels, second pass: roll out only the candidates whose failure[] carries the list-overflow warning (bit 32 << 8)
  int only_overflowed;
};

// weighted sum of norms over the (compile-time) term partition of the residual
template <class TK, typename T, int... K>
__device__ __forceinline__ T cost_terms(const T (&r)[TK::NR], const LaneTask<T>& tk,
                                        std::integer_sequence<int, K...>) {
  return ((tk.weight[K] * norm_value<T, TK::term_dim(K)>(&r[TK::term_off(K)], tk.norm[K], tk.norm_p[K], tk.norm_q[K])) + ... + T(0));
}

// ------------------------------------------------------------------ the per-candidate step engine
// mj_forward (position, velocity, actuation, acceleration, constraint stages) for ONE candidate held in
// the calling lane's registers. Shared by the rollout kernels, the finite-difference derivative kernel
// and anything else that needs "one MuJoCo step" of a registered small-model topology.
//   out: qacc, qfrc (= qfrc_smooth), qfrc_c (= qfrc_constraint), M (lower triangle), site_xpos
template <class TP, typename T, class MODEL>
__device__ __forceinline__ void lane_forward(const MODEL& m, const LaneTask<T>& tk, const T (&qpos)[TP::NV],
                                             const T (&qvel)[TP::NV], const T (&ctrl)[TP::NU], T (&qacc)[TP::NV],
                                             T (&qfrc)[TP::NV], T (&qfrc_c)[TP::NV], T (&M)[TP::NV][TP::NV],
                                             T (&site_xpos)[TP::NSITE > 0 ? TP::NSITE : 1][3],
                                             const T (*xfrc)[6] = nullptr) {
  constexpr int NB = TP::NB, NV = TP::NV, NU = TP::NU, NS = TP::NSITE;
  const T h = m.timestep;
  // ================= position stage: kinematics
  T xpos[NB][3], xquat[NB][4], xmat[NB][9], xipos[NB][3], ximat[NB][9];
  T xanchor[NV][3], xaxis[NV][3];
#pragma unroll
  for (int b = 1; b < NB; b++) {
    T pos[3], quat[4];
    if (TP::mocap(b) != 15) {
#pragma unroll
      for (int c = 0; c < 3; c++) pos[c] = tk.mocap_pos[TP::mocap(b)][c];
#pragma unroll
      for (int c = 0; c < 4; c++) quat[c] = tk.mocap_quat[TP::mocap(b)][c];
    } else {
      if (TP::parent(b) == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) pos[c] = m.body_pos[b][c];
#pragma unroll
        for (int c = 0; c < 4; c++) quat[c] = m.body_quat[b][c];
      } else {
        const int p = TP::parent(b);
        T bp[3] = {m.body_pos[b][0], m.body_pos[b][1], m.body_pos[b][2]};
        T bq[4] = {m.body_quat[b][0], m.body_quat[b][1], m.body_quat[b][2], m.body_quat[b][3]};
        mat_vec(pos, xmat[p], bp);
#pragma unroll
        for (int c = 0; c < 3; c++) pos[c] += xpos[p][c];
        quat_mul(quat, xquat[p], bq);
      }
#pragma unroll
      for (int j = 0; j < NV; j++) {
        if (TP::jbody(j) == b) {
          T R[9];
          quat_to_mat(R, quat);
          T jp[3] = {m.jnt_pos[j][0], m.jnt_pos[j][1], m.jnt_pos[j][2]};
          T ja[3] = {m.jnt_axis[j][0], m.jnt_axis[j][1], m.jnt_axis[j][2]};
          mat_vec(xanchor[j], R, jp);
#pragma unroll
          for (int c = 0; c < 3; c++) xanchor[j][c] += pos[c];
          mat_vec(xaxis[j], R, ja);
          const T dq = qpos[j] - m.qpos0[j];
          if (TP::jtype(j) == kJntSlide) {
#pragma unroll
            for (int c = 0; c < 3; c++) pos[c] += xaxis[j][c] * dq;
          } else {  // hinge
            T sn, cs;
            sincos_t(T(0.5) * dq, sn, cs);
            T ql[4] = {cs, ja[0] * sn, ja[1] * sn, ja[2] * sn};
            quat_mul(quat, quat, ql);
            T R2[9], v[3];
            quat_to_mat(R2, quat);
            mat_vec(v, R2, jp);
#pragma unroll
            for (int c = 0; c < 3; c++) pos[c] = xanchor[j][c] - v[c];
          }
        }
      }
    }
    // mj_kinematics renormalises every body quaternion to stop drift of FREE/BALL qpos quaternions.
    // Here only mocap poses (user input) can be unnormalised: a slide/hinge body's quaternion is a
    // product of unit quaternions (model constants and axis-angle factors), unit to rounding, so
    // the sqrt + divide on the per-step dependent chain is skipped for those bodies.
    if (TP::mocap(b) != 15) normalize4(quat);
#pragma unroll
    for (int c = 0; c < 3; c++) xpos[b][c] = pos[c];
#pragma unroll
    for (int c = 0; c < 4; c++) xquat[b][c] = quat[c];
    quat_to_mat(xmat[b], quat);
    T ip[3] = {m.body_ipos[b][0], m.body_ipos[b][1], m.body_ipos[b][2]};
    T iq[4] = {m.body_iquat[b][0], m.body_iquat[b][1], m.body_iquat[b][2], m.body_iquat[b][3]};
    T v[3], q2[4];
    mat_vec(v, xmat[b], ip);
#pragma unroll
    for (int c = 0; c < 3; c++) xipos[b][c] = pos[c] + v[c];
    quat_mul(q2, quat, iq);
    quat_to_mat(ximat[b], q2);
  }
#pragma unroll
  for (int s = 0; s < NS; s++) {
    const int b = TP::siteb(s);
    T sp[3] = {m.site_pos[s][0], m.site_pos[s][1], m.site_pos[s][2]};
    if (b == 0) {
#pragma unroll
      for (int c = 0; c < 3; c++) site_xpos[s][c] = sp[c];
    } else {
      T v[3];
      mat_vec(v, xmat[b], sp);
#pragma unroll
      for (int c = 0; c < 3; c++) site_xpos[s][c] = xpos[b][c] + v[c];
    }
  }

  // ================= comPos: subtree com of each moving tree, cinert, cdof
  T com[NB][3];  // only entries of moving roots are used
#pragma unroll
  for (int r = 1; r < NB; r++) {
    if (TP::parent(r) == 0 && TP::moves(r)) {
      T acc[3] = {0, 0, 0};
#pragma unroll
      for (int b = 1; b < NB; b++)
        if (TP::root(b) == r) {
#pragma unroll
          for (int c = 0; c < 3; c++) acc[c] += m.body_mass[b] * xipos[b][c];
        }
#pragma unroll
      for (int c = 0; c < 3; c++) com[r][c] = acc[c] * m.root_invmass[r];
    }
  }
  T cinert[NB][10];
#pragma unroll
  for (int b = 1; b < NB; b++) {
    if (TP::moves(b)) {
      T off[3];
#pragma unroll
      for (int c = 0; c < 3; c++) off[c] = xipos[b][c] - com[TP::root(b)][c];
      T bi[3] = {m.body_inertia[b][0], m.body_inertia[b][1], m.body_inertia[b][2]};
      inert_com(cinert[b], bi, ximat[b], off, m.body_mass[b]);
    }
  }
  T cdof[NV][6];
#pragma unroll
  for (int j = 0; j < NV; j++) {
    if (TP::jtype(j) == kJntSlide) {
      cdof[j][0] = cdof[j][1] = cdof[j][2] = 0;
#pragma unroll
      for (int c = 0; c < 3; c++) cdof[j][3 + c] = xaxis[j][c];
    } else {
      T off[3], cr[3];
#pragma unroll
      for (int c = 0; c < 3; c++) off[c] = com[TP::root(TP::jbody(j))][c] - xanchor[j][c];
      cross3(cr, xaxis[j], off);
#pragma unroll
      for (int c = 0; c < 3; c++) { cdof[j][c] = xaxis[j][c]; cdof[j][3 + c] = cr[c]; }
    }
  }

  // ================= CRB -> M (lower triangle, static sparsity), LDL' factor
  T crb[NB][10];
#pragma unroll
  for (int b = 1; b < NB; b++)
    if (TP::moves(b)) {
#pragma unroll
      for (int c = 0; c < 10; c++) crb[b][c] = cinert[b][c];
    }
#pragma unroll
  for (int b = NB - 1; b >= 1; b--)
    if (TP::moves(b) && TP::parent(b) != 0 && TP::moves(TP::parent(b))) {
#pragma unroll
      for (int c = 0; c < 10; c++) crb[TP::parent(b)][c] += crb[b][c];
    }
#pragma unroll
  for (int i = 0; i < NV; i++) {
    T buf[6];
    mul_inert_vec(buf, crb[TP::jbody(i)], cdof[i]);
    M[i][i] = m.dof_armature[i] + dot6(cdof[i], buf);
#pragma unroll
    for (int j = 0; j < i; j++) M[i][j] = TP::dof_ancestor(i, j) ? dot6(cdof[j], buf) : T(0);
  }
  // M = L D L' (unit lower L, stored in Lm below the diagonal; Dinv = 1/D)
  T Lm[NV][NV], Dinv[NV];
  ldl_factor<NV>(Lm, Dinv, M);

  // ================= velocity stage: comVel, passive, RNE bias
  T cvel[NB][6], cdof_dot[NV][6];
#pragma unroll
  for (int b = 1; b < NB; b++) {
    if (!TP::moves(b)) continue;
    T v[6];
    if (TP::parent(b) == 0 || !TP::moves(TP::parent(b))) {
#pragma unroll
      for (int c = 0; c < 6; c++) v[c] = 0;
    } else {
#pragma unroll
      for (int c = 0; c < 6; c++) v[c] = cvel[TP::parent(b)][c];
    }
#pragma unroll
    for (int j = 0; j < NV; j++)
      if (TP::jbody(j) == b) {
        cross_motion(cdof_dot[j], v, cdof[j]);
#pragma unroll
        for (int c = 0; c < 6; c++) v[c] += cdof[j][c] * qvel[j];
      }
#pragma unroll
    for (int c = 0; c < 6; c++) cvel[b][c] = v[c];
  }
  const bool passive_on = !(m.disableflags & (1 << 5));
#pragma unroll
  for (int j = 0; j < NV; j++) {
    T f = 0;
    if (passive_on) f = -m.jnt_stiffness[j] * (qpos[j] - m.qpos_spring[j]) - m.dof_damping[j] * qvel[j];
    qfrc[j] = f;
  }
  {
    T cfrc[NB][6];
    T g[3] = {0, 0, 0};
    if (!(m.disableflags & (1 << 6))) { g[0] = -m.gravity[0]; g[1] = -m.gravity[1]; g[2] = -m.gravity[2]; }
    T cacc[NB][6];
#pragma unroll
    for (int b = 1; b < NB; b++) {
      if (!TP::moves(b)) continue;
      if (TP::parent(b) == 0 || !TP::moves(TP::parent(b))) {
        cacc[b][0] = cacc[b][1] = cacc[b][2] = 0;
        cacc[b][3] = g[0]; cacc[b][4] = g[1]; cacc[b][5] = g[2];
      } else {
#pragma unroll
        for (int c = 0; c < 6; c++) cacc[b][c] = cacc[TP::parent(b)][c];
      }
#pragma unroll
      for (int j = 0; j < NV; j++)
        if (TP::jbody(j) == b) {
#pragma unroll
          for (int c = 0; c < 6; c++) cacc[b][c] += cdof_dot[j][c] * qvel[j];
        }
      T t1[6], t2[6], t3[6];
      mul_inert_vec(t1, cinert[b], cacc[b]);
      mul_inert_vec(t2, cinert[b], cvel[b]);
      cross_force(t3, cvel[b], t2);
#pragma unroll
      for (int c = 0; c < 6; c++) cfrc[b][c] = t1[c] + t3[c];
    }
#pragma unroll
    for (int b = NB - 1; b >= 1; b--)
      if (TP::moves(b) && TP::parent(b) != 0 && TP::moves(TP::parent(b))) {
#pragma unroll
        for (int c = 0; c < 6; c++) cfrc[TP::parent(b)][c] += cfrc[b][c];
      }
#pragma unroll
    for (int j = 0; j < NV; j++) qfrc[j] -= dot6(cdof[j], cfrc[TP::jbody(j)]);  // - qfrc_bias
  }

  // ================= actuation (joint transmission)
  if (!(m.disableflags & (1 << 10))) {
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const int j = TP::actj(u);
      T c = ctrl[u];
      if (m.act_ctrllimited[u] && !(m.disableflags & (1 << 7))) c = clampv(c, m.act_ctrlrange[u][0], m.act_ctrlrange[u][1]);
      T force = m.act_gain[u] * c;
      if (m.act_biastype[u] == 1)
        force += m.act_bias[u][0] + m.act_bias[u][1] * m.act_gear[u] * qpos[j] + m.act_bias[u][2] * m.act_gear[u] * qvel[j];
      if (m.act_forcelimited[u]) force = clampv(force, m.act_forcerange[u][0], m.act_forcerange[u][1]);
      qfrc[j] += m.act_gear[u] * force;
    }
  }

  // mj_xfrcAccumulate (Trajectory::NoisyRollout): Cartesian force / torque on each body, through the dofs above it; bodies in
  // ascending order per dof, as oracle/physics.c o_xfrc_accumulate sums them
  if (xfrc) {
#pragma unroll
    for (int b = 1; b < NB; b++) {
      if (TP::moves(b)) {
        T off[3], tq[3];
#pragma unroll
        for (int c = 0; c < 3; c++) off[c] = xipos[b][c] - com[TP::root(b)][c];
        const T f[3] = {xfrc[b][0], xfrc[b][1], xfrc[b][2]};
        cross3(tq, off, f);
#pragma unroll
        for (int c = 0; c < 3; c++) tq[c] += xfrc[b][3 + c];
#pragma unroll
        for (int i = 0; i < NV; i++)
          if (TP::dof_moves_body(i, b))
            qfrc[i] += cdof[i][0] * tq[0] + cdof[i][1] * tq[1] + cdof[i][2] * tq[2] + cdof[i][3] * f[0] + cdof[i][4] * f[1] + cdof[i][5] * f[2];
      }
    }
  }

  // ================= acceleration stage
  ldl_solve<NV>(qacc, Lm, Dinv, qfrc);  // qacc_smooth

  // joint-limit rows: at most one side per joint can be active (checked at create time);
  // lanes without an active row are predicated, waves without any skip the solve.
#pragma unroll
  for (int j = 0; j < NV; j++) qfrc_c[j] = 0;
  if (TP::num_limited() > 0 && !(m.disableflags & ((1 << 0) | (1 << 3)))) {
    bool act[NV];
    T sgn[NV], dist[NV];
    bool any = false;
#pragma unroll
    for (int j = 0; j < NV; j++) {
      act[j] = false; sgn[j] = 0; dist[j] = 0;
      if (TP::jlimited(j)) {
        const T dlo = qpos[j] - m.jnt_range[j][0], dhi = m.jnt_range[j][1] - qpos[j];
        if (dlo < m.jnt_margin[j]) { act[j] = true; sgn[j] = 1; dist[j] = dlo; }
        else if (dhi < m.jnt_margin[j]) { act[j] = true; sgn[j] = -1; dist[j] = dhi; }
        any |= act[j];
      }
    }
    if (__any(any)) {
      // Minv columns of the limited dofs, A = J Minv J' + R, b = J qacc_smooth - aref
      T Mi[NV][NV];  // Mi[j] = Minv e_j (only limited j used)
      T AR[NV][NV], bb[NV], Rr[NV], f[NV];
#pragma unroll
      for (int j = 0; j < NV; j++) {
        if (!TP::jlimited(j)) continue;
        T e[NV];
#pragma unroll
        for (int c = 0; c < NV; c++) e[c] = (c == j) ? T(1) : T(0);
        ldl_solve<NV>(Mi[j], Lm, Dinv, e);
        // impedance / reference (mj_makeImpedance)
        const T pos = dist[j] - m.jnt_margin[j];
        T dmin = clampv(m.jnt_solimp[j][0], T(kMinImp), T(kMaxImp));
        T dmax = clampv(m.jnt_solimp[j][1], T(kMinImp), T(kMaxImp));
        const T width = m.jnt_solimp[j][2];
        T mid = clampv(m.jnt_solimp[j][3], T(kMinImp), T(kMaxImp));
        T power = m.jnt_solimp[j][4] < 1 ? T(1) : m.jnt_solimp[j][4];
        T imp;
        if (dmin == dmax || width <= T(kMinVal)) {
          imp = T(0.5) * (dmin + dmax);
        } else {
          const T x = fabs(pos) / width;
          if (x >= 1) imp = dmax;
          else if (x <= 0) imp = dmin;
          else {
            T y;
            if (power == 1) y = x;
            else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
            else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
            imp = dmin + y * (dmax - dmin);
          }
        }
        T kk, bd;
        if (m.jnt_solref[j][0] > 0) {
          T tc = m.jnt_solref[j][0];
          if (!(m.disableflags & (1 << 11)) && tc < 2 * h) tc = 2 * h;
          kk = T(1) / (dmax * dmax * tc * tc * m.jnt_solref[j][1] * m.jnt_solref[j][1]);
          bd = T(2) / (dmax * tc);
        } else {
          kk = -m.jnt_solref[j][0] / (dmax * dmax);
          bd = -m.jnt_solref[j][1] / dmax;
        }
        const T aref = -bd * (sgn[j] * qvel[j]) - kk * imp * pos;
        T R = (1 - imp) / imp * m.dof_invweight0[j];
        Rr[j] = R < T(kMinVal) ? T(kMinVal) : R;
        bb[j] = sgn[j] * qacc[j] - aref;
        f[j] = 0;
      }
#pragma unroll
      for (int r = 0; r < NV; r++)
#pragma unroll
        for (int s = 0; s < NV; s++)
          if (TP::jlimited(r) && TP::jlimited(s)) AR[r][s] = sgn[r] * sgn[s] * Mi[s][r] + (r == s ? Rr[r] : T(0));
      // projected Gauss-Seidel on the dual (MuJoCo PGS), rows in joint order
      const T scale = T(1) / (m.meaninertia * T(NV > 1 ? NV : 1));
      bool done = !any;
      for (int it = 0; it < m.solver_iterations; it++) {
        T improvement = 0;
#pragma unroll
        for (int r = 0; r < NV; r++) {
          if (!TP::jlimited(r)) continue;
          if (act[r]) {
            T res = bb[r];
#pragma unroll
            for (int s = 0; s < NV; s++)
              if (TP::jlimited(s)) res += act[s] ? AR[r][s] * f[s] : T(0);
            const T old = f[r];
            T fn = old - res / AR[r][r];
            fn = fn < 0 ? T(0) : fn;
            if (!done) {
              f[r] = fn;
              const T delta = fn - old;
              improvement -= T(0.5) * delta * delta * AR[r][r] + delta * res;
            }
          }
        }
        done |= improvement * scale < m.solver_tolerance;
        if (__all(done)) break;
      }
#pragma unroll
      for (int r = 0; r < NV; r++)
        if (TP::jlimited(r)) {
          const T fr = act[r] ? f[r] : T(0);
          qfrc_c[r] += sgn[r] * fr;
#pragma unroll
          for (int c = 0; c < NV; c++) qacc[c] += Mi[r][c] * sgn[r] * fr;
        }
    }
  }

}

// the ResidualFn::Residual overrides of the registered tasks (the mjcb_sensor callback at mjSTAGE_ACC)
template <class TP, class TK, typename T>
__device__ __forceinline__ void lane_residual(const LaneTask<T>& tk, const T (&qpos)[TP::NV], const T (&qvel)[TP::NV],
                                              const T (&ctrl)[TP::NU], T (&r)[TK::NR]) {
  constexpr int NV = TP::NV;
  if (TK::RID == 1) {  // Particle: mjpc/test/testdata/particle_residual.h:33-43
    r[0] = qpos[0] - tk.mocap_pos[0][0];
    r[1] = qpos[1] - tk.mocap_pos[0][1];
    r[2] = qvel[0];
    r[3] = qvel[1];
  } else if (TK::RID == 2) {  // ParticleCopy: mjpc/test/agent/rollout_test.cc:37-42
#pragma unroll
    for (int i = 0; i < NV; i++) { r[i] = qpos[i]; r[NV + i] = qvel[i]; }
  } else if (TK::RID == 3) {  // Cartpole: mjpc/tasks/cartpole/cartpole.cc:36-49
    r[0] = cos(qpos[1]) - 1;
    r[1] = qpos[0] - tk.parameters[0];
    r[2] = qvel[1];
    r[3] = ctrl[0];
  }
}

// BaseResidualFn::CostValue (task.cc:71-110): weighted norms + exponential risk transform
template <class TK, typename T>
__device__ __forceinline__ T lane_cost(const LaneTask<T>& tk, const T (&r)[TK::NR]) {
  T cost = cost_terms<TK, T>(r, tk, std::make_integer_sequence<int, TK::NTERM>{});
  if (!(fabs(tk.risk) < T(1.0e-6))) cost = (exp(tk.risk * cost) - T(1)) / tk.risk;
  return cost;
}

// mj_Euler with implicit joint damping, then mj_advance (slide/hinge: qpos += h qvel)
template <class TP, typename T, class MODEL>
__device__ __forceinline__ void lane_euler(const MODEL& m, T (&qpos)[TP::NV], T (&qvel)[TP::NV], const T (&qacc)[TP::NV],
                                           const T (&qfrc)[TP::NV], const T (&qfrc_c)[TP::NV],
                                           const T (&M)[TP::NV][TP::NV]) {
  constexpr int NV = TP::NV;
  const T h = m.timestep;
  // ================= mj_Euler: implicit joint damping, then advance
  T qdd[NV];
  if (m.any_damping && !(m.disableflags & (1 << 14))) {
    T Mh[NV][NV], L2[NV][NV], D2[NV], rhs[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) {
#pragma unroll
      for (int j = 0; j < i; j++) Mh[i][j] = M[i][j];
      Mh[i][i] = M[i][i] + h * m.dof_damping[i];
      rhs[i] = qfrc[i] + qfrc_c[i];
    }
    ldl_factor<NV>(L2, D2, Mh);
    ldl_solve<NV>(qdd, L2, D2, rhs);
  } else {
#pragma unroll
    for (int i = 0; i < NV; i++) qdd[i] = qacc[i];
  }
#pragma unroll
  for (int i = 0; i < NV; i++) {
    qvel[i] += h * qdd[i];
    qpos[i] += h * qvel[i];
  }
}

// mj_step's integrator after the step's mj_forward: mj_Euler, or mj_RungeKutta(m, d, 4) (MuJoCo engine_forward.c). RK4: stage 0 is
// the forward pass just done (F_0 = (qvel, qacc)); stages 1..3 re-run mj_forward -- no sensor stage -- at X_0 + h A_i F_{i-1}
// (the Butcher tableau has one non-zero per row: 1/2, 1/2, 1), and the step is X_0 + h sum_i B_i F_i with B = (1/6, 1/3, 1/3,
// 1/6). Joint damping is explicit there. The three extra stages share ONE inlined copy of lane_forward (the loop is not
// unrolled), so the kernels grow by one forward pass of code, not three.
template <class TP, typename T, class MODEL, class TASKV>
__device__ __forceinline__ void lane_integrate(const MODEL& m, const TASKV& tk, T (&qpos)[TP::NV], T (&qvel)[TP::NV],
                                               const T (&ctrl)[TP::NU], const T (&qacc)[TP::NV], const T (&qfrc)[TP::NV],
                                               const T (&qfrc_c)[TP::NV], const T (&M)[TP::NV][TP::NV],
                                               const T (*xfrc)[6] = nullptr, T (*site_out)[3] = nullptr) {
  constexpr int NV = TP::NV, NS = TP::NSITE;
  if (m.integrator != 1) {
    lane_euler<TP, T>(m, qpos, qvel, qacc, qfrc, qfrc_c, M);
    return;
  }
  const T h = m.timestep;
  T q0[NV], v0[NV], kv[NV], ka[NV], sv[NV], sa[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) {
    q0[i] = qpos[i]; v0[i] = qvel[i];
    kv[i] = qvel[i]; ka[i] = qacc[i];
    sv[i] = kv[i] * T(1.0 / 6); sa[i] = ka[i] * T(1.0 / 6);
  }
#pragma unroll 1
  for (int stage = 1; stage < 4; stage++) {
    const T c = stage == 3 ? T(1) : T(0.5);
    const T b = stage == 3 ? T(1.0 / 6) : T(1.0 / 3);
#pragma unroll
    for (int i = 0; i < NV; i++) {
      qpos[i] = q0[i] + h * (c * kv[i]);
      qvel[i] = v0[i] + h * (c * ka[i]);
    }
    T qacc2[NV], qfrc2[NV], qfrc_c2[NV], M2[NV][NV];
    T site2[NS > 0 ? NS : 1][3];
    lane_forward<TP, T>(m, tk, qpos, qvel, ctrl, qacc2, qfrc2, qfrc_c2, M2, site2, xfrc);
#pragma unroll
    for (int i = 0; i < NV; i++) {
      kv[i] = qvel[i]; ka[i] = qacc2[i];
      sv[i] += b * kv[i]; sa[i] += b * ka[i];
    }
    // data->site_xpos after mj_step is the LAST stage's: that is what Trajectory::Rollout copies into the trace (trajectory.cc:165)
    if (site_out && stage == 3) {
#pragma unroll
      for (int k = 0; k < NS; k++)
#pragma unroll
        for (int c = 0; c < 3; c++) site_out[k][c] = site2[k][c];
    }
  }
#pragma unroll
  for (int i = 0; i < NV; i++) {
    qvel[i] = v0[i] + h * sa[i];
    qpos[i] = q0[i] + h * sv[i];
  }
}

// one step's Trajectory row of this candidate -> [step][field][candidate] SoA (coalesced across the wave)
template <class TP, class TK, typename T>
__device__ __forceinline__ void lane_record(const RolloutArgs<T>& a, int t, int cand, const T (&qpos)[TP::NV],
                                            const T (&qvel)[TP::NV], const T (&ctrl)[TP::NU], T time,
                                            const T (&r)[TK::NR], const T (&site_xpos)[TP::NSITE > 0 ? TP::NSITE : 1][3],
                                            T cost, bool bad) {
  constexpr int NV = TP::NV, NU = TP::NU, NR = TK::NR, NTR = TK::NTRACE, DS = 2 * NV;
  const size_t N = (size_t)a.N;
  const size_t base = (size_t)t * N + cand;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    a.states[((size_t)t * DS + i) * N + cand] = qpos[i];
    a.states[((size_t)t * DS + NV + i) * N + cand] = qvel[i];
  }
#pragma unroll
  for (int k = 0; k < NU; k++) a.actions[((size_t)t * NU + k) * N + cand] = ctrl[k];
  a.times[base] = time;
#pragma unroll
  for (int i = 0; i < NR; i++) a.residual[((size_t)t * NR + i) * N + cand] = r[i];
#pragma unroll
  for (int k = 0; k < NTR; k++)
#pragma unroll
    for (int c = 0; c < 3; c++) a.trace[((size_t)t * 3 * NTR + 3 * k + c) * N + cand] = site_xpos[TK::trace_site(k)][c];
  if (!bad) a.costs[base] = cost;
}

// the trace row of step t alone (RK4: re-recorded after the step, from the last stage's site positions)
template <class TP, class TK, typename T>
__device__ __forceinline__ void lane_record_trace(const RolloutArgs<T>& a, int t, int cand,
                                                  const T (&site_xpos)[TP::NSITE > 0 ? TP::NSITE : 1][3]) {
  constexpr int NTR = TK::NTRACE;
  const size_t N = (size_t)a.N;
#pragma unroll
  for (int k = 0; k < NTR; k++)
#pragma unroll
    for (int c = 0; c < 3; c++) a.trace[((size_t)t * 3 * NTR + 3 * k + c) * N + cand] = site_xpos[TK::trace_site(k)][c];
}

template <class TP, typename T>
__device__ __forceinline__ void lane_record_state(const RolloutArgs<T>& a, int t, int cand, const T (&qpos)[TP::NV],
                                                  const T (&qvel)[TP::NV], const T (&ctrl)[TP::NU], T time) {
  constexpr int NV = TP::NV, NU = TP::NU, DS = 2 * NV;
  const size_t N = (size_t)a.N;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    a.states[((size_t)t * DS + i) * N + cand] = qpos[i];
    a.states[((size_t)t * DS + NV + i) * N + cand] = qvel[i];
  }
#pragma unroll
  for (int k = 0; k < NU; k++) a.actions[((size_t)t * NU + k) * N + cand] = ctrl[k];
  a.times[(size_t)t * N + cand] = time;
}

// Where the model's numeric constants come from:
//   RuntimeModel  - the kernel-argument copy (general path: any model with this topology)
//   StaticXxx     - a generated constexpr object (generated/static_models.h): every constant is an
//                   immediate and the compiler folds the arithmetic on exact zeros / ones away.
struct RuntimeModel {
  template <typename T> __device__ static __forceinline__ const LaneModel<T>& get(const LaneModel<T>& karg) { return karg; }
};
template <class Gen>
struct StaticModel {  // returns the constexpr object BY VALUE: a local constant the optimiser scalarises
  template <typename T> __device__ static __forceinline__ constexpr LaneModel<T> get(const LaneModel<T>&) {
    return Gen::template make<T>();
  }
};

// ------------------------------------------------------------------ the kernel
// SPLIT = true is the first of three launches (launch_lane_impl): the time loop keeps only what the NEXT step depends on --
// policy, mj_forward, mj_Euler -- and records states / actions / times; the residual, the cost, the traces
// (cost_lane_kernel: one lane per (step, candidate), no serial dependence) and the ordered sum over the horizon
// (return_lane_kernel) run afterwards at full occupancy. With 64 wavefronts of one candidate per lane the loop is bound by
// its instruction count, and the sensor stage was ~40 % of it. `failure` carries (first bad step + 1) between the launches.
// NOISY = true: Trajectory::NoisyRollout -- Ornstein-Uhlenbeck xfrc_applied noise on every body (its own instantiation, so the
// plain rollout carries neither the 6 NB force registers nor the Philox draws).
template <class TP, class TK, typename T, class MC, bool SPLIT = false, bool NOISY = false>
__global__ __launch_bounds__(64) void rollout_lane_kernel(const LaneModel<T> m_karg, const LaneTask<T> tk,
                                                           const RolloutArgs<T> a) {
  decltype(auto) m = MC::template get<T>(m_karg);
  constexpr int NB = TP::NB, NV = TP::NV, NU = TP::NU, NS = TP::NSITE;
  [[maybe_unused]] constexpr int NR = TK::NR, NTR = TK::NTRACE, DS = 2 * NV;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* lnodes = reinterpret_cast<T*>(smem_raw);  // [P][NU][64]
  // node times also live in LDS: a global LOAD inside the time loop would need s_waitcnt vmcnt(0),
  // which also drains the step's 14 outstanding stores (vmcnt retires in order) -- ~1.7k cycles/step
  T* ltimes = lnodes + (size_t)a.P * NU * 64;  // [P]
  for (int p = threadIdx.x; p < a.P; p += 64) ltimes[p] = a.node_times[p];

  const int lane = threadIdx.x;
  const int cand = blockIdx.x * 64 + lane;
  const bool live = cand < a.N;
  const int ci = live ? cand : a.N - 1;  // clamp: dead lanes compute a duplicate, never store
  const size_t N = (size_t)a.N;
  const int P = a.P, H = a.H;

  // ---------------- candidate spline nodes -> LDS (+ HBM when generated here)
  if (a.noise.mode < 0) {
    for (int p = 0; p < P; p++)
#pragma unroll
      for (int k = 0; k < NU; k++) lnodes[(p * NU + k) * 64 + lane] = a.nodes[((size_t)p * NU + k) * N + ci];
  } else {
    const int gi = a.noise.candidate_offset + ci;
    double std = a.noise.std0;
    if (a.noise.mode == 0 && a.noise.std1 > 0) {
      if (bernoulli_uniform(a.noise.seed, (uint32_t)gi, a.noise.iteration) < 0.2) std = a.noise.std1;
    }
    const bool noised = gi != a.noise.nominal_candidate;
    const int np = P * NU;
    for (int j0 = 0; j0 < np; j0 += 2) {
      double z[2];
      gaussian_pair(a.noise.seed, (uint32_t)gi, (uint32_t)(j0 >> 1), a.noise.iteration, z);
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const int j = j0 + e;
        if (j < np) {
          const int k = j % NU;
          const double lo = (double)m.act_ctrlrange[k][0], hi = (double)m.act_ctrlrange[k][1];
          double v = (double)a.nominal[j];
          if (noised) {
            double sigma;
            if (a.noise.mode == 0) {
              sigma = 0.5 * (hi - lo) * std;
            } else {
              const double fl = gi < a.noise.explore_count ? a.noise.std0 : a.noise.std1;
              const double s = sqrt(a.noise.param_variance[j]);
              sigma = s > fl ? s : fl;
            }
            v = clampv(v + sigma * z[e], lo, hi);
          }
          lnodes[j * 64 + lane] = (T)v;
          if (live) a.nodes[(size_t)j * N + cand] = (T)v;
        }
      }
    }
  }
  // one wavefront per workgroup: LDS writes above are visible to this wave's own reads
  // in program order; no barrier needed.

  // ---------------- initial condition (Planner::SetState)
  T qpos[NV], qvel[NV], ctrl[NU];
#pragma unroll
  for (int i = 0; i < NV; i++) { qpos[i] = tk.qpos[i]; qvel[i] = tk.qvel[i]; }
#pragma unroll
  for (int k = 0; k < NU; k++) ctrl[k] = 0;
  T time = tk.time;
  const T h = m.timestep;

  // spline segment cache
  int up = 0, cached_up = -1;
  const T kTimeInf = T(3.0e38);
  T t_next = P > 0 ? ltimes[0] : kTimeInf;
  T sp0[NU], sp1[NU], sm0[NU], sm1[NU];
#pragma unroll
  for (int k = 0; k < NU; k++) sp0[k] = sp1[k] = sm0[k] = sm1[k] = 0;
  T tl = 0, tu = 1;

  double total = 0;
  bool failed = false;
  int first_bad = 0;
  T xfrc[NB][6];
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int c = 0; c < 6; c++) xfrc[b][c] = 0;  // (the reference inherits the pooled mjData's forces)

  for (int t = 0; t < H; t++) {
    const bool last = (t == H - 1);
    bool bad_ctrl = false;
    // ================= policy: TimeSpline::Sample + Clamp (spline.cc:103-156, policy.cc:52-59)
    if (!last) {
      // upper_bound; time is wave-uniform. The next node time sits in a register (t_next = ltimes[up], +inf past the end), so a
      // step inside a segment touches no LDS
      while (t_next <= time) { up++; t_next = up < P ? ltimes[up] : kTimeInf; }
      if (up != cached_up) {
        cached_up = up;
        const int lo = up - 1;
        if (up == P || up == 0) {
          const int n = up == 0 ? 0 : P - 1;
#pragma unroll
          for (int k = 0; k < NU; k++) sp0[k] = lnodes[(n * NU + k) * 64 + lane];
        } else {
          tl = ltimes[lo]; tu = ltimes[up];
#pragma unroll
          for (int k = 0; k < NU; k++) {
            sp0[k] = lnodes[(lo * NU + k) * 64 + lane];
            sp1[k] = lnodes[(up * NU + k) * 64 + lane];
          }
          if (a.interp == 2) {  // finite-difference slopes, spline.cc:269-287
            const T dt_mid = tu - tl;
#pragma unroll
            for (int k = 0; k < NU; k++) {
              const T fwd = (sp1[k] - sp0[k]) / dt_mid;
              if (lo == 0) {
                sm0[k] = fwd;
              } else {
                const T pv = lnodes[((lo - 1) * NU + k) * 64 + lane];
                sm0[k] = T(0.5) * (sp1[k] - sp0[k]) / dt_mid + T(0.5) * (sp0[k] - pv) / (tl - ltimes[lo - 1]);
              }
              if (up == P - 1) {
                sm1[k] = fwd;
              } else {
                const T nv = lnodes[((up + 1) * NU + k) * 64 + lane];
                sm1[k] = T(0.5) * (nv - sp1[k]) / (ltimes[up + 1] - tu) + T(0.5) * (sp1[k] - sp0[k]) / dt_mid;
              }
            }
          }
        }
      }
      if (up == P || up == 0 || a.interp == 0) {
#pragma unroll
        for (int k = 0; k < NU; k++) ctrl[k] = sp0[k];
      } else {
        const T s = (time - tl) / (tu - tl);
        if (a.interp == 1) {
#pragma unroll
          for (int k = 0; k < NU; k++) ctrl[k] = sp0[k] * (1 - s) + sp1[k] * s;
        } else {
          const T s2 = s * s, s3 = s * s * s;
          const T c0 = T(2) * s3 - T(3) * s2 + T(1);
          const T c1 = (s3 - T(2) * s2 + s) * (tu - tl);
          const T c2 = T(-2) * s3 + T(3) * s2;
          const T c3 = (s3 - s2) * (tu - tl);
#pragma unroll
          for (int k = 0; k < NU; k++) ctrl[k] = c0 * sp0[k] + c1 * sm0[k] + c2 * sp1[k] + c3 * sm1[k];
        }
      }
#pragma unroll
      for (int k = 0; k < NU; k++) {
        bad_ctrl |= is_bad(ctrl[k]);  // mjWARN_BADCTRL; tested before Clamp, which may not propagate NaN
        ctrl[k] = clampv(ctrl[k], m.act_ctrlrange[k][0], m.act_ctrlrange[k][1]);
      }
      if (bad_ctrl) {  // mj_fwdActuation: a bad control zeroes ALL controls for this forward pass (the failing step's residual sees 0)
#pragma unroll
        for (int k = 0; k < NU; k++) ctrl[k] = 0;
      }
    }
    // (last step: mj_forward with the previous control still in data->ctrl; the recorded
    //  action is a copy of the previous one, trajectory.cc:190-198)

    // ================= mj_checkPos / mj_checkVel
    bool bad = bad_ctrl;
    if (!last) {
#pragma unroll
      for (int i = 0; i < NV; i++) bad |= is_bad(qpos[i]) || is_bad(qvel[i]);
    }

    if constexpr (NOISY) {
      if (!last) {  // trajectory.cc:147-155: xfrc_applied = decay * xfrc_applied + N(0, scale) on every body entry
        const int gi = a.noise.candidate_offset + ci;
#pragma unroll
        for (int j = 0; j < 3 * NB; j++) {
          double z[2];
          gaussian_pair(a.xfrc_seed, (uint32_t)gi, (uint32_t)(t * 3 * NB + j), 0x58465243u, z);
          T* e = &xfrc[0][0] + 2 * j;
          e[0] = (T)a.xfrc_decay * e[0] + (T)(a.xfrc_scale * z[0]);
          e[1] = (T)a.xfrc_decay * e[1] + (T)(a.xfrc_scale * z[1]);
        }
      }
    }
    // ================= mj_forward for this candidate (lane_forward)
    T qacc[NV], qfrc[NV], qfrc_c[NV], M[NV][NV];
    T site_xpos[NS > 0 ? NS : 1][3];
    lane_forward<TP, T>(m, tk, qpos, qvel, ctrl, qacc, qfrc, qfrc_c, M, site_xpos, NOISY ? xfrc : nullptr);

    // ================= mj_checkAcc
    if (!last) {
#pragma unroll
      for (int i = 0; i < NV; i++) bad |= is_bad(qacc[i]);
    }

    if constexpr (SPLIT) {
      if (live && !failed) lane_record_state<TP, T>(a, t, cand, qpos, qvel, ctrl, time);
      if (bad && !failed) { failed = true; first_bad = t + 1; }
    } else {
      // ================= sensor stage: task residual (mjcb_sensor at mjSTAGE_ACC), cost
      T r[NR];
      lane_residual<TP, TK, T>(tk, qpos, qvel, ctrl, r);
      T cost = lane_cost<TK, T>(tk, r);  // task.cc:71-110

      // ================= record step t: coalesced [t][field][candidate] stores
      if (live && !failed) lane_record<TP, TK, T>(a, t, cand, qpos, qvel, ctrl, time, r, site_xpos, cost, bad);
      if (bad) failed = true;  // CheckWarnings -> abort (trajectory.cc:169-173)
      total += (double)cost;
    }
    if (last) break;

    // ================= mj_Euler (implicit joint damping) + advance
    lane_integrate<TP, T>(m, tk, qpos, qvel, ctrl, qacc, qfrc, qfrc_c, M, NOISY ? xfrc : nullptr, site_xpos);
    if (m.integrator == 1 && live && !failed) lane_record_trace<TP, TK, T>(a, t, cand, site_xpos);
    time += h;
  }

  if (live) {
    if constexpr (SPLIT) {
      a.failure[cand] = first_bad;
    } else {
      a.total_return[cand] = failed ? kMaxReturn : total / (double)(H > 1 ? H : 1);
      a.failure[cand] = failed ? 1 : 0;
    }
  }
}

// Second launch of the split rollout: the sensor stage of step t of candidate c for every (t, c) at once. Reads the recorded
// state / action, redoes the position stage (what remains of lane_forward once only site_xpos is used), evaluates the residual
// and the cost exactly as the fused loop does, and applies its recording rules: nothing after the first bad step, no cost AT it.
template <class TP, class TK, typename T, class MC>
__global__ __launch_bounds__(256) void cost_lane_kernel(const LaneModel<T> m_karg, const LaneTask<T> tk, const RolloutArgs<T> a) {
  decltype(auto) m = MC::template get<T>(m_karg);
  constexpr int NV = TP::NV, NU = TP::NU, NS = TP::NSITE, NR = TK::NR, DS = 2 * NV;
  const size_t N = (size_t)a.N;
  const size_t item = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (item >= N * (size_t)a.H) return;
  const int t = (int)(item / N), cand = (int)(item - (size_t)t * N);
  const int fb = a.failure[cand];
  if (fb && t > fb - 1) return;
  T qpos[NV], qvel[NV], ctrl[NU];
#pragma unroll
  for (int i = 0; i < NV; i++) {
    qpos[i] = a.states[((size_t)t * DS + i) * N + cand];
    qvel[i] = a.states[((size_t)t * DS + NV + i) * N + cand];
  }
#pragma unroll
  for (int k = 0; k < NU; k++) ctrl[k] = a.actions[((size_t)t * NU + k) * N + cand];
  T qacc[NV], qfrc[NV], qfrc_c[NV], M[NV][NV];
  T site_xpos[NS > 0 ? NS : 1][3];
  lane_forward<TP, T>(m, tk, qpos, qvel, ctrl, qacc, qfrc, qfrc_c, M, site_xpos);
  T r[NR];
  lane_residual<TP, TK, T>(tk, qpos, qvel, ctrl, r);
  const T cost = lane_cost<TK, T>(tk, r);
#pragma unroll
  for (int i = 0; i < NR; i++) a.residual[((size_t)t * NR + i) * N + cand] = r[i];
  // (RK4: the time loop recorded the trace of every integrated step from the last stage's site positions)
  if (m.integrator != 1 || t == a.H - 1) lane_record_trace<TP, TK, T>(a, t, cand, site_xpos);
  if (!(fb && t == fb - 1)) a.costs[item] = cost;
}

// Third launch: Trajectory::total_return = sum of the step costs in step order (the fused loop's order, so the same bits),
// divided by the horizon (trajectory.cc:203-207); kMaxReturn and failure = 1 for a failed rollout.
template <typename T>
__global__ __launch_bounds__(64) void return_lane_kernel(const RolloutArgs<T> a) {
  const int cand = blockIdx.x * 64 + threadIdx.x;
  if (cand >= a.N) return;
  const int fb = a.failure[cand];
  double total = 0;
  if (!fb) {
    // loads in batches of 32 (independent, one latency per batch), additions in step order
    int t = 0;
    for (; t + 32 <= a.H; t += 32) {
      T c[32];
#pragma unroll
      for (int i = 0; i < 32; i++) c[i] = a.costs[(size_t)(t + i) * a.N + cand];
#pragma unroll
      for (int i = 0; i < 32; i++) total += (double)c[i];
    }
    for (; t < a.H; t++) total += (double)a.costs[(size_t)t * a.N + cand];
  }
  a.total_return[cand] = fb ? kMaxReturn : total / (double)(a.H > 1 ? a.H : 1);
  a.failure[cand] = fb ? 1 : 0;
}

}  // namespace mjpcx
