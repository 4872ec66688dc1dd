// wave_ilqg.h -- iLQG pieces for the wavefront-per-candidate (contact / quaternion) models:
//   * transition_fd_wave_kernel : ModelDerivatives::Compute = Tn x mjd_transitionFD (model_derivatives.cc:45-106); ONE
//     wavefront per (time step, perturbation column) runs one mj_step of the perturbed state. Positions are perturbed
//     in the tangent space (mj_integratePos), as oracle/ilqg.c::state_perturb.
//   * fd_tangent_kernel : next states -> tangent coordinates (oracle state_tangent), after which the generic
//     fd_assemble_kernel of ilqg_kernels.h forms A, B, C, D.
//   * rollout_feedback_wave_kernel : Trajectory::RolloutDiscrete with the index policy of iLQGPlanner::ActionRollouts
//     (ilqg/planner.cc:630-692) and Trajectory::Rollout with iLQGPolicy::Action (ilqg/policy.cc:82-161), StateDiff in
//     the tangent space (mj_differentiatePos, utilities.cc:543-553).

namespace mjpcx { namespace WAVE_NS {

// rotation vector taking qb to qa in qb's frame (mju_subQuat); oracle iq_sub
__device__ __forceinline__ void wq_sub(wreal* res, const wreal* qa, const wreal* qb) {
  const wreal qn[4] = {qb[0], -qb[1], -qb[2], -qb[3]};
  wreal qd[4];
  q_mul(qd, qn, qa);
  wreal ax[3] = {qd[1], qd[2], qd[3]};
  const wreal s = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
  if (s > kMinVal) for (int k = 0; k < 3; k++) ax[k] /= s;
  wreal speed = 2 * atan2(s, qd[0]);
  if (speed > kQPi) speed -= 2 * kQPi;
  for (int k = 0; k < 3; k++) res[k] = ax[k] * speed;
}
// q <- normalize(q) * exp(v h / 2) (mju_quatIntegrate); oracle iq_integrate
__device__ __forceinline__ void wq_integrate(wreal* q, const wreal* v, wreal h) {
  wreal ax[3] = {v[0], v[1], v[2]};
  const wreal n = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
  if (n < kMinVal) { ax[0] = 1; ax[1] = ax[2] = 0; } else for (int k = 0; k < 3; k++) ax[k] /= n;
  wreal sn, cs;
  w_sincos(WL(0.5) * h * n, &sn, &cs);
  const wreal qr[4] = {cs, ax[0] * sn, ax[1] * sn, ax[2] * sn};
  q_norm(q);
  q_mul(q, q, qr);
}
// StateDiff: dx (2 nv, LDS) = (s2 - s1) / 1 in the tangent space; one lane per joint, then one per dof for velocities
template <class MODEL>
__device__ __forceinline__ void w_state_diff(const MODEL& m, wreal* dx, const wreal* s1, const wreal* s2, int lane) {
  const int nq = m.nq, nv = m.nv;
  if (lane < m.njnt) {
    int qa = m.jnt_qposadr[lane], da = m.jnt_dofadr[lane];
    const int jt = m.jnt_type[lane];
    if (jt == kJntFree) { for (int k = 0; k < 3; k++) dx[da + k] = s2[qa + k] - s1[qa + k]; qa += 3; da += 3; }
    if (jt == kJntFree || jt == kJntBall) {
      wreal a[4], b[4], r[3];
      for (int k = 0; k < 4; k++) { a[k] = s2[qa + k]; b[k] = s1[qa + k]; }
      wq_sub(r, a, b);
      for (int k = 0; k < 3; k++) dx[da + k] = r[k];
    } else dx[da] = s2[qa] - s1[qa];
  }
  if (lane < nv) dx[nv + lane] = s2[nq + lane] - s1[nq + lane];
}


// TREE: the Jacobian-free forward pass of wave_tree.h (its LDS layout, its contact lists at the large capacities) instead of the
// row-table one -- the same step function the rollout kernels of such a model use
template <int NMAX, bool TREE = false, bool RK4 = false>
__global__ __launch_bounds__(64) void transition_fd_wave_kernel(const WModel m, const WTask tk, const FdWaveArgs f) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x;
  const int t = blockIdx.x / f.ncol, col = blockIdx.x % f.ncol;
  const int nq = m.nq, nv = m.nv, nu = m.nu, ndx = 2 * nv, ds = nq + nv, nr = tk.nr;
  wreal *lnodes, *ltimes;
  TreeData tree;
  WaveData d;
  if constexpr (TREE) d = wave_carve_tree(smem_raw, m, tk, 1, ltimes, false, tree, kTreeMaxSimpleBig, kTreeMaxConeBig);
  else d = wave_carve(smem_raw, m, tk, 1, lnodes, ltimes);
  for (int i = lane; i < nq; i += 64) d.qpos[i] = f.states[(size_t)t * ds + i];
  for (int i = lane; i < nv; i += 64) d.qvel[i] = f.states[(size_t)t * ds + nq + i];
  if (lane < nu) d.ctrl[lane] = f.actions[(size_t)t * nu + lane];
  if (lane < 4) d.counters[lane] = 0;
  WSYNC();
  if (col > 0 && lane == 0) {
    int j = col - 1;
    wreal sign = 1;
    if (j < 2 * ndx) { if (j >= ndx) { j -= ndx; sign = -1; } }
    else { j -= 2 * ndx; if (j >= nu) { j -= nu; sign = -1; } j += ndx; }
    const wreal e = sign * f.eps;
    if (j >= ndx) d.ctrl[j - ndx] += e;
    else if (j >= nv) d.qvel[j - nv] += e;
    else {
      const int jn = m.dof_jntid[j], qa = m.jnt_qposadr[jn], da = m.jnt_dofadr[jn], jt = m.jnt_type[jn];
      if (jt == kJntFree && j - da < 3) d.qpos[qa + (j - da)] += e;
      else if (jt == kJntFree || jt == kJntBall) {
        wreal v[3] = {0, 0, 0}, q[4];
        const int qq = jt == kJntFree ? qa + 3 : qa, k = jt == kJntFree ? j - da - 3 : j - da;
        v[k] = 1;
        for (int c = 0; c < 4; c++) q[c] = d.qpos[qq + c];
        wq_integrate(q, v, e);
        for (int c = 0; c < 4; c++) d.qpos[qq + c] = q[c];
      } else d.qpos[qa] += e;
    }
  }
  WSYNC();
  wreal time = f.times[t];
  bool bad_ctrl = false;
  if constexpr (TREE) wt_forward<NMAX>(m, tk, d, tree, lane, bad_ctrl, nullptr, /*have_warm=*/false);
  else wf_forward<NMAX>(m, tk, d, lane, bad_ctrl, nullptr, /*have_warm=*/false);
  wr_residual(m, tk, d, time, lane);
  for (int i = lane; i < nr; i += 64) f.sensor[((size_t)t * f.ncol + col) * nr + i] = d.residual[i];
  bool rk4 = false;
  if constexpr (RK4) rk4 = m.integrator == 1;
  if (rk4) { if constexpr (RK4) (void)wave_rk4_step<NMAX, TREE>(m, tk, d, tree, lane, time, /*have_warm=*/false); }
  else if constexpr (TREE) wt_euler<NMAX>(m, d, lane, time);
  else wf_euler<NMAX>(m, d, lane, time);
  for (int i = lane; i < ds; i += 64) f.next[((size_t)t * f.ncol + col) * ds + i] = i < nq ? d.qpos[i] : d.qvel[i - nq];
}

// mjData kinematics of the state set by mjpcx_set_state, for Task::Transition implementations that read them (mj_kinematics,
// mj_comPos, mj_comVel, mj_subtreeVel): out = [xpos 3nb | xquat 4nb | xmat 9nb | xipos 3nb | site_xpos 3ns | subtree_com 3nb |
// subtree_linvel 3nb] with nb, ns = the MODEL's body / site counts (entries of bodies outside the device range stay zero).
__global__ __launch_bounds__(64) void kinematics_wave_kernel(const WModel m, const WTask tk, wreal* __restrict__ out, int nb_model,
                                                             int ns_model) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x;
  const int nq = m.nq, nv = m.nv, nb = m.nbody, ns = m.nsite;
  wreal *lnodes, *ltimes;
  WaveData d = wave_carve(smem_raw, m, tk, 1, lnodes, ltimes);
  for (int i = lane; i < nq; i += 64) d.qpos[i] = tk.blob[i];
  for (int i = lane; i < nv; i += 64) d.qvel[i] = tk.blob[nq + i];
  if (lane < 4) d.counters[lane] = 0;
  WSYNC();
  wf_kinematics(m, tk, d, lane);
  WSYNC();
  wf_compos(m, d, lane);
  wf_comvel(m, d, lane);
  WSYNC();
  wreal* o_xpos = out; wreal* o_xquat = o_xpos + 3 * nb_model; wreal* o_xmat = o_xquat + 4 * nb_model;
  wreal* o_xipos = o_xmat + 9 * nb_model; wreal* o_site = o_xipos + 3 * nb_model; wreal* o_com = o_site + 3 * ns_model;
  wreal* o_linvel = o_com + 3 * nb_model;
  for (int i = lane; i < 3 * nb; i += 64) { o_xpos[i] = d.xpos[i]; o_xipos[i] = d.xipos[i]; o_com[i] = d.subtree_com[i]; }
  for (int i = lane; i < 4 * nb; i += 64) o_xquat[i] = d.xquat[i];
  for (int i = lane; i < 9 * nb; i += 64) o_xmat[i] = d.xmat[i];
  for (int i = lane; i < 3 * ns; i += 64) o_site[i] = d.site_xpos[i];
  if (lane < nb) {  // subtree linear velocity: momentum of the bodies below / subtree mass (mj_subtreeVel)
    wreal mom[3] = {0, 0, 0};
    unsigned long long mask = m.body_subtree_mask[lane];
    while (mask) {
      const int i = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const wreal* cv = d.cvel + 6 * i;
      const wreal* com = d.subtree_com + 3 * m.body_rootid[i];
      const wreal off[3] = {d.xipos[3 * i] - com[0], d.xipos[3 * i + 1] - com[1], d.xipos[3 * i + 2] - com[2]};
      wreal lin[3];
      cr3(lin, cv, off);
      for (int k = 0; k < 3; k++) mom[k] += m.body_mass[i] * (cv[3 + k] + lin[k]);
    }
    const wreal mass = m.body_subtreemass[lane];
    for (int k = 0; k < 3; k++) o_linvel[3 * lane + k] = mass > kMinVal ? mom[k] / mass : WL(0.0);
  }
}

// raw next states [Tn][ncol][nq+nv] -> tangent coordinates [Tn][ncol][2 nv] relative to column 0 (oracle state_tangent)
__global__ void fd_tangent_kernel(const WModel m, const wreal* __restrict__ next, wreal* __restrict__ tan, int Tn, int ncol) {
  const int nq = m.nq, nv = m.nv, ds = nq + nv, ndx = 2 * nv;
  for (int item = blockIdx.x * blockDim.x + threadIdx.x; item < Tn * ncol; item += gridDim.x * blockDim.x) {
    const int t = item / ncol;
    const wreal* y = next + (size_t)item * ds;
    const wreal* y0 = next + (size_t)t * ncol * ds;
    wreal* z = tan + (size_t)item * ndx;
    for (int j = 0; j < m.njnt; j++) {
      int qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j];
      const int jt = m.jnt_type[j];
      if (jt == kJntFree) { for (int k = 0; k < 3; k++) z[da + k] = y[qa + k]; qa += 3; da += 3; }
      if (jt == kJntFree || jt == kJntBall) {
        wreal a[4], b[4], r[3];
        for (int k = 0; k < 4; k++) { a[k] = y[qa + k]; b[k] = y0[qa + k]; }
        wq_sub(r, a, b);
        for (int k = 0; k < 3; k++) z[da + k] = r[k];
      } else z[da] = y[qa];
    }
    for (int i = 0; i < nv; i++) z[nv + i] = y[nq + i];
  }
}


// one candidate's rollout under the feedback policy, by one wavefront; MODEL / TASK: the generic structs (arrays behind global pointers)
// or a registered model's LDS image (lds_model.h)
template <int NMAX, bool TREE, bool RK4, class MODEL, class TASK>
__device__ __forceinline__ void feedback_rollout_body(const MODEL& m, const TASK& tk, const RolloutArgs<wreal>& a, const FeedbackWaveArgs& fb,
                                                      unsigned char* smem_raw, int cand, int lane) {
  const int nq = m.nq, nv = m.nv, nu = m.nu, ndx = 2 * nv, ds = nq + nv, nr = tk.nr, H = a.H, Tn = fb.Tn;
  wreal *lnodes, *ltimes;
  // the policy scratch (dx[ndx], interpolated state[ds], current state[ds]) lives where the spline nodes would be
  TreeData tree;
  WaveData d;
  if constexpr (TREE) { d = wave_carve_tree(smem_raw, m, tk, /*P=*/ndx + 2 * ds, ltimes, false, tree, kTreeMaxSimpleBig, kTreeMaxConeBig); lnodes = ltimes; }
  else d = wave_carve(smem_raw, m, tk, /*P=*/(ndx + 2 * ds + nu - 1) / nu + 1, lnodes, ltimes);
  wreal* dx = lnodes; wreal* xi = dx + ndx; wreal* xs = xi + ds;
  for (int i = lane; i < nq; i += 64) d.qpos[i] = tk.blob[i];
  for (int i = lane; i < nv; i += 64) d.qvel[i] = tk.blob[nq + i];
  if (lane < nu) d.ctrl[lane] = 0;
  if (lane < 4) d.counters[lane] = 0;
  wreal time = tk.blob[tk.off_time];
  const wreal alpha = fb.alpha[cand];
  WSYNC();
  wreal total = 0;
  bool failed = false;
  for (int t = 0; t < H; t++) {
    const bool last = t == H - 1;
    bool bad = false;
    if (!last) {
      for (int i = lane; i < ds; i += 64) xs[i] = i < nq ? d.qpos[i] : d.qvel[i - nq];
      WSYNC();
      wreal u = 0;
      if (fb.mode == 0) {  // index policy: u = actions[t] + alpha improvement[t] + K[t] StateDiff(states[t], x)
        const int tt = t < Tn ? t : Tn - 1;
        w_state_diff(m, dx, fb.states + (size_t)tt * ds, xs, lane);
        WSYNC();
        if (lane < nu) {
          u = fb.actions[(size_t)tt * nu + lane] + alpha * fb.improvement[(size_t)tt * nu + lane];
          wreal s = 0;
          const wreal* K = fb.gains + ((size_t)tt * nu + lane) * ndx;
          for (int j = 0; j < ndx; j++) s += K[j] * dx[j];
          u += s;
        }
      } else {  // iLQGPolicy::Action at `time`
        int b0, b1;
        find_interval(fb.times, time, Tn, b0, b1);
        const int rep = (b0 == b1) ? 0 : fb.representation;
        // weights over the shorter range the reference passes for actions / gains (Tn - 1 entries) and over Tn for states
        const InterpWeights<wreal> wa = interp_weights(fb.times, time, Tn - 1, rep);
        if (lane < nu) {
          u = 0;
          for (int q = 0; q < 4; q++) u += wa.w[q] * fb.actions[(size_t)wa.i[q] * nu + lane];
        }
        if (fb.use_state) {
          const InterpWeights<wreal> ws = interp_weights(fb.times, time, Tn, rep);
          for (int i = lane; i < ds; i += 64) {
            wreal v = 0;
            for (int q = 0; q < 4; q++) v += ws.w[q] * fb.states[(size_t)ws.i[q] * ds + i];
            xi[i] = v;
          }
          WSYNC();
          if (lane < m.njnt) {  // policy.cc:118-125: renormalise interpolated quaternions
            const int jt = m.jnt_type[lane];
            if (jt == kJntFree || jt == kJntBall) {
              wreal* q = xi + m.jnt_qposadr[lane] + (jt == kJntFree ? 3 : 0);
              wreal qq[4] = {q[0], q[1], q[2], q[3]};
              q_norm(qq);
              for (int k = 0; k < 4; k++) q[k] = qq[k];
            }
          }
          WSYNC();
          w_state_diff(m, dx, xi, xs, lane);
          WSYNC();
          if (lane < nu) {
            wreal s = 0;
            for (int j = 0; j < ndx; j++) {
              wreal kj = 0;
              for (int q = 0; q < 4; q++) kj += wa.w[q] * fb.gains[((size_t)wa.i[q] * nu + lane) * ndx + j];
              s += kj * dx[j];
            }
            u += alpha * s;
          }
        }
      }
      if (lane < nu) {
        bad = is_bad(u);
        d.ctrl[lane] = clampv(u, m.actuator_ctrlrange[2 * lane], m.actuator_ctrlrange[2 * lane + 1]);
      }
      for (int i = lane; i < nq; i += 64) bad |= is_bad(d.qpos[i]);
      for (int i = lane; i < nv; i += 64) bad |= is_bad(d.qvel[i]);
    }
    WSYNC();
    bool bad_ctrl = false;
    if constexpr (TREE) wt_forward<NMAX>(m, tk, d, tree, lane, bad_ctrl, nullptr, t > 0);
    else wf_forward<NMAX>(m, tk, d, lane, bad_ctrl, nullptr, t > 0);
    if (!last) for (int i = lane; i < nv; i += 64) bad |= is_bad(d.qacc[i]);
    if (!last) bad |= d.counters[2] != 0;  // CheckWarnings: contact / row cap overflow, indefinite Hessian (oracle odata_warning)
    bad = __any(bad);
    wr_residual(m, tk, d, time, lane);
    // cost terms (task.cc:71-110): the per-entry part of every norm one lane per residual entry (scratch: efc_J, dead after
    // the solve), then one lane per term sums its entries in order and applies the norm's outer function and the weight
    for (int i = lane; i < nr; i += 64) {
      const int k = tk.res_term[i];
      d.efc_J[i] = w_norm_elem(d.residual[i], tk.norm[k], tk.blob[tk.off_normp + k], tk.blob[tk.off_normq + k]);
    }
    WSYNC();
    if (lane < tk.nterm) {
      const int off = tk.term_off[lane], n = tk.dim_norm_residual[lane];
      wreal c = 0;
      for (int i = 0; i < n; i++) c += d.efc_J[off + i];
      d.terms[lane] = tk.blob[tk.off_weight + lane] * w_norm_finish(c, tk.norm[lane], tk.blob[tk.off_normp + lane], tk.blob[tk.off_normq + lane]);
    }
    WSYNC();
    wreal cost = 0;
    for (int k = 0; k < tk.nterm; k++) cost += d.terms[k];
    const wreal risk = tk.blob[tk.off_risk];
    if (!(fabs(risk) < WL(1.0e-6))) cost = (exp(risk * cost) - WL(1.0)) / risk;
    if (!failed) {
      for (int i = lane; i < ds; i += 64) a.states[((size_t)cand * H + t) * ds + i] = i < nq ? d.qpos[i] : d.qvel[i - nq];
      if (lane < nu) a.actions[((size_t)cand * H + t) * nu + lane] = d.ctrl[lane];
      for (int i = lane; i < nr; i += 64) a.residual[((size_t)cand * H + t) * nr + i] = d.residual[i];
      if (lane < 3 * tk.ntrace) {
        const int ts = tk.trace_site[lane / 3];  // site id, or -1 - body id for a body frame
        a.trace[((size_t)cand * H + t) * 3 * tk.ntrace + lane] = ts >= 0 ? d.site_xpos[3 * ts + lane % 3] : d.xpos[3 * (-1 - ts) + lane % 3];
      }
      if (lane == 0) {
        a.times[(size_t)cand * H + t] = time;
        if (!bad) a.costs[(size_t)cand * H + t] = cost;
      }
    }
    if (bad) { failed = true; break; }  // RolloutDiscrete returns at the first warning (trajectory.cc:268-272)
    total += cost;
    if (last) break;
    if constexpr (RK4) {
      if (m.integrator == 1) {
        const bool warned = wave_rk4_step<NMAX, TREE>(m, tk, d, tree, lane, time, /*have_warm=*/t > 0);
        if (lane < 3 * tk.ntrace) {  // the trace of an RK4 step is the last stage's (wave_kernel.h)
          const int ts = tk.trace_site[lane / 3];
          a.trace[((size_t)cand * H + t) * 3 * tk.ntrace + lane] = ts >= 0 ? d.site_xpos[3 * ts + lane % 3] : d.xpos[3 * (-1 - ts) + lane % 3];
        }
        if (warned) { failed = true; break; }
        continue;
      }
    }
    if (lane < nv) d.qacc_warm[lane] = d.qacc[lane];
    if constexpr (TREE) wt_euler<NMAX>(m, d, lane, time);
    else wf_euler<NMAX>(m, d, lane, time);
  }
  if (lane == 0) {
    a.total_return[cand] = failed ? kMaxReturn : total / (wreal)(H > 1 ? H : 1);
    a.failure[cand] = failed ? 1 : 0;
  }
}

template <int NMAX, bool TREE = false, bool RK4 = false>
__global__ __launch_bounds__(64) void rollout_feedback_wave_kernel(const WModel m, const WTask tk, const RolloutArgs<wreal> a,
                                                                    const FeedbackWaveArgs fb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (fb.only_flagged && !(a.failure[blockIdx.x] & kQFallback)) return;  // (workgroup-uniform)
  feedback_rollout_body<NMAX, TREE, RK4>(m, tk, a, fb, smem_raw, blockIdx.x, threadIdx.x);
}

// the same for a REGISTERED model (tree_registry.h): its image and the per-plan blob staged into LDS as rollout_tree_kernel does
// (tree_kernel.h). The iLQG phases are 1 and ~10 rollouts of pure per-step latency, and every model read on that chain is an LDS read
// at a compile-time offset instead of a load through the caches: 0.20 against 0.26 ms per step (profiles/r03_latency_probe.log).
template <class C>
__global__ __launch_bounds__(64) void rollout_feedback_tree_kernel(const WModel m_in, const WTask tk_in, const RolloutArgs<wreal> a,
                                                                    const FeedbackWaveArgs fb, const unsigned char* __restrict__ image,
                                                                    unsigned blob_bytes) {
  typedef LdsLayout<C, wreal> L;
  const int lane = threadIdx.x;
  if (fb.only_flagged && !(a.failure[blockIdx.x] & kQFallback)) return;  // (workgroup-uniform)
  {
    const uint4* src = reinterpret_cast<const uint4*>(image);
    uint4* dst = reinterpret_cast<uint4*>(mjpcx_lds);
    for (unsigned i = lane; i < L::kBytes / 16; i += 64) dst[i] = src[i];
    const uint4* bs = reinterpret_cast<const uint4*>(tk_in.blob);
    uint4* bd = reinterpret_cast<uint4*>(mjpcx_lds + L::kBytes);
    for (unsigned i = lane; i < blob_bytes / 16; i += 64) bd[i] = bs[i];
  }
  __syncthreads();
  const LdsModelT<C, wreal> m(m_in);
  const LdsTaskT<C, wreal> tk(tk_in, reinterpret_cast<const wreal*>(mjpcx_lds + L::kBytes));
  feedback_rollout_body<C::NMAX, true, false>(m, tk, a, fb, mjpcx_lds + L::kBytes + blob_bytes, blockIdx.x, lane);
}

} }  // namespace mjpcx::WAVE_NS
