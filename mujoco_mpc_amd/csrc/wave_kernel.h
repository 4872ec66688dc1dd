// wave_kernel.h -- Trajectory::Rollout (mjpc/trajectory.cc:100-210) + UpdateReturn (:312-326) for one candidate per
// wavefront, on top of wf_forward / wf_euler. Candidate generation, spline policy, failure semantics and the output
// layout ([step][field][candidate]) are those of rollout_lane_kernel, so every downstream entry point (best, top-k,
// elite moments, fetch) serves both kernel families.

namespace mjpcx { namespace WAVE_NS {

// LDS footprint of one candidate in elements of the working type (ints and the contact structs are rounded up to it)
__host__ __device__ inline size_t wave_lds_elems(int nq, int nv, int nu, int nbody, int njnt, int nsite, int nr, int nterm, int P, int cone = 1, bool nodes_in_lds = true,
                                                 bool xfrc = false) {
  size_t n = 0;
  n += nq + nv + nu;                                   // qpos qvel ctrl
  n += 3 * nbody + 4 * nbody + 9 * nbody + 3 * nbody + 9 * nbody + 3 * nsite;  // kinematics (xanchor / xaxis alias efc_J)
  n += 3 * nbody + 10 * nbody + 6 * nv * 2 + 6 * nbody * 4 + 3;                           // com, cinert (crb aliases cacc|cfrc), spatial
  n += 2 * (size_t)nv * nv + nv;                       // M H (the factor of M lives in H until Newton) + reciprocal pivots
  n += 7 * nv + nu + 5 * nv;                           // qfrc_*, qacc*, actuator_force, grad search Ma tmpv qacc_warm
  n += (size_t)kWaveMaxEfc * nv + 7 * kWaveMaxEfc;     // efc_J + per-row reals (efc_pos / efc_margin alias jar / jv)
  n += (3 * kWaveMaxEfc * sizeof(int) + sizeof(wreal) - 1) / sizeof(wreal) + 1;  // per-row ints
  n += (cone == 1 ? 21 * kWaveMaxCon : 0) + 12 + nr + nterm + 8;  // coneH (lower triangles; elliptic cones only) foot_xpos residual terms scal
  n += (sizeof(WaveContact) * kWaveMaxCon + sizeof(wreal) - 1) / sizeof(wreal) + 1;
  n += 4 * sizeof(int) / sizeof(wreal) + 1;           // counters
  n += (nodes_in_lds ? (size_t)P * nu : 0) + P;        // spline nodes (the rollout kernel reads them from HBM / L2) + node times
  n += xfrc ? 6 * nbody : 0;                          // xfrc_applied (NoisyRollout only)
  return n + 4;
}

// the LDS layout of one candidate's mjData (shared by the rollout, feedback-rollout and finite-difference kernels)
template <class MODEL, class TASK>
__device__ __forceinline__ WaveData wave_carve(unsigned char* smem_raw, const MODEL& m, const TASK& tk, int P, wreal*& lnodes,
                                               wreal*& ltimes, bool nodes_in_lds = true, bool xfrc = false) {
  const int nq = m.nq, nv = m.nv, nu = m.nu, nb = m.nbody, nj = m.njnt, ns = m.nsite, nr = tk.nr;
  wreal* p = reinterpret_cast<wreal*>(smem_raw);
  auto take = [&](size_t n) { wreal* q = p; p += n; return q; };
  WaveData d;
  d.qpos = take(nq); d.qvel = take(nv); d.ctrl = take(nu);
  d.xpos = take(3 * nb); d.xquat = take(4 * nb); d.xmat = take(9 * nb); d.xipos = take(3 * nb); d.ximat = take(9 * nb);
  d.site_xpos = take(3 * ns);
  d.subtree_com = take(3 * nb); d.cinert = take(10 * nb); d.cdof = take(6 * nv); d.cdof_dot = take(6 * nv);
  d.cvel = take(6 * nb); d.cacc = take(6 * nb); d.cfrc = take(6 * nb); d.cfrc_sub = take(6 * nb); d.subtree_linvel = take(3);
  d.crb = d.cacc;  // composite inertias (10 nb) are dead before RNE writes cacc | cfrc (12 nb, contiguous)
  d.M = take((size_t)nv * nv); d.H = take((size_t)nv * nv); d.Ldinv = take(nv); d.dinv = d.Ldinv;  // the pivots of chol(M) are dead once qacc_smooth is solved (L aliases H likewise)
  d.L = d.H;       // the factor of M is only needed until qacc_smooth is solved, before Newton builds H
  d.qfrc_passive = take(nv); d.qfrc_bias = take(nv); d.qfrc_actuator = take(nv); d.qfrc_smooth = take(nv);
  d.qacc_smooth = take(nv); d.qacc = take(nv); d.qfrc_constraint = take(nv); d.actuator_force = take(nu);
  d.grad = take(nv); d.search = take(nv); d.Ma = take(nv); d.Ms = nullptr; d.tmpv = take(nv); d.qacc_warm = take(nv);
  d.efc_J = take((size_t)kWaveMaxEfc * nv);
  d.xanchor = d.efc_J; d.xaxis = d.efc_J + 3 * nj;  // joint anchors / axes are dead after cdof, before the rows are assembled
  d.efc_D = take(kWaveMaxEfc); d.efc_R = take(kWaveMaxEfc);
  d.efc_aref = take(kWaveMaxEfc); d.efc_floss = take(kWaveMaxEfc); d.efc_force = take(kWaveMaxEfc); d.jar = take(kWaveMaxEfc);
  d.jv = take(kWaveMaxEfc);
  d.efc_pos = d.jar; d.efc_margin = d.jv;  // row assembly only; the Newton solver overwrites jar / jv afterwards
  int* ip = reinterpret_cast<int*>(take((3 * kWaveMaxEfc * sizeof(int) + sizeof(wreal) - 1) / sizeof(wreal) + 1));
  d.efc_type = ip; d.efc_id = ip + kWaveMaxEfc; d.efc_zone = ip + 2 * kWaveMaxEfc;
  d.coneH = m.cone == 1 ? take(21 * kWaveMaxCon) : nullptr; d.foot_xpos = take(12); d.residual = take(nr); d.terms = take(tk.nterm); d.scal = take(8);
  d.con = reinterpret_cast<WaveContact*>(take((sizeof(WaveContact) * kWaveMaxCon + sizeof(wreal) - 1) / sizeof(wreal) + 1));
  d.counters = reinterpret_cast<int*>(take(4 * sizeof(int) / sizeof(wreal) + 1));
  lnodes = nodes_in_lds ? take((size_t)P * nu) : nullptr;  // [P][nu]
  ltimes = take(P);
  d.xfrc = xfrc ? take(6 * nb) : nullptr;

  return d;
}

// One candidate's Trajectory::Rollout + UpdateReturn by one wavefront; `smem_raw` is the wavefront's own arena in LDS.
// mj_RungeKutta(m, d, 4), MuJoCo engine_forward.c (restated in oracle/physics.c o_rk4), after the step's own forward pass: F_0 = (qvel,
// qacc) of that pass; stages 1..3 at X_0 + h A_i F_{i-1} with A = (1/2, 1/2, 1), no sensor stage, the solver warm-started from the
// previous STEP in every stage; then X_0 + h sum B_i F_i, B = (1/6, 1/3, 1/3, 1/6), and the last stage's qacc becomes the next
// step's warm start. No implicit joint damping. The stage values live in registers (one lane per coordinate: nq, nv <= 64), so
// the LDS arena is the Euler kernel's. On return d holds the last stage's kinematics (site_xpos / xpos: the trace of the step) and
// the advanced state; the result says whether any stage raised a solver / capacity warning (wave-uniform).
template <int NMAX, bool TREE, class MODEL, class TASK>
__device__ __forceinline__ bool wave_rk4_step(const MODEL& m, const TASK& tk, WaveData& d, TreeData& tree, int lane, wreal& time, bool have_warm) {
  const int nq = m.nq, nv = m.nv;
  const wreal hh = m.timestep, t0 = time;
  const wreal q0 = lane < nq ? d.qpos[lane] : WL(0.0), v0 = lane < nv ? d.qvel[lane] : WL(0.0);
  wreal kv = v0, ka = lane < nv ? d.qacc[lane] : WL(0.0);
  wreal sv = WL(1.0) / WL(6.0) * kv, sa = WL(1.0) / WL(6.0) * ka;
#pragma unroll 1
  for (int stage = 1; stage < 4; stage++) {
    const wreal c = stage == 3 ? WL(1.0) : WL(0.5), b = stage == 3 ? WL(1.0) / WL(6.0) : WL(1.0) / WL(3.0);
    WSYNC();
    if (lane < nq) d.qpos[lane] = q0;
    if (lane < nv) { d.tmpv[lane] = c * kv; d.qvel[lane] = v0 + hh * (c * ka); }
    WSYNC();
    w_integrate_pos(m, d, d.tmpv, hh, lane);
    WSYNC();
    time = t0 + c * hh;
    bool bc2 = false;
    if constexpr (TREE) wt_forward<NMAX>(m, tk, d, tree, lane, bc2, nullptr, have_warm);
    else wf_forward<NMAX>(m, tk, d, lane, bc2, nullptr, have_warm);
    if (lane < nv) { kv = d.qvel[lane]; ka = d.qacc[lane]; }
    sv += b * kv; sa += b * ka;
  }
  const bool warned = d.counters[2] != 0;
  WSYNC();
  if (lane < nq) d.qpos[lane] = q0;
  if (lane < nv) { d.tmpv[lane] = sv; d.qvel[lane] = v0 + hh * sa; d.qacc_warm[lane] = d.qacc[lane]; }
  WSYNC();
  w_integrate_pos(m, d, d.tmpv, hh, lane);
  time = t0 + hh;
  WSYNC();
  return __any(warned);
}

// RK4 = true compiles mj_RungeKutta(m, d, 4) in beside mj_Euler (chosen at run time by the model's integrator): three more forward
// passes per step, which share one extra inlined copy of the forward pass. The stage values live in registers (one lane per
// coordinate), so the LDS arena is the Euler kernel's.
template <int NMAX, bool TREE, int CAPS = kTreeMaxSimple, int CAPC = kTreeMaxCone, bool RK4 = false, class MODEL, class TASK>
__device__ __forceinline__ void wave_rollout_body(const MODEL& m, const TASK& tk, const RolloutArgs<wreal>& a, unsigned char* smem_raw, int cand, int lane,
                                                  wreal* cone_slab = nullptr) {
  const int nq = m.nq, nv = m.nv, nu = m.nu, nb = m.nbody, nr = tk.nr;
  const int P = a.P, H = a.H;
  const size_t N = (size_t)a.N;
  // ---- LDS carve
  wreal* lnodes; wreal* ltimes;
  const bool noisy = a.xfrc_scale > 0;
  TreeData tree;
  WaveData d;
  if constexpr (TREE) { lnodes = nullptr; d = wave_carve_tree(smem_raw, m, tk, P, ltimes, noisy, tree, CAPS, CAPC);
    if (cone_slab) { tree.ovf = cone_slab; tree.cap_tot = kTreeMaxConeTotal; } }
  else d = wave_carve(smem_raw, m, tk, P, lnodes, ltimes, /*nodes_in_lds=*/false, noisy);
  // The candidate's spline nodes stay in a.nodes ([node][actuator][candidate], HBM / L2): at most four of them per
  // actuator are read per step; volatile reads, because other lanes of this wavefront wrote them.
  const volatile wreal* gnodes = a.nodes + cand;
#define WNODE(j) gnodes[(size_t)(j) * N]
  // ---- candidate spline nodes (SamplingPlanner / CrossEntropyPlanner::AddNoiseToPolicy, as rollout_lane_kernel)
  for (int q = lane; q < P; q += 64) ltimes[q] = a.node_times[q];
  const int np = P * nu;
  if (a.noise.mode < 0) {
    // the caller's splines are already in place
  } else {
    const int gi = a.noise.candidate_offset + cand;
    wreal std = a.noise.std0;
    if (a.noise.mode == 0 && a.noise.std1 > 0) {
      if (bernoulli_uniform(a.noise.seed, (uint32_t)gi, a.noise.iteration) < WL(0.2)) std = a.noise.std1;
    }
    const bool noised = gi != a.noise.nominal_candidate;
    for (int j0 = 2 * lane; j0 < np; j0 += 128) {
      double z[2];  // the Philox / Box-Muller pair is generated in double whatever the working type
      gaussian_pair(a.noise.seed, (uint32_t)gi, (uint32_t)(j0 >> 1), a.noise.iteration, z);
      for (int e = 0; e < 2; e++) {
        const int j = j0 + e;
        if (j < np) {
          const int k = j % nu;
          const wreal lo = m.actuator_ctrlrange[2 * k], hi = m.actuator_ctrlrange[2 * k + 1];
          wreal v = a.nominal[j];
          if (noised) {
            wreal sigma;
            if (a.noise.mode == 0) sigma = WL(0.5) * (hi - lo) * std;
            else {
              const wreal fl = gi < a.noise.explore_count ? a.noise.std0 : a.noise.std1;
              const wreal s = sqrt(a.noise.param_variance[j]);
              sigma = s > fl ? s : fl;
            }
            v = clampv(v + sigma * (wreal)z[e], lo, hi);
          }
          a.nodes[(size_t)j * N + cand] = v;
        }
      }
    }
  }
  __threadfence();  // the nodes this wavefront wrote are read back (by other lanes) through L2
  // ---- initial condition (Planner::SetState)
  for (int i = lane; i < nq; i += 64) d.qpos[i] = tk.blob[i];
  for (int i = lane; i < nv; i += 64) d.qvel[i] = tk.blob[nq + i];
  if (lane < nu) d.ctrl[lane] = 0;
  if (lane < 4) d.counters[lane] = 0;
  if (noisy) for (int i = lane; i < 6 * nb; i += 64) d.xfrc[i] = 0;  // (the reference inherits the pooled mjData's forces)
  wreal time = tk.blob[tk.off_time];
  WSYNC();

  const int ds = nq + nv;
  wreal total = 0;
  bool failed = false;
  int fail_info = 0;  // diagnostics of a failed rollout: warning bits << 8 | step << 16 (failure stays "non-zero = failed")
  for (int t = 0; t < H; t++) {
    const bool last = t == H - 1;
    bool bad = false;
    // ================= policy: TimeSpline::Sample + Clamp, one lane per actuator
    if (!last) {
      wreal u = 0;
      if (lane < nu) {
        const int k = lane;
        int up = 0;
        while (up < P && ltimes[up] <= time) up++;
        if (up == P || up == 0) {
          u = WNODE((up == 0 ? 0 : P - 1) * nu + k);
        } else {
          const int lo = up - 1;
          const wreal tl = ltimes[lo], tu = ltimes[up];
          const wreal p0 = WNODE(lo * nu + k), p1 = WNODE(up * nu + k);
          if (a.interp == 0) u = p0;
          else {
            const wreal s = (time - tl) / (tu - tl);
            if (a.interp == 1) u = p0 * (1 - s) + p1 * s;
            else {
              const wreal dt_mid = tu - tl, fwd = (p1 - p0) / dt_mid;
              wreal m0, m1;
              if (lo == 0) m0 = fwd;
              else m0 = WL(0.5) * (p1 - p0) / dt_mid + WL(0.5) * (p0 - WNODE((lo - 1) * nu + k)) / (tl - ltimes[lo - 1]);
              if (up == P - 1) m1 = fwd;
              else m1 = WL(0.5) * (WNODE((up + 1) * nu + k) - p1) / (ltimes[up + 1] - tu) + WL(0.5) * (p1 - p0) / dt_mid;
              const wreal s2 = s * s, s3 = s * s * s;
              const wreal c0 = 2 * s3 - 3 * s2 + 1, c1 = (s3 - 2 * s2 + s) * (tu - tl), c2 = -2 * s3 + 3 * s2, c3 = (s3 - s2) * (tu - tl);
              u = c0 * p0 + c1 * m0 + c2 * p1 + c3 * m1;
            }
          }
        }
        bad = is_bad(u);
        u = clampv(u, m.actuator_ctrlrange[2 * k], m.actuator_ctrlrange[2 * k + 1]);
        d.ctrl[k] = u;
      }
      // mj_checkPos / mj_checkVel
      for (int i = lane; i < nq; i += 64) bad |= is_bad(d.qpos[i]);
      for (int i = lane; i < nv; i += 64) bad |= is_bad(d.qvel[i]);
    }
    if (noisy && !last) {
      // Trajectory::NoisyRollout (trajectory.cc:147-155): xfrc_applied = decay * xfrc_applied + N(0, scale) on every body entry
      const int gi = a.noise.candidate_offset + cand;
      for (int j = lane; j < 3 * nb; j += 64) {
        double z[2];
        gaussian_pair(a.xfrc_seed, (uint32_t)gi, (uint32_t)(t * 3 * m.nbody_model + j), 0x58465243u, z);
        d.xfrc[2 * j] = (wreal)a.xfrc_decay * d.xfrc[2 * j] + (wreal)(a.xfrc_scale * z[0]);
        d.xfrc[2 * j + 1] = (wreal)a.xfrc_decay * d.xfrc[2 * j + 1] + (wreal)(a.xfrc_scale * z[1]);
      }
    }
    WSYNC();
    // ================= mj_forward
    bool bad_ctrl = false;
    long long* stamp = (tk.stamps && cand == 0 && t == tk.stamp_step) ? tk.stamps : nullptr;
    WSTAMP(0);
    if constexpr (TREE) wt_forward<NMAX>(m, tk, d, tree, lane, bad_ctrl, stamp, /*have_warm=*/t > 0);
    else wf_forward<NMAX>(m, tk, d, lane, bad_ctrl, stamp, /*have_warm=*/t > 0);
    if (!last) for (int i = lane; i < nv; i += 64) bad |= is_bad(d.qacc[i]);  // mj_checkAcc
    if (!last) bad |= d.counters[2] != 0;  // CheckWarnings: contact / row cap overflow, indefinite Hessian (oracle odata_warning)
    bad = __any(bad);
    // ================= sensor stage: task residual and cost (task.cc:71-110)
    wr_residual(m, tk, d, time, lane);
    WSTAMP(12);
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
    // ================= record step t
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
    if (bad) { failed = true; fail_info = (d.counters[2] << 8) | (t << 16); break; }  // Trajectory::Rollout returns at the first warning (trajectory.cc:169-173); `bad` is wave-uniform
    total += cost;
    if (last) break;
    WSTAMP(13);
    if constexpr (RK4) {
      if (m.integrator == 1) {
        const bool warned = wave_rk4_step<NMAX, TREE>(m, tk, d, tree, lane, time, /*have_warm=*/t > 0);
        // data->site_xpos / xpos after mj_step are the LAST stage's: that is what Trajectory::Rollout copies into the trace
        // (trajectory.cc:165), so the row recorded above from the first stage is replaced
        if (lane < 3 * tk.ntrace) {
          const int ts = tk.trace_site[lane / 3];
          a.trace[((size_t)cand * H + t) * 3 * tk.ntrace + lane] = ts >= 0 ? d.site_xpos[3 * ts + lane % 3] : d.xpos[3 * (-1 - ts) + lane % 3];
        }
        if (warned) { failed = true; fail_info = (d.counters[2] << 8) | (t << 16); break; }  // CheckWarnings after mj_step
        continue;
      }
    }
    // ================= mj_Euler + advance (qacc is kept as the next step's warm start)
    if (lane < nv) d.qacc_warm[lane] = d.qacc[lane];
    if constexpr (TREE) wt_euler<NMAX>(m, d, lane, time);
    else wf_euler<NMAX>(m, d, lane, time);
    WSTAMP(14);
  }
  if (lane == 0) {
    a.total_return[cand] = failed ? kMaxReturn : total / (wreal)(H > 1 ? H : 1);
    a.failure[cand] = failed ? (1 | fail_info) : 0;
  }
}

// generic models: one wavefront per workgroup, one candidate per wavefront, the model behind the kernel-argument pointers
// SMALL (Jacobian-free path only): the first pass with the short contact lists (12 frictionless + 16 cones: three times the
// candidates per CU); a candidate that overflows them is flagged and rolled out again by a second launch of the large-list
// instantiation with a.only_overflowed set -- the scheme of the registered-model kernel (tree_kernel.h) for the generic one.
template <int NMAX, bool TREE = false, bool RK4 = false, bool SMALL = false>
__global__ __launch_bounds__(64) WAVE_KERNEL_ATTR void rollout_wave_kernel(const WModel m, const WTask tk, const RolloutArgs<wreal> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (a.only_overflowed && !((a.failure[blockIdx.x] >> 8) & 32)) return;  // (wave-uniform)
  // The model and task structs stay in the kernel-argument segment (scalar loads). Staging the model allocation into
  // LDS was tried: no shorter step, fewer candidates per CU, and a run-time-rebased copy of this struct
  // ends up in the private segment -- every pointer fetch becomes a scratch load. (Registered models: tree_kernel.h.)
  wave_rollout_body<NMAX, TREE, SMALL ? kTreeMaxSimple : kTreeMaxSimpleBig, SMALL ? kTreeMaxCone : kTreeMaxConeBig, RK4>(m, tk, a, smem_raw, blockIdx.x, threadIdx.x);
}

} }  // namespace mjpcx::WAVE_NS
