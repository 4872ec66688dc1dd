// simt_kernel.h -- residuals and the rollout kernel of the lane-per-candidate contact path (see rollout_simt.h).
#pragma once

namespace mjpcx {

__device__ __forceinline__ double simt_ray_down(const WaveModel& m, const SimtData& d, const double* from) {
  double best = -1;
  for (int gi = 0; gi < m.nray_geom; gi++) {
    const int g = m.ray_geom[gi], type = m.geom_type[g];
    double p[3], R[9];
    simt_geom_pose(m, d, g, p, R);
    const double* s = m.geom_size + 3 * g;
    double x = -1;
    if (type == MJPCX_GEOM_PLANE) {
      const double n[3] = {R[2], R[5], R[8]};
      const double denom = -n[2];
      if (fabs(denom) >= kMinVal) {
        const double t = -((from[0] - p[0]) * n[0] + (from[1] - p[1]) * n[1] + (from[2] - p[2]) * n[2]) / denom;
        if (t >= 0) {
          const double hit[3] = {from[0] - p[0], from[1] - p[1], from[2] - t - p[2]};
          const double lx = R[0] * hit[0] + R[3] * hit[1] + R[6] * hit[2], ly = R[1] * hit[0] + R[4] * hit[1] + R[7] * hit[2];
          if ((s[0] <= 0 || fabs(lx) <= s[0]) && (s[1] <= 0 || fabs(ly) <= s[1])) x = t;
        }
      }
    } else if (type == MJPCX_GEOM_SPHERE) {
      const double o[3] = {from[0] - p[0], from[1] - p[1], from[2] - p[2]};
      const double b = -o[2], c = o[0] * o[0] + o[1] * o[1] + o[2] * o[2] - s[0] * s[0];
      const double disc = b * b - c;
      if (disc >= 0) {
        const double sq = sqrt(disc), t0 = -b - sq, t1 = -b + sq;
        x = t0 >= 0 ? t0 : (t1 >= 0 ? t1 : -1);
      }
    } else {
      double o[3], dl[3];
      const double rel[3] = {from[0] - p[0], from[1] - p[1], from[2] - p[2]};
      for (int k = 0; k < 3; k++) { o[k] = R[k] * rel[0] + R[3 + k] * rel[1] + R[6 + k] * rel[2]; dl[k] = -R[6 + k]; }
      double tmin = -1e300, tmax = 1e300;
      bool miss = false;
      for (int k = 0; k < 3; k++) {
        if (fabs(dl[k]) < kMinVal) { if (fabs(o[k]) > s[k]) miss = true; continue; }
        double ta = (-s[k] - o[k]) / dl[k], tb = (s[k] - o[k]) / dl[k];
        if (ta > tb) { const double tt = ta; ta = tb; tb = tt; }
        if (ta > tmin) tmin = ta;
        if (tb < tmax) tmax = tb;
      }
      if (!(miss || tmin > tmax || tmax < 0)) x = tmin >= 0 ? tmin : tmax;
    }
    if (x >= 0 && (best < 0 || x < best)) best = x;
  }
  return best;
}

// QuadrupedFlat::ResidualFn::Residual (quadruped.cc:33-226); state layout as documented in wave_residual.h
__device__ __forceinline__ void simt_quadruped(const WaveModel& m, const WaveTask& tk, const SimtData& d, double time) {
  const SimtLayout& o = d.o;
  const int* ri = (const int*)(tk.blob + tk.off_rint);
  const double* re = tk.blob + tk.off_rreal;
  const double* par = tk.blob + tk.off_param;
  const double* mocap = tk.blob + tk.off_mocap;
  const int mode = ri[0], torso = ri[1];
  const double kGaitPhase[5][4] = {{0, 0, 0, 0}, {0, 0.75, 0.5, 0.25}, {0, 0.5, 0.5, 0}, {0, 0.33, 0.33, 0.66}, {0, 0.4, 0.05, 0.35}};
  double fp[12];
  for (int f = 0; f < 4; f++) { double R[9]; simt_geom_pose(m, d, ri[4 + f], fp + 3 * f, R); }
  // subtree linear velocity of the torso (oracle o_subtree_linvel: ascending body order)
  double comvel[3] = {0, 0, 0};
  {
    double mom[3] = {0, 0, 0}, mass = 0;
    const unsigned long long mask = m.body_subtree_mask[torso];
    for (int i = torso; i < m.nbody; i++) {
      if (!((mask >> i) & 1ull)) continue;
      double cv[6], com[3], ip[3], lin[3];
      d.ld(cv, o.cvel, 6 * i, 6); d.ld(com, o.subtree_com, 3 * m.body_rootid[i], 3); d.ld(ip, o.xipos, 3 * i, 3);
      const double off[3] = {ip[0] - com[0], ip[1] - com[1], ip[2] - com[2]};
      cr3(lin, cv, off);
      for (int k = 0; k < 3; k++) mom[k] += m.body_mass[i] * (cv[3 + k] + lin[k]);
      mass += m.body_mass[i];
    }
    for (int k = 0; k < 3; k++) comvel[k] = mass > kMinVal ? mom[k] / mass : 0.0;
  }
  const int handstand = ri[10];
  const bool is_biped = mode == 1;
  double avg[3];
  if (is_biped) {
    const int a = handstand ? 0 : 1, b = handstand ? 2 : 3;
    for (int k = 0; k < 3; k++) avg[k] = 0.5 * (fp[3 * a + k] + fp[3 * b + k]);
  } else {
    for (int k = 0; k < 3; k++) avg[k] = 0.25 * (fp[3 + k] + fp[9 + k] + fp[k] + fp[6 + k]);
  }
  const double height_goal = is_biped ? 0.6 : 0.25;
  double xmat[9], xq[4], tpos[3], head[3], compos[3];
  d.ld(xmat, o.xmat, 9 * torso, 9); d.ld(xq, o.xquat, 4 * torso, 4); d.ld(tpos, o.xipos, 3 * torso, 3);
  d.ld(head, o.site_xpos, 3 * ri[2], 3); d.ld(compos, o.subtree_com, 3 * torso, 3);
  const double* goal = mocap + 7 * ri[3];
  int c = 0;
  auto R = [&](int i) -> double& { return d.at(o.residual, i); };
  if (mode != 4) {
    if (is_biped) R(c++) = xmat[6] - (handstand ? -1 : 1);
    else R(c++) = xmat[8] - 1;
    R(c++) = 0; R(c++) = 0;
  } else {
    const double ft = time - re[0];
    const double jump_time = re[22], flight_time = re[18], land_time = re[24], crouch_time = re[20];
    double angle = 0, tt = ft;
    if (tt >= jump_time + flight_time + land_time) angle = 2 * kQPi;
    else if (tt >= crouch_time && tt < jump_time) { tt -= crouch_time; angle = 0.5 * re[28] * tt * tt + re[27] * tt; }
    else if (tt >= jump_time && tt < jump_time + flight_time) { tt -= jump_time; angle = kQPi / 2 + re[26] * tt; }
    else if (tt >= jump_time + flight_time) { tt -= jump_time + flight_time; angle = 1.75 * kQPi + re[26] * tt - 0.5 * re[29] * tt * tt; }
    const double axis[3] = {0, ri[9] ? 1.0 : -1.0, 0};
    double q[4], quat[4], qd[4];
    aa2quat(q, axis, angle);
    q_mul(quat, re + 9, q);
    const double qn[4] = {quat[0], -quat[1], -quat[2], -quat[3]};
    q_mul(qd, qn, xq);
    double ax[3] = {qd[1], qd[2], qd[3]};
    const double sin_a_2 = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    if (sin_a_2 > kMinVal) for (int k = 0; k < 3; k++) ax[k] /= sin_a_2;
    double speed = 2 * atan2(sin_a_2, qd[0]);
    if (speed > kQPi) speed -= 2 * kQPi;
    for (int k = 0; k < 3; k++) R(c++) = ax[k] * speed;
  }
  if (mode == 3) R(c++) = 0;
  else if (mode == 4) {
    const double jump_time = re[22], flight_time = re[18], land_time = re[24], ground = re[8];
    double hgt, tt = time - re[0];
    if (tt >= jump_time + flight_time + land_time) hgt = 0.25 + ground;
    else {
      double hh = 0;
      if (tt < jump_time) hh = 0.25 + tt * re[23] + 0.5 * tt * tt * re[19];
      else if (tt >= jump_time && tt < jump_time + flight_time) { tt -= jump_time; hh = 0.5 + re[17] * tt - 0.5 * 9.81 * tt * tt; }
      else if (tt >= jump_time + flight_time) { tt -= jump_time + flight_time; hh = 0.5 - re[17] * tt + 0.5 * re[25] * tt * tt; }
      hgt = hh + ground;
    }
    R(c++) = tpos[2] - hgt;
  } else R(c++) = (tpos[2] - avg[2]) - height_goal;
  double target[3];
  if (mode == 2) {
    const double t = time - re[0];
    const double* position = re + 1; const double* heading = re + 4;
    const double speed = re[6], angvel = re[7];
    if (fabs(angvel) < 0.01) {
      double fwd[2] = {heading[0], heading[1]};
      const double n = sqrt(fwd[0] * fwd[0] + fwd[1] * fwd[1]);
      if (n > kMinVal) { fwd[0] /= n; fwd[1] /= n; } else { fwd[0] = 1; fwd[1] = 0; }
      target[0] = position[0] + heading[0] + t * speed * fwd[0];
      target[1] = position[1] + heading[1] + t * speed * fwd[1];
    } else {
      double sn, cs;
      sincos(t * angvel, &sn, &cs);
      target[0] = cs * heading[0] - sn * heading[1] + position[0];
      target[1] = sn * heading[0] + cs * heading[1] + position[1];
    }
    target[2] = 0;
  } else { target[0] = goal[0]; target[1] = goal[1]; target[2] = goal[2]; }
  R(c++) = head[0] - target[0];
  R(c++) = head[1] - target[1];
  R(c++) = mode == 3 ? 2 * (head[2] - target[2]) : 0;
  {
    const int gait = is_biped ? 2 : ri[8];
    const double phase = re[13] + (time - re[14]) * re[15];
    for (int foot = 0; foot < 4; foot++) {
      const double step = par[ri[11]] * wr_step_height(phase, 2 * kQPi * kGaitPhase[gait][foot], par[ri[12]]);
      double out = 0;
      const bool front_hand = !handstand && (foot == 0 || foot == 2), back_hand = handstand && (foot == 1 || foot == 3);
      if (!(is_biped && (front_hand || back_hand))) {
        double query[3] = {fp[3 * foot], fp[3 * foot + 1], fp[3 * foot + 2]};
        if (mode == 3) {
          double v[2] = {goal[0] - fp[3 * foot], goal[1] - fp[3 * foot + 1]};
          const double n = sqrt(v[0] * v[0] + v[1] * v[1]);
          if (n > kMinVal) { v[0] /= n; v[1] /= n; } else { v[0] = 1; v[1] = 0; }
          query[0] += 0.15 * v[0]; query[1] += 0.15 * v[1];
        }
        const double from[3] = {query[0], query[1], query[2] + 0.5};
        const double ground = query[2] + 0.5 - simt_ray_down(m, d, from);
        double diff = fp[3 * foot + 2] - (ground + 0.02 + step);
        if (mode == 3) diff = diff < 0 ? diff : 0;
        out = step ? diff : 0;
      }
      R(c++) = out;
    }
  }
  const double fall_time = sqrt(2 * height_goal / 9.81);
  R(c++) = compos[0] + comvel[0] * fall_time - avg[0];
  R(c++) = compos[1] + comvel[1] * fall_time - avg[1];
  for (int i = 0; i < m.nu; i++) R(c + i) = 2e-2 * d.at(o.actuator_force, i);
  c += m.nu;
  const double* home = m.key_qpos + (size_t)m.nq * ri[15];
  for (int i = 0; i < m.nu; i++) R(c + i) = d.at(o.qpos, 7 + i) - home[7 + i];
  if (mode == 4) {
    const double ft = time - re[0];
    if (ft < re[20]) {
      const double* crouch = m.key_qpos + (size_t)m.nq * ri[16];
      for (int i = 0; i < m.nu; i++) R(c + i) = d.at(o.qpos, 7 + i) - crouch[7 + i];
    } else if (ft >= re[20] && ft < re[22] + re[18]) {
      for (int i = 0; i < m.nu; i++) R(c + i) = 0;
    }
  }
  const double gain[3] = {2, 1, 1};
  for (int foot = 0; foot < 4; foot++) for (int j = 0; j < 3; j++) R(c + 3 * foot + j) *= gain[j];
  if (is_biped) {
    const double arm = par[ri[13]];
    const int base = handstand ? 6 : 0;
    for (int i = 0; i < 6; i++) R(c + base + i) *= arm;
  }
  c += m.nu;
  double th[2] = {xmat[0], xmat[3]};
  if (is_biped) { const int hs = handstand ? 1 : -1; th[0] = hs * xmat[2]; th[1] = hs * xmat[5]; }
  const double n = sqrt(th[0] * th[0] + th[1] * th[1]);
  if (n < kMinVal) { th[0] = 1; th[1] = 0; } else { th[0] /= n; th[1] /= n; }
  double sn, cs;
  sincos(par[ri[14]], &sn, &cs);
  R(c++) = th[0] - cs;
  R(c++) = th[1] - sn;
  for (int k = 0; k < 3; k++) R(c++) = comvel[k];
}

__global__ __launch_bounds__(64) void rollout_simt_kernel(const WaveModel m, const WaveTask tk, const RolloutArgs<double> a, const SimtLayout lay,
                                                          double* scratch) {
  const int lane = threadIdx.x;
  const int cand = blockIdx.x * 64 + lane;
  const bool live = cand < a.N;
  const int ci = live ? cand : a.N - 1;  // dead lanes integrate a duplicate and never store
  const size_t N = (size_t)a.N;
  SimtData d;
  d.o = lay;
#if MJPCX_SIMT_PRIVATE
  double priv[kSimtPrivateDoubles];
  d.slab = priv;
#else
  d.slab = scratch + (size_t)blockIdx.x * lay.total * 64 + lane;
#endif
  const SimtLayout& o = d.o;
  const int nq = m.nq, nv = m.nv, nu = m.nu, nr = tk.nr, P = a.P, H = a.H, np = P * nu;
  // ---- candidate spline nodes
  if (a.noise.mode < 0) {
    for (int j = 0; j < np; j++) d.at(o.nodes, j) = a.nodes[(size_t)j * N + ci];
  } else {
    const int gi = a.noise.candidate_offset + ci;
    double std = a.noise.std0;
    if (a.noise.mode == 0 && a.noise.std1 > 0) {
      if (bernoulli_uniform(a.noise.seed, (uint32_t)gi, a.noise.iteration) < 0.2) std = a.noise.std1;
    }
    const bool noised = gi != a.noise.nominal_candidate;
    for (int j0 = 0; j0 < np; j0 += 2) {
      double z[2];
      gaussian_pair(a.noise.seed, (uint32_t)gi, (uint32_t)(j0 >> 1), a.noise.iteration, z);
      for (int e = 0; e < 2; e++) {
        const int j = j0 + e;
        if (j < np) {
          const int k = j % nu;
          const double lo = m.actuator_ctrlrange[2 * k], hi = m.actuator_ctrlrange[2 * k + 1];
          double v = a.nominal[j];
          if (noised) {
            double sigma;
            if (a.noise.mode == 0) sigma = 0.5 * (hi - lo) * std;
            else {
              const double fl = gi < a.noise.explore_count ? a.noise.std0 : a.noise.std1;
              const double s = sqrt(a.noise.param_variance[j]);
              sigma = s > fl ? s : fl;
            }
            v = clampv(v + sigma * z[e], lo, hi);
          }
          d.at(o.nodes, j) = v;
          if (live) a.nodes[(size_t)j * N + cand] = v;
        }
      }
    }
  }
  for (int i = 0; i < nq; i++) d.at(o.qpos, i) = tk.blob[i];
  for (int i = 0; i < nv; i++) d.at(o.qvel, i) = tk.blob[nq + i];
  for (int i = 0; i < nu; i++) d.at(o.ctrl, i) = 0;
  double time = tk.blob[tk.off_time];
  const int ds = nq + nv;
  double total = 0;
  bool failed = false;
  for (int t = 0; t < H; t++) {
    const bool last = t == H - 1;
    bool bad = false;
    if (!last) {
      int up = 0;
      while (up < P && a.node_times[up] <= time) up++;
      for (int k = 0; k < nu; k++) {
        double u;
        if (up == P || up == 0) u = d.at(o.nodes, (up == 0 ? 0 : P - 1) * nu + k);
        else {
          const int lo = up - 1;
          const double tl = a.node_times[lo], tu = a.node_times[up];
          const double p0 = d.at(o.nodes, lo * nu + k), p1 = d.at(o.nodes, up * nu + k);
          if (a.interp == 0) u = p0;
          else {
            const double s = (time - tl) / (tu - tl);
            if (a.interp == 1) u = p0 * (1 - s) + p1 * s;
            else {
              const double dt_mid = tu - tl, fwd = (p1 - p0) / dt_mid;
              double m0, m1;
              if (lo == 0) m0 = fwd;
              else m0 = 0.5 * (p1 - p0) / dt_mid + 0.5 * (p0 - d.at(o.nodes, (lo - 1) * nu + k)) / (tl - a.node_times[lo - 1]);
              if (up == P - 1) m1 = fwd;
              else m1 = 0.5 * (d.at(o.nodes, (up + 1) * nu + k) - p1) / (a.node_times[up + 1] - tu) + 0.5 * (p1 - p0) / dt_mid;
              const double s2 = s * s, s3 = s * s * s;
              const double c0 = 2 * s3 - 3 * s2 + 1, c1 = (s3 - 2 * s2 + s) * (tu - tl), c2 = -2 * s3 + 3 * s2, c3 = (s3 - s2) * (tu - tl);
              u = c0 * p0 + c1 * m0 + c2 * p1 + c3 * m1;
            }
          }
        }
        bad |= is_bad(u);
        d.at(o.ctrl, k) = clampv(u, m.actuator_ctrlrange[2 * k], m.actuator_ctrlrange[2 * k + 1]);
      }
      for (int i = 0; i < nq; i++) bad |= is_bad(d.at(o.qpos, i));
      for (int i = 0; i < nv; i++) bad |= is_bad(d.at(o.qvel, i));
    }
    int ncon = 0, nefc = 0, warning = 0;
    simt_forward(m, tk, d, ncon, nefc, warning, /*have_warm=*/t > 0);
    if (!last) for (int i = 0; i < nv; i++) bad |= is_bad(d.at(o.qacc, i));
    // residual + cost
    if (tk.residual_id == MJPCX_RESIDUAL_QUADRUPED_FLAT) simt_quadruped(m, tk, d, time);
    else for (int i = 0; i < nr; i++) d.at(o.residual, i) = 0;
    double cost = 0;
    {
      int off = 0;
      for (int k = 0; k < tk.nterm; k++) {
        const int dim = tk.dim_norm_residual[k];
        double x[16];
        double term;
        if (dim <= 16) { d.ld(x, o.residual, off, dim); term = w_norm_value(x, dim, tk.norm[k], tk.blob[tk.off_normp + k], tk.blob[tk.off_normq + k]); }
        else term = 0;  // terms wider than 16 entries are not used by the built tasks
        cost += tk.blob[tk.off_weight + k] * term;
        off += dim;
      }
      const double risk = tk.blob[tk.off_risk];
      if (!(fabs(risk) < 1.0e-6)) cost = (exp(risk * cost) - 1.0) / risk;
    }
    if (live && !failed) {
      for (int i = 0; i < ds; i++) a.states[((size_t)t * ds + i) * N + cand] = i < nq ? d.at(o.qpos, i) : d.at(o.qvel, i - nq);
      for (int k = 0; k < nu; k++) a.actions[((size_t)t * nu + k) * N + cand] = d.at(o.ctrl, k);
      for (int i = 0; i < nr; i++) a.residual[((size_t)t * nr + i) * N + cand] = d.at(o.residual, i);
      for (int k = 0; k < tk.ntrace; k++)
        for (int c = 0; c < 3; c++) a.trace[((size_t)t * 3 * tk.ntrace + 3 * k + c) * N + cand] = d.at(o.site_xpos, 3 * tk.trace_site[k] + c);
      a.times[(size_t)t * N + cand] = time;
      if (!bad) a.costs[(size_t)t * N + cand] = cost;
    }
    if (bad) failed = true;
    total += cost;
    if (last) break;
    for (int i = 0; i < nv; i++) d.at(o.qacc_warm, i) = d.at(o.qacc, i);
    simt_euler(m, d, time);
  }
  if (live) {
    a.total_return[cand] = failed ? kMaxReturn : total / (double)(H > 1 ? H : 1);
    a.failure[cand] = failed ? 1 : 0;
  }
}

}  // namespace mjpcx
