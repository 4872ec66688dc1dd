// wave_residual.h -- device ResidualFn::Residual overrides of the wave-kernel models.
//   MJPCX_RESIDUAL_QUADRUPED_FLAT : QuadrupedFlat::ResidualFn::Residual (mjpc/tasks/quadruped/quadruped.cc:33-226,
//   helpers :609-720, constants quadruped.h:69-140). Layout of the frozen ResidualFn state in the plan blob:
//     int : [0] current_mode_ [1] torso_body_id_ [2] head_site_id_ [3] goal_mocap_id_ [4..7] foot_geom_id_ (FL HL FR HR)
//           [8] gait [9] flip_dir [10] biped_type [11] amplitude_param_id_ [12] duty_param_id_ [13] arm_posture_param_id_
//           [14] ParameterIndex("Heading") [15] key id "home" [16] key id "crouch"
//     real: [0] mode_start_time_ [1..3] position_ [4..5] heading_ [6] speed_ [7] angvel_ [8] ground_ [9..12] orientation_
//           [13] phase_start_ [14] phase_start_time_ [15] phase_velocity_ [16] gravity_ [17] jump_vel_ [18] flight_time_
//           [19] jump_acc_ [20] crouch_time_ [21] leap_time_ [22] jump_time_ [23] crouch_vel_ [24] land_time_
//           [25] land_acc_ [26] flight_rot_vel_ [27] jump_rot_vel_ [28] jump_rot_acc_ [29] land_rot_acc_

namespace mjpcx { namespace WAVE_NS {

constexpr wreal kQPi = WL(3.14159265358979323846);

// mj_ray straight down against the geoms of group 0 (plane / sphere / box), as Ground() asks (utilities.cc:556-574)
template <class MODEL>
__device__ __forceinline__ wreal wr_ray_down(const MODEL& m, const WaveData& d, const wreal* from) {
  wreal best = -1;
  for (int gi = 0; gi < m.nray_geom; gi++) {
    const int g = m.ray_geom[gi];
    const int type = m.geom_type[g];
    wreal p[3], R[9];
    wf_geom_pose(m, d, g, p, R);
    const wreal* s = m.geom_size + 3 * g;
    wreal x = -1;
    if (type == MJPCX_GEOM_PLANE) {
      const wreal n[3] = {R[2], R[5], R[8]};
      const wreal denom = -n[2];
      if (fabs(denom) >= kMinVal) {
        const wreal t = -((from[0] - p[0]) * n[0] + (from[1] - p[1]) * n[1] + (from[2] - p[2]) * n[2]) / denom;
        if (t >= 0) {
          const wreal hit[3] = {from[0] - p[0], from[1] - p[1], from[2] - t - p[2]};
          const wreal lx = R[0] * hit[0] + R[3] * hit[1] + R[6] * hit[2], ly = R[1] * hit[0] + R[4] * hit[1] + R[7] * hit[2];
          if ((s[0] <= 0 || fabs(lx) <= s[0]) && (s[1] <= 0 || fabs(ly) <= s[1])) x = t;
        }
      }
    } else if (type == MJPCX_GEOM_SPHERE) {
      const wreal o[3] = {from[0] - p[0], from[1] - p[1], from[2] - p[2]};
      const wreal b = -o[2], c = o[0] * o[0] + o[1] * o[1] + o[2] * o[2] - s[0] * s[0];
      const wreal disc = b * b - c;
      if (disc >= 0) {
        const wreal sq = sqrt(disc), t0 = -b - sq, t1 = -b + sq;
        x = t0 >= 0 ? t0 : (t1 >= 0 ? t1 : -1);
      }
    } else {
      wreal o[3], dl[3];
      const wreal rel[3] = {from[0] - p[0], from[1] - p[1], from[2] - p[2]};
      for (int k = 0; k < 3; k++) { o[k] = R[k] * rel[0] + R[3 + k] * rel[1] + R[6 + k] * rel[2]; dl[k] = -R[6 + k]; }
      wreal tmin = -WL(1e300), tmax = WL(1e300);
      bool miss = false;
      for (int k = 0; k < 3; k++) {
        if (fabs(dl[k]) < kMinVal) { if (fabs(o[k]) > s[k]) miss = true; continue; }
        wreal ta = (-s[k] - o[k]) / dl[k], tb = (s[k] - o[k]) / dl[k];
        if (ta > tb) { const wreal tt = ta; ta = tb; tb = tt; }
        if (ta > tmin) tmin = ta;
        if (tb < tmax) tmax = tb;
      }
      if (!(miss || tmin > tmax || tmax < 0)) x = tmin >= 0 ? tmin : tmax;
    }
    if (x >= 0 && (best < 0 || x < best)) best = x;
  }
  return best;
}
__device__ __forceinline__ wreal wr_step_height(wreal time, wreal footphase, wreal duty_ratio) {
  wreal angle = fmod(time + kQPi - footphase, 2 * kQPi) - kQPi;
  wreal value = 0;
  if (duty_ratio < 1) {
    angle *= WL(0.5) / (1 - duty_ratio);
    value = cos(clampv(angle, -kQPi / 2, kQPi / 2));
  }
  return fabs(value) < WL(1e-6) ? WL(0.0) : value;
}

template <class MODEL, class TASK>
__device__ __forceinline__ void wr_quadruped(const MODEL& m, const TASK& tk, WaveData& d, wreal time, int lane) {
  const int* ri = (const int*)(tk.blob + tk.off_rint);
  const wreal* re = tk.blob + tk.off_rreal;
  const wreal* par = tk.blob + tk.off_param;
  const wreal* mocap = tk.blob + tk.off_mocap;
  const int mode = ri[0], torso = ri[1];
  const wreal kGaitPhase[5][4] = {{0, 0, 0, 0}, {0, WL(0.75), WL(0.5), WL(0.25)}, {0, WL(0.5), WL(0.5), 0}, {0, WL(0.33), WL(0.33), WL(0.66)}, {0, WL(0.4), WL(0.05), WL(0.35)}};
  // feet positions (geom_xpos of the four foot geoms)
  if (lane < 4) {
    wreal p[3], R[9];
    wf_geom_pose(m, d, ri[4 + lane], p, R);
    for (int k = 0; k < 3; k++) d.foot_xpos[3 * lane + k] = p[k];
  }
  // subtree linear velocity of the torso (sensors torso_subtreelinvel and "torso_angmom"): lanes = bodies
  {
    wreal mom[3] = {0, 0, 0};
    const unsigned long long mask = m.body_subtree_mask[torso];
    if (lane < m.nbody && ((mask >> lane) & 1ull)) {
      const int i = lane;
      const wreal* cv = d.cvel + 6 * i;
      const wreal* com = d.subtree_com + 3 * m.body_rootid[i];
      const wreal off[3] = {d.xipos[3 * i] - com[0], d.xipos[3 * i + 1] - com[1], d.xipos[3 * i + 2] - com[2]};
      wreal lin[3];
      cr3(lin, cv, off);
      for (int k = 0; k < 3; k++) mom[k] = m.body_mass[i] * (cv[3 + k] + lin[k]);
    }
    for (int k = 0; k < 3; k++) mom[k] = wave_sum(mom[k]);
    const wreal mass = m.body_subtreemass[torso];
    if (lane == 0) for (int k = 0; k < 3; k++) d.subtree_linvel[k] = mass > kMinVal ? mom[k] / mass : WL(0.0);
  }
  WSYNC();
  const wreal* fp = d.foot_xpos;  // FL HL FR HR
  const int handstand = ri[10];
  const bool is_biped = mode == 1;
  wreal avg[3];
  if (is_biped) {
    const int a = handstand ? 0 : 1, b = handstand ? 2 : 3;
    for (int k = 0; k < 3; k++) avg[k] = WL(0.5) * (fp[3 * a + k] + fp[3 * b + k]);
  } else {
    for (int k = 0; k < 3; k++) avg[k] = WL(0.25) * (fp[3 + k] + fp[9 + k] + fp[k] + fp[6 + k]);
  }
  const wreal height_goal = is_biped ? WL(0.6) : WL(0.25);
  wreal* r = d.residual;
  // ---------- Gait: one lane per foot (ray cast against the terrain)
  if (lane < 4) {
    const int foot = lane;
    const int gait = is_biped ? 2 : ri[8];
    const wreal phase = re[13] + (time - re[14]) * re[15];
    const wreal step = par[ri[11]] * wr_step_height(phase, 2 * kQPi * kGaitPhase[gait][foot], par[ri[12]]);
    wreal out = 0;
    const bool front_hand = !handstand && (foot == 0 || foot == 2), back_hand = handstand && (foot == 1 || foot == 3);
    if (!(is_biped && (front_hand || back_hand))) {
      wreal query[3] = {fp[3 * foot], fp[3 * foot + 1], fp[3 * foot + 2]};
      if (mode == 3) {
        const wreal* goal = mocap + 7 * ri[3];
        wreal v[2] = {goal[0] - fp[3 * foot], goal[1] - fp[3 * foot + 1]};
        const wreal n = sqrt(v[0] * v[0] + v[1] * v[1]);
        if (n > kMinVal) { v[0] /= n; v[1] /= n; } else { v[0] = 1; v[1] = 0; }
        query[0] += WL(0.15) * v[0]; query[1] += WL(0.15) * v[1];
      }
      const wreal from[3] = {query[0], query[1], query[2] + WL(0.5)};
      const wreal ground = query[2] + WL(0.5) - wr_ray_down(m, d, from);
      wreal diff = fp[3 * foot + 2] - (ground + WL(0.02) + step);
      if (mode == 3) diff = diff < 0 ? diff : 0;
      out = step ? diff : 0;
    }
    r[7 + foot] = out;
  }
  // ---------- everything else: lane 4 (cheap, serial)
  if (lane == 4) {
    const wreal* xmat = d.xmat + 9 * torso;
    const wreal* goal = mocap + 7 * ri[3];
    int c = 0;
    if (mode != 4) {
      if (is_biped) r[c++] = xmat[6] - (handstand ? -1 : 1);
      else r[c++] = xmat[8] - 1;
      r[c++] = 0; r[c++] = 0;
    } else {
      // FlipQuat
      const wreal ft = time - re[0];
      const wreal jump_time = re[22], flight_time = re[18], land_time = re[24], crouch_time = re[20];
      wreal angle = 0, tt = ft;
      if (tt >= jump_time + flight_time + land_time) angle = 2 * kQPi;
      else if (tt >= crouch_time && tt < jump_time) { tt -= crouch_time; angle = WL(0.5) * re[28] * tt * tt + re[27] * tt; }
      else if (tt >= jump_time && tt < jump_time + flight_time) { tt -= jump_time; angle = kQPi / 2 + re[26] * tt; }
      else if (tt >= jump_time + flight_time) { tt -= jump_time + flight_time; angle = WL(1.75) * kQPi + re[26] * tt - WL(0.5) * re[29] * tt * tt; }
      const wreal axis[3] = {0, ri[9] ? WL(1.0) : -WL(1.0), 0};
      wreal q[4], quat[4];
      aa2quat(q, axis, angle);
      q_mul(quat, re + 9, q);
      // mju_subQuat(res, torso_xquat, quat)
      const wreal* qa = d.xquat + 4 * torso;
      const wreal qn[4] = {quat[0], -quat[1], -quat[2], -quat[3]};
      wreal qd[4];
      q_mul(qd, qn, qa);
      wreal ax[3] = {qd[1], qd[2], qd[3]};
      const wreal sin_a_2 = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
      if (sin_a_2 > kMinVal) for (int k = 0; k < 3; k++) ax[k] /= sin_a_2;
      wreal speed = 2 * atan2(sin_a_2, qd[0]);
      if (speed > kQPi) speed -= 2 * kQPi;
      for (int k = 0; k < 3; k++) r[c++] = ax[k] * speed;
    }
    // Height
    const wreal* tpos = d.xipos + 3 * torso;
    if (mode == 3) r[c++] = 0;
    else if (mode == 4) {
      const wreal ft = time - re[0];
      const wreal jump_time = re[22], flight_time = re[18], land_time = re[24], ground = re[8];
      wreal hgt, tt = ft;
      if (tt >= jump_time + flight_time + land_time) hgt = WL(0.25) + ground;
      else {
        wreal hh = 0;
        if (tt < jump_time) hh = WL(0.25) + tt * re[23] + WL(0.5) * tt * tt * re[19];
        else if (tt >= jump_time && tt < jump_time + flight_time) { tt -= jump_time; hh = WL(0.5) + re[17] * tt - WL(0.5) * WL(9.81) * tt * tt; }
        else if (tt >= jump_time + flight_time) { tt -= jump_time + flight_time; hh = WL(0.5) - re[17] * tt + WL(0.5) * re[25] * tt * tt; }
        hgt = hh + ground;
      }
      r[c++] = tpos[2] - hgt;
    } else r[c++] = (tpos[2] - avg[2]) - height_goal;
    // Position
    const wreal* head = d.site_xpos + 3 * ri[2];
    wreal target[3];
    if (mode == 2) {
      const wreal t = time - re[0];
      const wreal* position = re + 1; const wreal* heading = re + 4;
      const wreal speed = re[6], angvel = re[7];
      if (fabs(angvel) < WL(0.01)) {
        wreal fwd[2] = {heading[0], heading[1]};
        const wreal n = sqrt(fwd[0] * fwd[0] + fwd[1] * fwd[1]);
        if (n > kMinVal) { fwd[0] /= n; fwd[1] /= n; } else { fwd[0] = 1; fwd[1] = 0; }
        target[0] = position[0] + heading[0] + t * speed * fwd[0];
        target[1] = position[1] + heading[1] + t * speed * fwd[1];
      } else {
        wreal sn, cs;
        w_sincos(t * angvel, &sn, &cs);
        target[0] = cs * heading[0] - sn * heading[1] + position[0];
        target[1] = sn * heading[0] + cs * heading[1] + position[1];
      }
      target[2] = 0;
    } else { target[0] = goal[0]; target[1] = goal[1]; target[2] = goal[2]; }
    r[c++] = head[0] - target[0];
    r[c++] = head[1] - target[1];
    r[c++] = mode == 3 ? 2 * (head[2] - target[2]) : 0;
    c += 4;  // Gait: lanes 0..3
    // Balance
    const wreal* compos = d.subtree_com + 3 * torso;
    const wreal* comvel = d.subtree_linvel;
    const wreal fall_time = sqrt(2 * height_goal / WL(9.81));
    r[c++] = compos[0] + comvel[0] * fall_time - avg[0];
    r[c++] = compos[1] + comvel[1] * fall_time - avg[1];
    // Effort
    for (int i = 0; i < m.nu; i++) r[c + i] = WL(2e-2) * d.actuator_force[i];
    c += m.nu;
    // Posture
    const wreal* home = m.key_qpos + (size_t)m.nq * ri[15];
    for (int i = 0; i < m.nu; i++) r[c + i] = d.qpos[7 + i] - home[7 + i];
    if (mode == 4) {
      const wreal ft = time - re[0];
      if (ft < re[20]) {
        const wreal* crouch = m.key_qpos + (size_t)m.nq * ri[16];
        for (int i = 0; i < m.nu; i++) r[c + i] = d.qpos[7 + i] - crouch[7 + i];
      } else if (ft >= re[20] && ft < re[22] + re[18]) {
        for (int i = 0; i < m.nu; i++) r[c + i] = 0;
      }
    }
    const wreal gain[3] = {2, 1, 1};
    for (int foot = 0; foot < 4; foot++) for (int j = 0; j < 3; j++) r[c + 3 * foot + j] *= gain[j];
    if (is_biped) {
      const wreal arm = par[ri[13]];
      const int base = handstand ? 6 : 0;
      for (int i = 0; i < 6; i++) r[c + base + i] *= arm;
    }
    c += m.nu;
    // Yaw
    wreal th[2] = {xmat[0], xmat[3]};
    if (is_biped) { const int hs = handstand ? 1 : -1; th[0] = hs * xmat[2]; th[1] = hs * xmat[5]; }
    const wreal n = sqrt(th[0] * th[0] + th[1] * th[1]);
    if (n < kMinVal) { th[0] = 1; th[1] = 0; } else { th[0] /= n; th[1] /= n; }
    wreal sn, cs;
    w_sincos(par[ri[14]], &sn, &cs);
    r[c++] = th[0] - cs;
    r[c++] = th[1] - sn;
    for (int k = 0; k < 3; k++) r[c++] = comvel[k];
  }
  WSYNC();
}

// ---- mjpc::humanoid::Tracking::ResidualFn::Residual (mjpc/tasks/humanoid/tracking/tracking.cc:94-216; oracle/humanoid.inc).
// residual_int = [first key, last key, 16 tracking-site ids, 16 mocap ids], residual_real = [reference_time].
// Lanes 0..15: one marker each (interpolated keyframe position, site position and linear velocity); the averages are
// wave reductions in the oracle's summation order (serial over the 16 markers).
template <class MODEL, class TASK>
__device__ __forceinline__ void wr_humanoid_track(const MODEL& m, const TASK& tk, WaveData& d, wreal time, int lane) {
  const int* ri = reinterpret_cast<const int*>(tk.blob + tk.off_rint);
  const wreal ref_time = tk.blob[tk.off_rreal];
  const int start = ri[0], last = ri[1];
  const wreal kFps = WL(30.0);
  const wreal index = (time - ref_time) * kFps + start;
  const wreal clamped = index < WL(0.0) ? WL(0.0) : (index > (wreal)last ? (wreal)last : index);
  const int k0 = (int)floor(clamped);
  const int k1 = k0 + 1 < last ? k0 + 1 : last;
  const wreal w1 = clamped - k0, w0 = WL(1.0) - w1;
  wreal* r = d.residual;
  const int nj = m.nv - 6;
  if (lane < nj) r[lane] = d.qvel[6 + lane];
  if (lane < m.nu) r[nj + lane] = d.ctrl[lane];
  wreal mp[3] = {0, 0, 0}, sp[3] = {0, 0, 0}, dv[3] = {0, 0, 0};
  if (lane < 16) {
    const int site = ri[2 + lane], mc = ri[18 + lane];
    const wreal* key0 = m.key_mpos + ((size_t)m.nmocap * k0 + mc) * 3;
    const wreal* key1 = m.key_mpos + ((size_t)m.nmocap * k1 + mc) * 3;
    const int body = m.site_bodyid[site];
    const wreal* cv = d.cvel + 6 * body;
    const wreal* com = d.subtree_com + 3 * m.body_rootid[body];
    wreal off[3], lin[3];
    for (int k = 0; k < 3; k++) { sp[k] = d.site_xpos[3 * site + k]; off[k] = sp[k] - com[k]; }
    cr3(lin, cv, off);
    for (int k = 0; k < 3; k++) {
      wreal v = key0[k] * w0;
      v += key1[k] * w1;
      mp[k] = v;
      dv[k] = (key1[k] - key0[k]) * kFps - (cv[3 + k] + lin[k]);
    }
  }
  // averages: serial sums over the 16 markers, as the reference accumulates them
  wreal am[3] = {0, 0, 0}, as[3] = {0, 0, 0};
  for (int b = 0; b < 16; b++)
    for (int k = 0; k < 3; k++) { am[k] += __shfl(mp[k], b, 64); as[k] += __shfl(sp[k], b, 64); }
  for (int k = 0; k < 3; k++) { am[k] *= WL(1.0) / 16; as[k] *= WL(1.0) / 16; }
  const int c = nj + m.nu;
  if (lane < 3) r[c + lane] = am[lane] - as[lane];
  if (lane < 16)
    for (int k = 0; k < 3; k++) {
      r[c + 3 + 3 * lane + k] = (mp[k] - am[k]) - (sp[k] - as[k]);
      r[c + 51 + 3 * lane + k] = dv[k];
    }
  WSYNC();
}

template <class MODEL, class TASK>
__device__ __forceinline__ void wr_residual(const MODEL& m, const TASK& tk, WaveData& d, wreal time, int lane) {
  if (tk.residual_id == MJPCX_RESIDUAL_QUADRUPED_FLAT) { wr_quadruped(m, tk, d, time, lane); return; }
  if (tk.residual_id == MJPCX_RESIDUAL_HUMANOID_TRACK) { wr_humanoid_track(m, tk, d, time, lane); return; }
  for (int i = lane; i < tk.nr; i += 64) d.residual[i] = 0;
  WSYNC();
}

} }  // namespace mjpcx::WAVE_NS
