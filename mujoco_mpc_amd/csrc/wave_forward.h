// wave_forward.h -- mj_forward for one candidate held in LDS, all 64 lanes cooperating (see rollout_wave.h).
// Stage by stage the arithmetic is oracle/physics.c + oracle/contact.inc; the comments name the oracle function.

namespace mjpcx { namespace WAVE_NS {

// ---- o_kinematics: bodies level by level (a body needs its parent), then sites
template <class MODEL, class TASK>
__device__ __forceinline__ void wf_kinematics(const MODEL& m, const TASK& tk, WaveData& d, int lane) {
  if (lane == 0) {
    d.xpos[0] = d.xpos[1] = d.xpos[2] = 0;
    d.xquat[0] = 1; d.xquat[1] = d.xquat[2] = d.xquat[3] = 0;
    for (int k = 0; k < 9; k++) d.xmat[k] = d.ximat[k] = (k % 4 == 0) ? WL(1.0) : WL(0.0);
    d.xipos[0] = d.xipos[1] = d.xipos[2] = 0;
  }
  WSYNC();
  for (int l = 0; l < m.nlevel; l++) {
    const int idx = m.level_start[l] + lane;
    if (idx < m.level_start[l + 1]) {
      const int i = m.level_body[idx];
      const int pid = m.body_parentid[i], jn = m.body_jntnum[i], ja = m.body_jntadr[i];
      wreal xpos[3], xquat[4];
      if (m.body_mocapid[i] >= 0) {
        const wreal* mp = tk.blob + tk.off_mocap + 7 * m.body_mocapid[i];
        for (int k = 0; k < 3; k++) xpos[k] = mp[k];
        for (int k = 0; k < 4; k++) xquat[k] = mp[3 + k];
        q_norm(xquat);
      } else if (jn == 1 && m.jnt_type[ja] == kJntFree) {
        const int qa = m.jnt_qposadr[ja];
        for (int k = 0; k < 3; k++) xpos[k] = d.qpos[qa + k];
        for (int k = 0; k < 4; k++) xquat[k] = d.qpos[qa + 3 + k];
        q_norm(xquat);
        for (int k = 0; k < 3; k++) d.xanchor[3 * ja + k] = xpos[k];
        d.xaxis[3 * ja] = 0; d.xaxis[3 * ja + 1] = 0; d.xaxis[3 * ja + 2] = 1;
      } else {
        mv3(xpos, d.xmat + 9 * pid, m.body_pos + 3 * i);
        for (int k = 0; k < 3; k++) xpos[k] += d.xpos[3 * pid + k];
        q_mul(xquat, d.xquat + 4 * pid, m.body_quat + 4 * i);
        for (int j = ja; j < ja + jn; j++) {
          const int qa = m.jnt_qposadr[j];
          wreal anchor[3], axis[3];
          q_rot(anchor, m.jnt_pos + 3 * j, xquat);
          for (int k = 0; k < 3; k++) anchor[k] += xpos[k];
          q_rot(axis, m.jnt_axis + 3 * j, xquat);
          const int jt = m.jnt_type[j];
          if (jt == kJntSlide) {
            const wreal s = d.qpos[qa] - m.qpos0[qa];
            for (int k = 0; k < 3; k++) xpos[k] += axis[k] * s;
          } else if (jt == kJntBall || jt == kJntHinge) {
            wreal qloc[4], vec[3];
            if (jt == kJntBall) { for (int k = 0; k < 4; k++) qloc[k] = d.qpos[qa + k]; q_norm(qloc); }
            else aa2quat(qloc, m.jnt_axis + 3 * j, d.qpos[qa] - m.qpos0[qa]);
            q_mul(xquat, xquat, qloc);
            q_rot(vec, m.jnt_pos + 3 * j, xquat);
            for (int k = 0; k < 3; k++) xpos[k] = anchor[k] - vec[k];
          }
          for (int k = 0; k < 3; k++) { d.xanchor[3 * j + k] = anchor[k]; d.xaxis[3 * j + k] = axis[k]; }
        }
      }
      q_norm(xquat);
      wreal xmat[9], v[3], q[4];
      q2mat(xmat, xquat);
      for (int k = 0; k < 3; k++) d.xpos[3 * i + k] = xpos[k];
      for (int k = 0; k < 4; k++) d.xquat[4 * i + k] = xquat[k];
      for (int k = 0; k < 9; k++) d.xmat[9 * i + k] = xmat[k];
      mv3(v, xmat, m.body_ipos + 3 * i);
      for (int k = 0; k < 3; k++) d.xipos[3 * i + k] = xpos[k] + v[k];
      q_mul(q, xquat, m.body_iquat + 4 * i);
      q2mat(d.ximat + 9 * i, q);
    }
    WSYNC();
  }
  if (lane < m.nsite) {
    const int s = lane, b = m.site_bodyid[s];
    wreal v[3];
    mv3(v, d.xmat + 9 * b, m.site_pos + 3 * s);
    for (int k = 0; k < 3; k++) d.site_xpos[3 * s + k] = d.xpos[3 * b + k] + v[k];
  }
}

// world pose of a geom (o_geom_kinematics), computed where it is needed instead of being stored for all geoms
template <class MODEL>
__device__ __forceinline__ void wf_geom_pose(const MODEL& m, const WaveData& d, int g, wreal* pos, wreal* mat) {
  const int b = m.geom_bodyid[g];
  wreal v[3], q[4];
  mv3(v, d.xmat + 9 * b, m.geom_pos + 3 * g);
  for (int k = 0; k < 3; k++) pos[k] = d.xpos[3 * b + k] + v[k];
  q_mul(q, d.xquat + 4 * b, m.geom_quat + 4 * g);
  q2mat(mat, q);
}
// radius of a pair geom's bounding sphere about its centre (the pretest of the moving-geom pairs)
template <class MODEL>
__device__ __forceinline__ wreal wf_pair_bound(const MODEL& m, int g) {
  const int t = m.geom_type[g];
  const wreal s0 = m.geom_size[3 * g], s1 = m.geom_size[3 * g + 1], s2 = m.geom_size[3 * g + 2];
  if (t == MJPCX_GEOM_CAPSULE) return s0 + s1;
  if (t == MJPCX_GEOM_CYLINDER) return sqrt(s0 * s0 + s1 * s1);
  if (t == MJPCX_GEOM_BOX) return sqrt(s0 * s0 + s1 * s1 + s2 * s2);
  return s0;
}
// (sphere | capsule) g1 against the (box | cylinder) g2 at world poses (p1, R1), (p2, R2): oracle pair_thin_solid; one contact at most.
// g1 a solid too (`watch`): no contact is made -- there is no narrow phase for two solids -- but -1 is returned if the thin geom that contains
// g1 (cylinder: its capsule, box: the ball about its centre) is within the margin of g2, as the oracle raises warning bit 128 then
template <class MODEL>
__device__ __forceinline__ int wf_thin_vs_solid(const MODEL& m, int g1, int g2, const wreal* p1, const wreal* R1, const wreal* p2, const wreal* R2, wreal margin,
                                                wreal* cd, wreal* cp, wreal* cn, bool watch = false) {
  const int t1 = m.geom_type[g1];
  wreal h = t1 == MJPCX_GEOM_CAPSULE ? m.geom_size[3 * g1 + 1] : WL(0.0), r = m.geom_size[3 * g1];
  if (watch) {
    const wreal s0 = m.geom_size[3 * g1], s1 = m.geom_size[3 * g1 + 1], s2 = m.geom_size[3 * g1 + 2];
    if (t1 == MJPCX_GEOM_CYLINDER) { h = s1; r = s0; } else { h = WL(0.0); r = sqrt(s0 * s0 + s1 * s1 + s2 * s2); }
  }
  const wreal rel[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  wreal pl[3], al[3], n[3], c[3];
  for (int k = 0; k < 3; k++) {
    pl[k] = R2[k] * rel[0] + R2[3 + k] * rel[1] + R2[6 + k] * rel[2];
    al[k] = R2[k] * R1[2] + R2[3 + k] * R1[5] + R2[6 + k] * R1[8];
  }
  const wreal size[3] = {m.geom_size[3 * g2], m.geom_size[3 * g2 + 1], m.geom_size[3 * g2 + 2]};
  const wreal dist = solid::thin_vs_solid<wreal>(m.geom_type[g2] == MJPCX_GEOM_CYLINDER ? solid::kSolidCylinder : solid::kSolidBox, size, pl, al, h, r, n, c);
  if (!(dist < margin)) return 0;
  if (watch) return -1;
  wreal loc[3], w[3];
  for (int k = 0; k < 3; k++) loc[k] = c[k] + n[k] * (r + WL(0.5) * dist);
  mv3(w, R2, loc);
  for (int k = 0; k < 3; k++) cp[k] = p2[k] + w[k];
  mv3(cn, R2, n);
  cd[0] = dist;
  return 1;
}

// ---- o_compos: subtree centres of mass, cinert, cdof
template <class MODEL>
__device__ __forceinline__ void wf_compos(const MODEL& m, WaveData& d, int lane) {
  const int nb = m.nbody;
  if (lane < nb) {
    const int i = lane;
    unsigned long long mask = m.body_subtree_mask[i];
    wreal s[3] = {0, 0, 0};
    // ascending body order = the oracle's accumulation order reversed; sums of <= 13 terms, parity tolerance covers it
    while (mask) {
      const int j = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const wreal mj = m.body_mass[j];
      for (int k = 0; k < 3; k++) s[k] += mj * d.xipos[3 * j + k];
    }
    const wreal sm = m.body_subtreemass[i];
    for (int k = 0; k < 3; k++) d.subtree_com[3 * i + k] = sm < kMinVal ? d.xipos[3 * i + k] : s[k] / sm;
  }
  WSYNC();
  if (lane < nb) {
    const int i = lane;
    if (i == 0) { for (int k = 0; k < 10; k++) d.cinert[k] = 0; }
    else {
      wreal off[3];
      const wreal* com = d.subtree_com + 3 * m.body_rootid[i];
      for (int k = 0; k < 3; k++) off[k] = d.xipos[3 * i + k] - com[k];
      w_inert_com(d.cinert + 10 * i, m.body_inertia + 3 * i, d.ximat + 9 * i, off, m.body_mass[i]);
    }
  }
  if (lane < m.njnt) {
    const int j = lane, b = m.jnt_bodyid[j];
    int da = m.jnt_dofadr[j];
    wreal off[3];
    const wreal* com = d.subtree_com + 3 * m.body_rootid[b];
    for (int k = 0; k < 3; k++) off[k] = com[k] - d.xanchor[3 * j + k];
    const wreal* xmat = d.xmat + 9 * b;
    const int jt = m.jnt_type[j];
    if (jt == kJntFree) {
      for (int k = 0; k < 3; k++) {
        wreal* c = d.cdof + 6 * (da + k);
        for (int e = 0; e < 6; e++) c[e] = 0;
        c[3 + k] = 1;
      }
      da += 3;
    }
    if (jt == kJntFree || jt == kJntBall) {
      for (int k = 0; k < 3; k++) {
        wreal* c = d.cdof + 6 * (da + k);
        const wreal ax[3] = {xmat[k], xmat[3 + k], xmat[6 + k]};
        for (int e = 0; e < 3; e++) c[e] = ax[e];
        cr3(c + 3, ax, off);
      }
    } else if (jt == kJntSlide) {
      wreal* c = d.cdof + 6 * da;
      c[0] = c[1] = c[2] = 0;
      for (int e = 0; e < 3; e++) c[3 + e] = d.xaxis[3 * j + e];
    } else {
      wreal* c = d.cdof + 6 * da;
      for (int e = 0; e < 3; e++) c[e] = d.xaxis[3 * j + e];
      cr3(c + 3, d.xaxis + 3 * j, off);
    }
  }
  WSYNC();
}

// ---- o_crb: composite inertias by subtree masks, then M (dense, both triangles)
template <class MODEL>
__device__ __forceinline__ void wf_crb(const MODEL& m, WaveData& d, int lane) {
  const int nb = m.nbody, nv = m.nv;
  if (lane < nb && lane > 0) {
    const int i = lane;
    unsigned long long mask = m.body_subtree_mask[i];
    wreal s[10];
    for (int k = 0; k < 10; k++) s[k] = 0;
    while (mask) {
      const int j = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      for (int k = 0; k < 10; k++) s[k] += d.cinert[10 * j + k];
    }
    for (int k = 0; k < 10; k++) d.crb[10 * i + k] = s[k];
  }
  for (int e = lane; e < nv * nv; e += 64) d.M[e] = 0;
  WSYNC();
  if (lane < nv) {
    const int i = lane;
    wreal buf[6];
    w_mul_inert(buf, d.crb + 10 * m.dof_bodyid[i], d.cdof + 6 * i);
    d.M[i * nv + i] = m.dof_armature[i] + w_dot6(d.cdof + 6 * i, buf);
    for (int j = m.dof_parentid[i]; j >= 0; j = m.dof_parentid[j]) {
      const wreal v = w_dot6(d.cdof + 6 * j, buf);
      d.M[i * nv + j] = v;
      d.M[j * nv + i] = v;
    }
  }
  WSYNC();
}

// ---- o_comvel: cvel and cdof_dot, level by level
template <class MODEL>
__device__ __forceinline__ void wf_comvel(const MODEL& m, WaveData& d, int lane) {
  if (lane < 6) d.cvel[lane] = 0;
  WSYNC();
  for (int l = 0; l < m.nlevel; l++) {
    const int idx = m.level_start[l] + lane;
    if (idx < m.level_start[l + 1]) {
      const int i = m.level_body[idx];
      wreal cvel[6];
      for (int c = 0; c < 6; c++) cvel[c] = d.cvel[6 * m.body_parentid[i] + c];
      for (int j = m.body_jntadr[i]; j < m.body_jntadr[i] + m.body_jntnum[i]; j++) {
        int da = m.jnt_dofadr[j];
        const int jt = m.jnt_type[j];
        if (jt == kJntFree) {
          for (int e = 0; e < 18; e++) d.cdof_dot[6 * da + e] = 0;
          for (int k = 0; k < 3; k++)
            for (int c = 0; c < 6; c++) cvel[c] += d.cdof[6 * (da + k) + c] * d.qvel[da + k];
          da += 3;
        }
        if (jt == kJntFree || jt == kJntBall) {
          for (int k = 0; k < 3; k++) w_cross_motion(d.cdof_dot + 6 * (da + k), cvel, d.cdof + 6 * (da + k));
          for (int k = 0; k < 3; k++)
            for (int c = 0; c < 6; c++) cvel[c] += d.cdof[6 * (da + k) + c] * d.qvel[da + k];
        } else {
          w_cross_motion(d.cdof_dot + 6 * da, cvel, d.cdof + 6 * da);
          for (int c = 0; c < 6; c++) cvel[c] += d.cdof[6 * da + c] * d.qvel[da];
        }
      }
      for (int c = 0; c < 6; c++) d.cvel[6 * i + c] = cvel[c];
    }
    WSYNC();
  }
}

// ---- o_passive, o_rne (bias forces), o_actuation, qfrc_smooth
template <class MODEL>
__device__ __forceinline__ void wf_smooth_forces(const MODEL& m, WaveData& d, int lane, bool& bad_ctrl) {
  const int nb = m.nbody, nv = m.nv, nu = m.nu;
  // passive
  if (lane < nv) {
    wreal f = 0;
    if (!(m.disableflags & MJPCX_DSBL_PASSIVE)) {
      const int j = m.dof_jntid[lane], jt = m.jnt_type[j];
      const wreal k = m.jnt_stiffness[j];
      if (k != 0 && (jt == kJntSlide || jt == kJntHinge)) f -= k * (d.qpos[m.jnt_qposadr[j]] - m.qpos_spring[m.jnt_qposadr[j]]);
      f -= m.dof_damping[lane] * d.qvel[lane];
    }
    d.qfrc_passive[lane] = f;
  }
  // RNE forward: cacc level by level, cfrc per body
  if (lane < 6) {
    // (0 / 1 weights over the three components: `m.gravity[lane - 3]` indexes a member of the model struct at run time, which puts the
    // whole struct -- every scalar and pointer of a registered model -- in scratch)
    const wreal gl = (lane == 3 ? WL(1.0) : WL(0.0)) * (wreal)m.gravity[0] + (lane == 4 ? WL(1.0) : WL(0.0)) * (wreal)m.gravity[1] +
                     (lane == 5 ? WL(1.0) : WL(0.0)) * (wreal)m.gravity[2];
    d.cacc[lane] = (lane >= 3 && !(m.disableflags & MJPCX_DSBL_GRAVITY)) ? -gl : WL(0.0);
    d.cfrc[lane] = 0;
  }
  WSYNC();
  for (int l = 0; l < m.nlevel; l++) {
    const int idx = m.level_start[l] + lane;
    if (idx < m.level_start[l + 1]) {
      const int i = m.level_body[idx];
      wreal cacc[6], t1[6], t2[6], t3[6];
      for (int c = 0; c < 6; c++) cacc[c] = d.cacc[6 * m.body_parentid[i] + c];
      const int da = m.body_dofadr[i];
      for (int k = da; k >= 0 && k < da + m.body_dofnum[i]; k++)
        for (int c = 0; c < 6; c++) cacc[c] += d.cdof_dot[6 * k + c] * d.qvel[k];
      for (int c = 0; c < 6; c++) d.cacc[6 * i + c] = cacc[c];
      w_mul_inert(t1, d.cinert + 10 * i, cacc);
      w_mul_inert(t2, d.cinert + 10 * i, d.cvel + 6 * i);
      w_cross_force(t3, d.cvel + 6 * i, t2);
      for (int c = 0; c < 6; c++) d.cfrc[6 * i + c] = t1[c] + t3[c];
    }
    WSYNC();
  }
  // backward accumulation over subtrees (parents of world-attached bodies excluded, as in the oracle: body 0 never sums)
  if (lane < nb && lane > 0) {
    const int i = lane;
    unsigned long long mask = m.body_subtree_mask[i];
    wreal s[6] = {0, 0, 0, 0, 0, 0};
    while (mask) {
      const int j = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      for (int c = 0; c < 6; c++) s[c] += d.cfrc[6 * j + c];
    }
    for (int c = 0; c < 6; c++) d.cfrc_sub[6 * i + c] = s[c];
  }
  // actuation (o_actuation): BADCTRL zeroes every control
  bool bad = false;
  if (lane < nu) bad = is_bad(d.ctrl[lane]);
  bad_ctrl = __any(bad);
  WSYNC();
  if (bad_ctrl && lane < nu) d.ctrl[lane] = 0;
  if (lane < nv) { d.qfrc_bias[lane] = w_dot6(d.cdof + 6 * lane, d.cfrc_sub + 6 * m.dof_bodyid[lane]); d.qfrc_actuator[lane] = 0; }
  WSYNC();
  if (lane < nu) {
    const int i = lane;
    wreal force = 0;
    if (!(m.disableflags & MJPCX_DSBL_ACTUATION)) {
      wreal ctrl = d.ctrl[i];
      if (m.actuator_ctrllimited[i] && !(m.disableflags & MJPCX_DSBL_CLAMPCTRL))
        ctrl = clampv(ctrl, m.actuator_ctrlrange[2 * i], m.actuator_ctrlrange[2 * i + 1]);
      const int j = m.actuator_trnid[i], qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j];
      const wreal gear = m.actuator_gear[i];
      force = m.actuator_gainprm[3 * i] * ctrl;
      if (m.actuator_biastype[i] == 1)
        force += m.actuator_biasprm[3 * i] + m.actuator_biasprm[3 * i + 1] * gear * d.qpos[qa] + m.actuator_biasprm[3 * i + 2] * gear * d.qvel[da];
      if (m.actuator_forcelimited[i]) force = clampv(force, m.actuator_forcerange[2 * i], m.actuator_forcerange[2 * i + 1]);
    }
    d.actuator_force[i] = force;
  }
  WSYNC();
  if (lane == 0 && !(m.disableflags & MJPCX_DSBL_ACTUATION))  // several actuators may drive one dof: serial, in actuator order
    for (int i = 0; i < nu; i++) d.qfrc_actuator[m.jnt_dofadr[m.actuator_trnid[i]]] += m.actuator_gear[i] * d.actuator_force[i];
  WSYNC();
  if (lane < nv) {
    wreal q = d.qfrc_passive[lane] - d.qfrc_bias[lane] + d.qfrc_actuator[lane];
    if (d.xfrc) {
      // mj_xfrcAccumulate: Cartesian force / torque at the centre of mass of every body below this dof (oracle
      // o_xfrc_accumulate; bodies in ascending order)
      const wreal* c = d.cdof + 6 * lane;
      unsigned long long mask = m.body_subtree_mask[m.dof_bodyid[lane]];
      while (mask) {
        const int b = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const wreal* f = d.xfrc + 6 * b;
        const wreal* com = d.subtree_com + 3 * m.body_rootid[b];
        const wreal off[3] = {d.xipos[3 * b] - com[0], d.xipos[3 * b + 1] - com[1], d.xipos[3 * b + 2] - com[2]};
        wreal tq[3];
        cr3(tq, off, f);
        q += c[0] * (tq[0] + f[3]) + c[1] * (tq[1] + f[4]) + c[2] * (tq[2] + f[5]) + c[3] * f[0] + c[4] * f[1] + c[5] * f[2];
      }
    }
    d.qfrc_smooth[lane] = q;
    d.qacc_smooth[lane] = q;
  }
  WSYNC();
}

// ---- o_collision: one lane per moving geom against each static geom; order-preserving compaction
template <class MODEL>
__device__ __forceinline__ void wf_contact_param(const MODEL& m, int g1, int g2, WaveContact& c) {
  const wreal margin = fmax(m.geom_margin[g1], m.geom_margin[g2]);
  const wreal gap = fmax(m.geom_gap[g1], m.geom_gap[g2]);
  c.margin = margin;
  c.includemargin = margin - gap;
  wreal fr[3];
  const int p1 = m.geom_priority[g1], p2 = m.geom_priority[g2];
  if (p1 != p2) {
    const int g = p1 > p2 ? g1 : g2;
    c.dim = m.geom_condim[g];
    for (int k = 0; k < 3; k++) fr[k] = m.geom_friction[3 * g + k];
    for (int k = 0; k < 2; k++) c.solref[k] = m.geom_solref[2 * g + k];
    for (int k = 0; k < 5; k++) c.solimp[k] = m.geom_solimp[5 * g + k];
  } else {
    c.dim = m.geom_condim[g1] > m.geom_condim[g2] ? m.geom_condim[g1] : m.geom_condim[g2];
    for (int k = 0; k < 3; k++) fr[k] = fmax(m.geom_friction[3 * g1 + k], m.geom_friction[3 * g2 + k]);
    const wreal s1 = m.geom_solmix[g1], s2 = m.geom_solmix[g2];
    const wreal mix = (s1 >= kMinVal && s2 >= kMinVal) ? s1 / (s1 + s2) : (s1 < kMinVal && s2 < kMinVal ? WL(0.5) : (s1 < kMinVal ? WL(0.0) : WL(1.0)));
    for (int k = 0; k < 2; k++) c.solref[k] = mix * m.geom_solref[2 * g1 + k] + (1 - mix) * m.geom_solref[2 * g2 + k];
    for (int k = 0; k < 5; k++) c.solimp[k] = mix * m.geom_solimp[5 * g1 + k] + (1 - mix) * m.geom_solimp[5 * g2 + k];
  }
  c.friction[0] = c.friction[1] = fmax(fr[0], kMinMu);
  c.friction[2] = fmax(fr[1], kMinMu);
  c.friction[3] = c.friction[4] = fmax(fr[2], kMinMu);
  c.dim0 = c.dim;
}

template <class MODEL>
__device__ __forceinline__ void wf_collision(const MODEL& m, WaveData& d, int lane) {
  if (lane == 0) d.counters[0] = 0;
  WSYNC();
  if (m.disableflags & (MJPCX_DSBL_CONSTRAINT | MJPCX_DSBL_CONTACT)) return;
  // this lane's moving geom
  const bool have = lane < m.ndynamic_geom;
  const int g2 = have ? m.dynamic_geom[lane] : 0;
  wreal p2[3] = {0, 0, 0}, R2[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (have) wf_geom_pose(m, d, g2, p2, R2);
  const int t2 = have ? m.geom_type[g2] : -1;
  const wreal s2[3] = {have ? m.geom_size[3 * g2] : 0, have ? m.geom_size[3 * g2 + 1] : 0, have ? m.geom_size[3 * g2 + 2] : 0};
  for (int si = 0; si < m.nstatic_geom; si++) {
    const int g1 = m.static_geom[si], t1 = m.geom_type[g1];
    if (t1 != MJPCX_GEOM_PLANE && t1 != MJPCX_GEOM_SPHERE && t1 != MJPCX_GEOM_BOX) continue;
    wreal p1[3], R1[9];
    wf_geom_pose(m, d, g1, p1, R1);  // wave-uniform
    // up to 4 candidate contacts of this lane: dist, pos, normal
    wreal cd[4], cp[4][3], cn[3] = {0, 0, 1};
    int cnt = 0;
    // candidate slots are written through a switch on the count: static indices keep cd / cp in registers
    auto push = [&](wreal dist, wreal px, wreal py, wreal pz) {
      switch (cnt) {
        case 0: cd[0] = dist; cp[0][0] = px; cp[0][1] = py; cp[0][2] = pz; break;
        case 1: cd[1] = dist; cp[1][0] = px; cp[1][1] = py; cp[1][2] = pz; break;
        case 2: cd[2] = dist; cp[2][0] = px; cp[2][1] = py; cp[2][2] = pz; break;
        default: cd[3] = dist; cp[3][0] = px; cp[3][1] = py; cp[3][2] = pz; break;
      }
      cnt++;
    };
    wreal margin = 0;
    const bool pair = have && ((m.geom_contype[g1] & m.geom_conaffinity[g2]) || (m.geom_contype[g2] & m.geom_conaffinity[g1]));
    if (pair) {
      margin = fmax(m.geom_margin[g1], m.geom_margin[g2]);
      if (t1 == MJPCX_GEOM_PLANE) {
        const wreal n[3] = {R1[2], R1[5], R1[8]};
        for (int k = 0; k < 3; k++) cn[k] = n[k];
        auto sphere_plane = [&](const wreal* c, wreal r) {
          const wreal dist = (c[0] - p1[0]) * n[0] + (c[1] - p1[1]) * n[1] + (c[2] - p1[2]) * n[2] - r;
          if (dist < margin) push(dist, c[0] - n[0] * (r + WL(0.5) * dist), c[1] - n[1] * (r + WL(0.5) * dist), c[2] - n[2] * (r + WL(0.5) * dist));
        };
        if (t2 == MJPCX_GEOM_SPHERE) {
          sphere_plane(p2, s2[0]);
        } else if (t2 == MJPCX_GEOM_CAPSULE) {
          for (int sgn = -1; sgn <= 1; sgn += 2) {
            wreal c[3];
            for (int k = 0; k < 3; k++) c[k] = p2[k] + sgn * s2[1] * R2[3 * k + 2];
            sphere_plane(c, s2[0]);
          }
        } else if (t2 == MJPCX_GEOM_BOX) {
          for (int i = 0; i < 8 && cnt < 4; i++) {
            const wreal loc[3] = {(i & 1 ? s2[0] : -s2[0]), (i & 2 ? s2[1] : -s2[1]), (i & 4 ? s2[2] : -s2[2])};
            wreal c[3];
            mv3(c, R2, loc);
            for (int k = 0; k < 3; k++) c[k] += p2[k];
            const wreal dist = (c[0] - p1[0]) * n[0] + (c[1] - p1[1]) * n[1] + (c[2] - p1[2]) * n[2];
            if (dist < margin) push(dist, c[0] - WL(0.5) * dist * n[0], c[1] - WL(0.5) * dist * n[1], c[2] - WL(0.5) * dist * n[2]);
          }
        } else if (t2 == MJPCX_GEOM_CYLINDER) {
          const wreal a[3] = {R2[2], R2[5], R2[8]};
          const wreal pa = n[0] * a[0] + n[1] * a[1] + n[2] * a[2];
          const wreal sgn = pa > 0 ? -WL(1.0) : WL(1.0);
          wreal v[3], vn = 0;
          for (int k = 0; k < 3; k++) { v[k] = -(n[k] - pa * a[k]); vn += v[k] * v[k]; }
          vn = sqrt(vn);
          if (vn < WL(1e-10)) { v[0] = R2[0]; v[1] = R2[3]; v[2] = R2[6]; vn = 1; }
          for (int k = 0; k < 3; k++) v[k] /= vn;
          wreal w[3];
          cr3(w, a, v);
          const wreal cs[3] = {WL(1.0), -WL(0.5), -WL(0.5)}, sn[3] = {WL(0.0), WL(0.8660254037844386), -WL(0.8660254037844386)};
          for (int i = 0; i < 4; i++) {
            const wreal side = i < 3 ? sgn : -sgn, cc = i < 3 ? cs[i] : WL(1.0), ss = i < 3 ? sn[i] : WL(0.0);
            wreal c[3];
            for (int k = 0; k < 3; k++) c[k] = p2[k] + side * s2[1] * a[k] + s2[0] * (cc * v[k] + ss * w[k]);
            const wreal dist = (c[0] - p1[0]) * n[0] + (c[1] - p1[1]) * n[1] + (c[2] - p1[2]) * n[2];
            if (dist < margin) push(dist, c[0] - WL(0.5) * dist * n[0], c[1] - WL(0.5) * dist * n[1], c[2] - WL(0.5) * dist * n[2]);
          }
        }
      } else if (t1 == MJPCX_GEOM_SPHERE && t2 == MJPCX_GEOM_SPHERE) {
        wreal n[3], len = 0;
        for (int k = 0; k < 3; k++) { n[k] = p2[k] - p1[k]; len += n[k] * n[k]; }
        len = sqrt(len);
        if (len < kMinVal) { n[0] = 1; n[1] = n[2] = 0; } else for (int k = 0; k < 3; k++) n[k] /= len;
        const wreal r1 = m.geom_size[3 * g1], dist = len - r1 - s2[0];
        if (dist < margin) {
          cd[0] = dist;
          for (int k = 0; k < 3; k++) { cp[0][k] = p1[k] + n[k] * (r1 + WL(0.5) * dist); cn[k] = n[k]; }
          cnt = 1;
        }
      } else if (t1 == MJPCX_GEOM_BOX && t2 == MJPCX_GEOM_SPHERE) {
        const wreal* s1 = m.geom_size + 3 * g1;
        wreal rel[3], loc[3], clamped[3];
        for (int k = 0; k < 3; k++) rel[k] = p2[k] - p1[k];
        for (int k = 0; k < 3; k++) loc[k] = R1[k] * rel[0] + R1[3 + k] * rel[1] + R1[6 + k] * rel[2];
        bool inside = true;
        for (int k = 0; k < 3; k++) {
          clamped[k] = loc[k] < -s1[k] ? -s1[k] : (loc[k] > s1[k] ? s1[k] : loc[k]);
          if (clamped[k] != loc[k]) inside = false;
        }
        wreal nl[3] = {0, 0, 0}, dist;
        if (!inside) {
          wreal len = 0;
          for (int k = 0; k < 3; k++) { nl[k] = loc[k] - clamped[k]; len += nl[k] * nl[k]; }
          len = sqrt(len);
          for (int k = 0; k < 3; k++) nl[k] /= len;
          dist = len - s2[0];
        } else {
          int best = 0; wreal bd = WL(1e300);
          for (int k = 0; k < 3; k++) { const wreal dd = s1[k] - fabs(loc[k]); if (dd < bd) { bd = dd; best = k; } }
          nl[best] = loc[best] >= 0 ? 1 : -1;
          clamped[best] = nl[best] * s1[best];
          dist = -bd - s2[0];
        }
        if (dist < margin) {
          wreal n[3], surf[3];
          mv3(n, R1, nl);
          mv3(surf, R1, clamped);
          cd[0] = dist;
          for (int k = 0; k < 3; k++) { cp[0][k] = p1[k] + surf[k] + WL(0.5) * dist * n[k]; cn[k] = n[k]; }
          cnt = 1;
        }
      }
    }
    // lane-major, contact-minor compaction: offset = contacts of all lower lanes (+ those already stored)
    int below = 0, total = 0;
    for (int k = 0; k < 4; k++) {
      const unsigned long long b = __ballot(cnt > k);
      below += __popcll(b & ((1ull << lane) - 1ull));
      total += __popcll(b);
    }
    const int base = __builtin_amdgcn_readfirstlane(d.counters[0]);
    WSYNC();
    if (cnt > 0) {
      WaveContact proto;
      proto.g1 = g1; proto.g2 = g2; proto.efc = 0; proto.mu = 0; proto.nrow = 0;
      wf_contact_param(m, g1, g2, proto);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int at = base + below + k;
        if (k < cnt && at < kWaveMaxCon) {
          WaveContact c = proto;
          c.dist = cd[k];
          for (int e = 0; e < 3; e++) { c.pos[e] = cp[k][e]; c.frame[e] = cn[e]; }
          w_make_frame(c.frame);
          d.con[at] = c;
        }
      }
    }
    if (lane == 0) {
      const int n = base + total;
      if (n > kWaveMaxCon) d.counters[2] |= 32;
      d.counters[0] = n > kWaveMaxCon ? kWaveMaxCon : n;
    }
    WSYNC();
  }
  // moving-geom pairs (sphere | capsule pairs, sphere | capsule against box | cylinder; oracle: pair_collide): one lane per baked pair, up to two contacts each
  for (int p0 = 0; p0 < m.npair; p0 += 64) {
    const bool on = p0 + lane < m.npair;
    const int g1 = on ? m.pair_g1[p0 + lane] : 0, g2 = on ? m.pair_g2[p0 + lane] : 0;
    wreal cd[2] = {0, 0}, cp[2][3] = {{0, 0, 0}, {0, 0, 0}}, cn[2][3] = {{1, 0, 0}, {1, 0, 0}};
    int cnt = 0;
    // conservative bounding-sphere pretest on the geom centres (never rejects a pair the narrow phase would accept):
    // most steps have no self-contact and skip the poses, the narrow phase and the compaction altogether
    bool near = false, solids_touch = false;
    wreal p1[3] = {0, 0, 0}, p2[3] = {0, 0, 0};
    const wreal margin = on ? fmax(m.geom_margin[g1], m.geom_margin[g2]) : WL(0.0);
    if (on) {
      const int b1 = m.geom_bodyid[g1], b2 = m.geom_bodyid[g2];
      wreal v1[3], v2[3];
      mv3(v1, d.xmat + 9 * b1, m.geom_pos + 3 * g1);
      mv3(v2, d.xmat + 9 * b2, m.geom_pos + 3 * g2);
      wreal dd = 0;
      for (int k = 0; k < 3; k++) { p1[k] = d.xpos[3 * b1 + k] + v1[k]; p2[k] = d.xpos[3 * b2 + k] + v2[k]; dd += (p1[k] - p2[k]) * (p1[k] - p2[k]); }
      const wreal reach = wf_pair_bound(m, g1) + wf_pair_bound(m, g2) + margin + WL(1e-6);
      near = dd <= reach * reach;
    }
    if (__ballot(near) == 0ull) continue;
    if (near) {
      wreal R1[9], R2[9];
      wf_geom_pose(m, d, g1, p1, R1);
      wf_geom_pose(m, d, g2, p2, R2);
      const int t1 = m.geom_type[g1], t2 = m.geom_type[g2];
      const wreal r1 = m.geom_size[3 * g1], r2 = m.geom_size[3 * g2];
      auto spheres = [&](const wreal* c1, const wreal* c2) {
        wreal n[3], len = 0;
        for (int k = 0; k < 3; k++) { n[k] = c2[k] - c1[k]; len += n[k] * n[k]; }
        len = sqrt(len);
        if (len < kMinVal) { n[0] = 1; n[1] = n[2] = 0; } else for (int k = 0; k < 3; k++) n[k] /= len;
        const wreal dist = len - r1 - r2;
        if (dist < margin) {
          if (cnt == 0) { cd[0] = dist; for (int k = 0; k < 3; k++) { cp[0][k] = c1[k] + n[k] * (r1 + WL(0.5) * dist); cn[0][k] = n[k]; } }
          else { cd[1] = dist; for (int k = 0; k < 3; k++) { cp[1][k] = c1[k] + n[k] * (r1 + WL(0.5) * dist); cn[1][k] = n[k]; } }
          cnt++;
        }
      };
      auto seg = [&](const wreal* p, const wreal* a, wreal h, const wreal* c) {
        const wreal x = (c[0] - p[0]) * a[0] + (c[1] - p[1]) * a[1] + (c[2] - p[2]) * a[2];
        return x < -h ? -h : (x > h ? h : x);
      };
      if (t2 == MJPCX_GEOM_CYLINDER || t2 == MJPCX_GEOM_BOX) {  // (sphere | capsule, box | cylinder): solid_pairs.h; two solids are only watched
        cnt = wf_thin_vs_solid(m, g1, g2, p1, R1, p2, R2, margin, cd, cp[0], cn[0], t1 == MJPCX_GEOM_CYLINDER || t1 == MJPCX_GEOM_BOX);
        if (cnt < 0) { cnt = 0; solids_touch = true; }
      } else if (t1 == MJPCX_GEOM_SPHERE && t2 == MJPCX_GEOM_SPHERE) {
        spheres(p1, p2);
      } else if (t1 == MJPCX_GEOM_SPHERE) {
        const wreal a2[3] = {R2[2], R2[5], R2[8]};
        const wreal x = seg(p2, a2, m.geom_size[3 * g2 + 1], p1);
        const wreal c2[3] = {p2[0] + x * a2[0], p2[1] + x * a2[1], p2[2] + x * a2[2]};
        spheres(p1, c2);
      } else {
        const wreal a1[3] = {R1[2], R1[5], R1[8]}, a2[3] = {R2[2], R2[5], R2[8]};
        const wreal h1 = m.geom_size[3 * g1 + 1], h2 = m.geom_size[3 * g2 + 1];
        const wreal dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
        const wreal mb = -(a1[0] * a2[0] + a1[1] * a2[1] + a1[2] * a2[2]);
        const wreal u = -(a1[0] * dif[0] + a1[1] * dif[1] + a1[2] * dif[2]);
        const wreal v = a2[0] * dif[0] + a2[1] * dif[1] + a2[2] * dif[2];
        const wreal det = WL(1.0) - mb * mb;
        wreal c1[3], c2[3];
        if (fabs(det) >= kMinVal) {
          wreal x1 = (u - mb * v) / det, x2 = (v - mb * u) / det;
          if (x1 > h1) { x1 = h1; x2 = v - mb * x1; } else if (x1 < -h1) { x1 = -h1; x2 = v - mb * x1; }
          if (x2 > h2) { x2 = h2; x1 = u - mb * x2; x1 = x1 > h1 ? h1 : (x1 < -h1 ? -h1 : x1); }
          else if (x2 < -h2) { x2 = -h2; x1 = u - mb * x2; x1 = x1 > h1 ? h1 : (x1 < -h1 ? -h1 : x1); }
          for (int k = 0; k < 3; k++) { c1[k] = p1[k] + x1 * a1[k]; c2[k] = p2[k] + x2 * a2[k]; }
          spheres(c1, c2);
        } else {
          for (int e = 0; e < 4 && cnt < 2; e++) {
            const wreal sgn = (e & 1) ? -WL(1.0) : WL(1.0);
            if (e < 2) {
              for (int k = 0; k < 3; k++) c1[k] = p1[k] + sgn * h1 * a1[k];
              const wreal x2 = seg(p2, a2, h2, c1);
              for (int k = 0; k < 3; k++) c2[k] = p2[k] + x2 * a2[k];
            } else {
              for (int k = 0; k < 3; k++) c2[k] = p2[k] + sgn * h2 * a2[k];
              const wreal x1 = seg(p1, a1, h1, c2);
              for (int k = 0; k < 3; k++) c1[k] = p1[k] + x1 * a1[k];
            }
            spheres(c1, c2);
          }
        }
      }
    }
    if (__ballot(solids_touch) != 0ull && lane == 0) d.counters[2] |= 128;  // (two solids within reach: no narrow phase -- the rollout fails, as the oracle's does)
    int below = 0, total = 0;
    for (int k = 0; k < 2; k++) {
      const unsigned long long b = __ballot(cnt > k);
      below += __popcll(b & ((1ull << lane) - 1ull));
      total += __popcll(b);
    }
    const int base = __builtin_amdgcn_readfirstlane(d.counters[0]);
    WSYNC();
    if (cnt > 0) {
      WaveContact proto;
      proto.g1 = g1; proto.g2 = g2; proto.efc = 0; proto.mu = 0; proto.nrow = 0;
      wf_contact_param(m, g1, g2, proto);
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int at = base + below + k;
        if (k < cnt && at < kWaveMaxCon) {
          WaveContact c = proto;
          c.dist = cd[k];
          for (int e = 0; e < 3; e++) { c.pos[e] = cp[k][e]; c.frame[e] = cn[k][e]; }
          w_make_frame(c.frame);
          d.con[at] = c;
        }
      }
    }
    if (lane == 0) {
      const int n = base + total;
      if (n > kWaveMaxCon) d.counters[2] |= 32;
      d.counters[0] = n > kWaveMaxCon ? kWaveMaxCon : n;
    }
    WSYNC();
  }
}

// ---- o_make_constraint_full: rows in the order friction loss, limits, contacts; then impedance/aref/R per row
template <class MODEL>
__device__ __forceinline__ void wf_make_constraint(const MODEL& m, WaveData& d, int lane) {
  const int nv = m.nv;
  int nefc = 0;
  if (m.disableflags & MJPCX_DSBL_CONSTRAINT) { if (lane == 0) { d.counters[0] = 0; d.counters[1] = 0; } WSYNC(); return; }
  for (int e = lane; e < kWaveMaxEfc * nv; e += 64) d.efc_J[e] = 0;
  if (lane < kWaveMaxEfc) { d.efc_pos[lane] = 0; d.efc_margin[lane] = 0; d.efc_floss[lane] = 0; d.efc_type[lane] = -1; d.efc_id[lane] = 0; }
  WSYNC();
  // friction loss: one lane per dof
  {
    const bool on = !(m.disableflags & MJPCX_DSBL_FRICTIONLOSS) && lane < nv && m.dof_frictionloss[lane] > 0;
    const unsigned long long b = __ballot(on);
    const int r = nefc + __popcll(b & ((1ull << lane) - 1ull));
    if (on && r < kWaveMaxEfc) {
      d.efc_type[r] = kEfcFriction; d.efc_id[r] = lane;
      d.efc_J[r * nv + lane] = 1;
      d.efc_floss[r] = m.dof_frictionloss[lane];
    }
    nefc += __popcll(b);
  }
  // limits: one lane per joint, sides -1 then +1
  {
    bool on0 = false, on1 = false;
    wreal dist0 = 0, dist1 = 0, margin = 0;
    if (!(m.disableflags & MJPCX_DSBL_LIMIT) && lane < m.njnt && m.jnt_limited[lane] &&
        (m.jnt_type[lane] == kJntSlide || m.jnt_type[lane] == kJntHinge)) {
      const wreal value = d.qpos[m.jnt_qposadr[lane]];
      margin = m.jnt_margin[lane];
      dist0 = -(m.jnt_range[2 * lane] - value);      // side -1: lower bound
      dist1 = m.jnt_range[2 * lane + 1] - value;     // side +1: upper bound
      on0 = dist0 < margin; on1 = dist1 < margin;
    }
    const unsigned long long b0 = __ballot(on0), b1 = __ballot(on1);
    const unsigned long long lower = (1ull << lane) - 1ull;
    const int r0 = nefc + __popcll(b0 & lower) + __popcll(b1 & lower);
    const int r1 = r0 + (on0 ? 1 : 0);
    if (on0 && r0 < kWaveMaxEfc) {
      d.efc_type[r0] = kEfcLimit; d.efc_id[r0] = lane; d.efc_J[r0 * nv + m.jnt_dofadr[lane]] = 1;  // -side, side = -1
      d.efc_pos[r0] = dist0; d.efc_margin[r0] = margin;
    }
    if (on1 && r1 < kWaveMaxEfc) {
      d.efc_type[r1] = kEfcLimit; d.efc_id[r1] = lane; d.efc_J[r1 * nv + m.jnt_dofadr[lane]] = -1;
      d.efc_pos[r1] = dist1; d.efc_margin[r1] = margin;
    }
    nefc += __popcll(b0) + __popcll(b1);
  }
  // fixed-tendon limits: one lane per tendon (constant Jacobian = the wrap coefficients)
  {
    bool on0 = false, on1 = false;
    wreal dist0 = 0, dist1 = 0, margin = 0;
    if (!(m.disableflags & MJPCX_DSBL_LIMIT) && lane < m.ntendon && m.tendon_limited[lane]) {
      wreal value = 0;
      for (int w = m.tendon_adr[lane]; w < m.tendon_adr[lane] + m.tendon_num[lane]; w++) value += m.wrap_prm[w] * d.qpos[m.jnt_qposadr[m.wrap_objid[w]]];
      margin = m.tendon_margin[lane];
      dist0 = -(m.tendon_range[2 * lane] - value);
      dist1 = m.tendon_range[2 * lane + 1] - value;
      on0 = dist0 < margin; on1 = dist1 < margin;
    }
    const unsigned long long b0 = __ballot(on0), b1 = __ballot(on1);
    if (b0 | b1) {
      const unsigned long long lower = (1ull << lane) - 1ull;
      const int r0 = nefc + __popcll(b0 & lower) + __popcll(b1 & lower);
      const int r1 = r0 + (on0 ? 1 : 0);
      for (int side = 0; side < 2; side++) {
        const int r = side ? r1 : r0;
        if ((side ? on1 : on0) && r < kWaveMaxEfc) {
          d.efc_type[r] = kEfcTendon; d.efc_id[r] = lane;
          for (int w = m.tendon_adr[lane]; w < m.tendon_adr[lane] + m.tendon_num[lane]; w++)
            d.efc_J[r * nv + m.jnt_dofadr[m.wrap_objid[w]]] = side ? -m.wrap_prm[w] : m.wrap_prm[w];
          d.efc_pos[r] = side ? dist1 : dist0; d.efc_margin[r] = margin;
        }
      }
      nefc += __popcll(b0) + __popcll(b1);
    }
  }
  bool overflow = nefc > kWaveMaxEfc;  // rows beyond the cap are dropped and flagged (oracle new_row: warning 64 -> the rollout fails)
  if (nefc > kWaveMaxEfc) nefc = kWaveMaxEfc;
  WSYNC();
  // contacts: row ranges by a serial prefix over (<= 16) contacts, every lane computes the same numbers
  const int ncon = __builtin_amdgcn_readfirstlane(d.counters[0]);  // wave-uniform, but an LDS load lands in a VGPR: keep loops scalar
  const bool pyramidal = m.cone != 1;
  int my_efc = 0, my_rows = 0;
  {
    int at = nefc;
    for (int ci = 0; ci < ncon; ci++) {
      const int dim = __builtin_amdgcn_readfirstlane(d.con[ci].dim0);
      const int rows = (dim > 1 && pyramidal) ? 2 * (dim - 1) : dim;
      const int fit = at + rows <= kWaveMaxEfc ? rows : (kWaveMaxEfc - at > 0 ? kWaveMaxEfc - at : 0);
      overflow |= fit < rows;
      if (ci == lane) { my_efc = at; my_rows = fit; }
      at += fit;
    }
    nefc = at;
  }
  if (lane < ncon) {
    WaveContact& c = d.con[lane];
    const int dim0 = c.dim0;
    const bool pyr = dim0 > 1 && pyramidal;
    c.efc = my_efc; c.nrow = my_rows;
    c.dim = pyr ? dim0 : my_rows;
    c.dofmask = m.body_dofmask[m.geom_bodyid[c.g1]] ^ m.body_dofmask[m.geom_bodyid[c.g2]];  // common ancestors cancel exactly
    c.mu = pyr ? c.friction[0] : c.friction[0] / sqrt(m.impratio > kMinVal ? m.impratio : WL(1.0));
    for (int row = 0; row < my_rows; row++) {
      const int r = my_efc + row;
      d.efc_type[r] = pyr ? kEfcPyramid : (dim0 == 1 ? kEfcNormal : (row == 0 ? kEfcElliptic : kEfcConeRow));
      d.efc_id[r] = lane;
      if (row == 0 || pyr) { d.efc_pos[r] = c.dist; d.efc_margin[r] = c.includemargin; }
    }
  }
  if (lane == 0) { d.counters[1] = nefc; if (overflow) d.counters[2] |= 64; }
  WSYNC();
  // contact Jacobian rows, J = J(body of geom2) - J(body of geom1) at the contact point: one lane per ROW walks the dofs
  // of its contact's two chains (the rest of the row stays at the zero fill); the contact's data is read once per row
  if (lane < nefc) {
    const int r = lane, t = d.efc_type[r];
    if (t == kEfcNormal || t == kEfcElliptic || t == kEfcConeRow || t == kEfcPyramid) {
      const WaveContact& c = d.con[d.efc_id[r]];
      const int row = r - c.efc;
      const int b1 = m.geom_bodyid[c.g1], b2 = m.geom_bodyid[c.g2];
      const unsigned mask2 = m.body_dofmask[b2];
      const wreal pos[3] = {c.pos[0], c.pos[1], c.pos[2]};
      wreal off1[3], off2[3];
      for (int e = 0; e < 3; e++) {
        off1[e] = pos[e] - d.subtree_com[3 * m.body_rootid[b1] + e];
        off2[e] = pos[e] - d.subtree_com[3 * m.body_rootid[b2] + e];
      }
      // the row is axis ja of the contact frame, plus f times axis jb for a pyramid edge; axes 0..2 act on the translational,
      // 3..5 on the rotational Jacobian of the point
      const bool pyr = t == kEfcPyramid;
      const int ja = pyr ? 0 : row, jb = pyr ? 1 + row / 2 : 0;
      const wreal f = pyr ? ((row & 1) ? -c.friction[jb - 1] : c.friction[jb - 1]) : WL(0.0);
      wreal axa[3], axb[3];
      for (int e = 0; e < 3; e++) { axa[e] = c.frame[3 * (ja < 3 ? ja : ja - 3) + e]; axb[e] = c.frame[3 * (jb < 3 ? jb : jb - 3) + e]; }
      unsigned mask = c.dofmask;
      while (mask) {
        const int k = __ffs((int)mask) - 1;
        mask &= mask - 1;
        const bool second = (mask2 >> k) & 1u;  // the dof is on exactly one of the two chains
        const wreal* cd = d.cdof + 6 * k;
        const wreal cdv[6] = {cd[0], cd[1], cd[2], cd[3], cd[4], cd[5]};
        wreal lin[3];
        cr3(lin, cdv, second ? off2 : off1);
        wreal v = ja < 3 ? axa[0] * (cdv[3] + lin[0]) + axa[1] * (cdv[4] + lin[1]) + axa[2] * (cdv[5] + lin[2])
                         : axa[0] * cdv[0] + axa[1] * cdv[1] + axa[2] * cdv[2];
        if (pyr) {
          const wreal w = jb < 3 ? axb[0] * (cdv[3] + lin[0]) + axb[1] * (cdv[4] + lin[1]) + axb[2] * (cdv[5] + lin[2])
                                 : axb[0] * cdv[0] + axb[1] * cdv[1] + axb[2] * cdv[2];
          v = v + f * w;
        }
        d.efc_J[r * nv + k] = second ? v : -v;
      }
    }
  }
  WSYNC();
  // per-row impedance, reference acceleration, regulariser (cone rows after their normal row)
  wreal kk = 0, bb = 0, vel = 0;
  int type = -1, id = 0;
  if (lane < nefc) {
    const int r = lane;
    type = d.efc_type[r]; id = d.efc_id[r];
    const wreal *solref, *solimp;
    wreal diag;
    if (type == kEfcFriction) { solref = m.dof_solref + 2 * id; solimp = m.dof_solimp + 5 * id; diag = m.dof_invweight0[id]; }
    else if (type == kEfcLimit) { solref = m.jnt_solref + 2 * id; solimp = m.jnt_solimp + 5 * id; diag = m.dof_invweight0[m.jnt_dofadr[id]]; }
    else if (type == kEfcTendon) { solref = m.tendon_solref_lim + 2 * id; solimp = m.tendon_solimp_lim + 5 * id; diag = m.tendon_invweight0[id]; }
    else {
      const WaveContact& c = d.con[id];
      solref = c.solref; solimp = c.solimp;
      const int b1 = m.geom_bodyid[c.g1], b2 = m.geom_bodyid[c.g2];
      diag = m.body_invweight0[2 * b1] + m.body_invweight0[2 * b2];
      if (type == kEfcPyramid) {  // mj_diagApprox: tran + friction^2 * (tran | rot)
        const int j = 1 + (r - c.efc) / 2;
        const wreal f = c.friction[j - 1];
        diag += f * f * (j < 3 ? diag : m.body_invweight0[2 * b1 + 1] + m.body_invweight0[2 * b2 + 1]);
      }
    }
#pragma unroll 6
    for (int k = 0; k < nv; k++) vel += d.efc_J[r * nv + k] * d.qvel[k];
    w_solref_kb(m, solref, solimp, kk, bb);
    if (type != kEfcConeRow) {
      const wreal pos = d.efc_pos[r] - d.efc_margin[r];
      const wreal imp = w_impedance(solimp, pos);
      const wreal R = (1 - imp) / imp * diag;
      d.efc_R[r] = R < kMinVal ? kMinVal : R;
      d.efc_aref[r] = -bb * vel - kk * imp * pos;
      d.efc_D[r] = WL(1.0) / d.efc_R[r];
    }
  }
  WSYNC();
  wreal Rpy = 0;
  if (lane < nefc && type == kEfcConeRow) {
    const int r = lane;
    const WaveContact& c = d.con[id];
    const wreal f = c.friction[r - c.efc - 1];
    d.efc_R[r] = d.efc_R[c.efc] * (c.mu * c.mu) / (f * f);
    d.efc_aref[r] = -bb * vel;
    d.efc_D[r] = WL(1.0) / d.efc_R[r];
  } else if (lane < nefc && type == kEfcPyramid) {  // every edge: Rpy = 2 mu^2 R of the first edge (read, sync, then write)
    const WaveContact& c = d.con[id];
    Rpy = 2 * c.mu * c.mu * d.efc_R[c.efc];
    if (Rpy < kMinVal) Rpy = kMinVal;
  }
  WSYNC();
  if (lane < nefc && type == kEfcPyramid) { d.efc_R[lane] = Rpy; d.efc_D[lane] = WL(1.0) / Rpy; }
  WSYNC();
}

// ---- penalties: one lane per row; a cone is evaluated by the lane of its first row
struct ConeEval { wreal cost, g, h; };
// value/force/zone at x = jar (+ alpha jv when jv != nullptr); derivative terms along jv when requested
// LDS-typed views (address space 3): the out-of-line row evaluator would otherwise see generic pointers and go through
// FLAT loads (aperture check, vmcnt + lgkmcnt) instead of ds_read
typedef __attribute__((address_space(3))) wreal wlds_f64;
typedef __attribute__((address_space(3))) int wlds_i32;
typedef __attribute__((address_space(3))) WaveContact wlds_con;
struct RowView {  // by value into the out-of-line row evaluator
  const wlds_i32 *efc_type, *efc_id;
  wlds_i32* efc_zone;
  const wlds_f64 *efc_D, *efc_R, *efc_floss;
  wlds_f64* efc_force;
  const wlds_con* con;
};
struct RowResult { wreal cost, g1, h2; };
__device__ __noinline__ RowResult wf_row_eval_impl(const RowView d, int r, const wlds_f64* jar, const wlds_f64* jv, bool have_jv,
                                                   wreal alpha, bool write_force) {
  wreal cost = 0, g1 = 0, h2 = 0;
  const int type = d.efc_type[r];
  const wreal D = d.efc_D[r];
  const wreal v = have_jv ? jv[r] : WL(0.0);
  const wreal x = jar[r] + alpha * v;
  if (type == kEfcFriction) {
    const wreal f = d.efc_floss[r], R = d.efc_R[r];
    if (x <= -R * f) { cost = -WL(0.5) * R * f * f - f * x; g1 = -f * v; if (write_force) { d.efc_force[r] = f; d.efc_zone[r] = kZoneTop; } }
    else if (x >= R * f) { cost = -WL(0.5) * R * f * f + f * x; g1 = f * v; if (write_force) { d.efc_force[r] = -f; d.efc_zone[r] = kZoneTop; } }
    else { cost = WL(0.5) * D * x * x; g1 = D * x * v; h2 = D * v * v; if (write_force) { d.efc_force[r] = -D * x; d.efc_zone[r] = kZoneBottom; } }
  } else if (type == kEfcLimit || type == kEfcNormal || type == kEfcTendon || type == kEfcPyramid) {
    if (x < 0) { cost = WL(0.5) * D * x * x; g1 = D * x * v; h2 = D * v * v; if (write_force) { d.efc_force[r] = -D * x; d.efc_zone[r] = kZoneBottom; } }
    else if (write_force) { d.efc_force[r] = 0; d.efc_zone[r] = kZoneTop; }
  } else if (type == kEfcElliptic) {
    // loops are unrolled to the maximum cone dimension with guards: static indices keep U/V/X in registers (run-time
    // trip counts would put them in scratch, and this runs for every line-search trial)
    const wlds_con& c = d.con[d.efc_id[r]];
    const int dim = c.dim;
    const wreal mu = c.mu;
    wreal U[6], V[6], X[6], Dj[6], T = 0;
    X[0] = x; U[0] = x * mu; V[0] = v * mu; Dj[0] = D;
#pragma unroll
    for (int j = 1; j < 6; j++) {
      if (j < dim) {
        const wreal vj = have_jv ? jv[r + j] : WL(0.0), fj = c.friction[j - 1];
        X[j] = jar[r + j] + alpha * vj;
        U[j] = X[j] * fj;
        V[j] = vj * fj;
        Dj[j] = d.efc_D[r + j];
        T += U[j] * U[j];
      } else { X[j] = U[j] = V[j] = Dj[j] = 0; }
    }
    T = sqrt(T);
    const wreal N = U[0];
    if (N >= mu * T || (T <= 0 && N >= 0)) {
      if (write_force) {
#pragma unroll
        for (int j = 0; j < 6; j++) if (j < dim) d.efc_force[r + j] = 0;
        d.efc_zone[r] = kZoneTop;
      }
    } else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
      // bottom zone: plain quadratic in every row (vj recovered from jv to keep the oracle's arithmetic)
#pragma unroll
      for (int j = 0; j < 6; j++) {
        if (j < dim) {
          const wreal vj = have_jv ? jv[r + j] : WL(0.0);
          cost += WL(0.5) * Dj[j] * X[j] * X[j]; g1 += Dj[j] * X[j] * vj; h2 += Dj[j] * vj * vj;
          if (write_force) d.efc_force[r + j] = -Dj[j] * X[j];
        }
      }
      if (write_force) d.efc_zone[r] = kZoneBottom;
    } else {
      const wreal Dm = D / (mu * mu * (1 + mu * mu)), NT = N - mu * T;
      cost = WL(0.5) * Dm * NT * NT;
      wreal UV = 0, VV = 0;
#pragma unroll
      for (int j = 1; j < 6; j++) if (j < dim) { UV += U[j] * V[j]; VV += V[j] * V[j]; }
      const wreal dNT = V[0] - mu * UV / T;
      const wreal d2NT = -mu * (VV / T - UV * UV / (T * T * T));
      g1 = Dm * NT * dNT;
      h2 = Dm * (dNT * dNT + NT * d2NT);
      if (write_force) {
        d.efc_force[r] = -Dm * NT * mu;
#pragma unroll
        for (int j = 1; j < 6; j++) if (j < dim) d.efc_force[r + j] = Dm * NT * mu * U[j] * c.friction[j - 1] / T;
        d.efc_zone[r] = kZoneMiddle;
      }
    }
  }
  return RowResult{cost, g1, h2};
}
__device__ __forceinline__ void wf_row_eval(const WaveData& d, int r, const wreal* jar, const wreal* jv, wreal alpha,
                                            bool write_force, wreal& cost, wreal& g1, wreal& h2) {
  const RowView v{(const wlds_i32*)d.efc_type, (const wlds_i32*)d.efc_id, (wlds_i32*)d.efc_zone, (const wlds_f64*)d.efc_D,
                  (const wlds_f64*)d.efc_R, (const wlds_f64*)d.efc_floss, (wlds_f64*)d.efc_force, (const wlds_con*)d.con};
  const RowResult res = wf_row_eval_impl(v, r, (const wlds_f64*)jar, (const wlds_f64*)(jv ? jv : jar), jv != nullptr, alpha, write_force);
  cost = res.cost; g1 = res.g1; h2 = res.h2;
}

// ---- line search: the row's data is loaded ONCE per Newton iteration into registers; every trial alpha is then pure
// register arithmetic (same operations, in the same order, as wf_row_eval_impl with write_force = false)
struct LsRow {
  int type, dim;          // type < 0: nothing to evaluate (inactive lane, member row of a cone)
  wreal D, R, fl, mu;     // row 0 of the cone / the row itself
  wreal x0[6], v[6], f[6], Dj[6];
};
__device__ __forceinline__ LsRow ls_load(const WaveData& d, int r, int ne) {
  LsRow q;
  q.type = -1; q.dim = 1; q.D = q.R = q.fl = q.mu = 0;
#pragma unroll
  for (int j = 0; j < 6; j++) { q.x0[j] = q.v[j] = q.f[j] = q.Dj[j] = 0; }
  if (r >= ne) return q;
  const int t = d.efc_type[r];
  if (t == kEfcConeRow) return q;
  q.type = t;
  q.D = d.efc_D[r]; q.R = d.efc_R[r]; q.fl = d.efc_floss[r];
  q.x0[0] = d.jar[r]; q.v[0] = d.jv[r]; q.Dj[0] = q.D;
  if (t == kEfcElliptic) {
    const WaveContact& c = d.con[d.efc_id[r]];
    q.dim = c.dim; q.mu = c.mu;
#pragma unroll
    for (int j = 1; j < 6; j++)
      if (j < q.dim) { q.x0[j] = d.jar[r + j]; q.v[j] = d.jv[r + j]; q.f[j] = c.friction[j - 1]; q.Dj[j] = d.efc_D[r + j]; }
  }
  return q;
}
// first and second derivative along the search direction of this row's penalty at jar + alpha jv
__device__ __forceinline__ void ls_eval(const LsRow& q, wreal alpha, wreal& g1, wreal& h2) {
  g1 = 0; h2 = 0;
  if (q.type < 0) return;
  const wreal v = q.v[0], x = q.x0[0] + alpha * v, D = q.D;
  if (q.type == kEfcFriction) {
    const wreal f = q.fl, R = q.R;
    if (x <= -R * f) g1 = -f * v;
    else if (x >= R * f) g1 = f * v;
    else { g1 = D * x * v; h2 = D * v * v; }
  } else if (q.type != kEfcElliptic) {  // limit, tendon limit, frictionless contact, pyramid edge
    if (x < 0) { g1 = D * x * v; h2 = D * v * v; }
  } else {
    const wreal mu = q.mu;
    wreal U[6], V[6], X[6], T = 0;
    X[0] = x; U[0] = x * mu; V[0] = v * mu;
#pragma unroll
    for (int j = 1; j < 6; j++) {
      if (j < q.dim) { X[j] = q.x0[j] + alpha * q.v[j]; U[j] = X[j] * q.f[j]; V[j] = q.v[j] * q.f[j]; T += U[j] * U[j]; }
      else { X[j] = U[j] = V[j] = 0; }
    }
    T = sqrt(T);
    const wreal N = U[0];
    if (N >= mu * T || (T <= 0 && N >= 0)) {
    } else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
#pragma unroll
      for (int j = 0; j < 6; j++)
        if (j < q.dim) { g1 += q.Dj[j] * X[j] * q.v[j]; h2 += q.Dj[j] * q.v[j] * q.v[j]; }
    } else {
      const wreal Dm = D / (mu * mu * (1 + mu * mu)), NT = N - mu * T;
      wreal UV = 0, VV = 0;
#pragma unroll
      for (int j = 1; j < 6; j++) if (j < q.dim) { UV += U[j] * V[j]; VV += V[j] * V[j]; }
      const wreal dNT = V[0] - mu * UV / T;
      const wreal d2NT = -mu * (VV / T - UV * UV / (T * T * T));
      g1 = Dm * NT * dNT;
      h2 = Dm * (dNT * dNT + NT * d2NT);
    }
  }
}

// cost of all rows at jar; writes force and zone. Returns the wave-uniform sum.
__device__ __forceinline__ wreal wf_constraint_cost(WaveData& d, int nefc, int lane) {
  wreal c = 0, g, h;
  if (lane < nefc) wf_row_eval(d, lane, d.jar, nullptr, WL(0.0), true, c, g, h);
  c = wave_sum(c);
  WSYNC();
  return c;
}

// ---- o_constraint_newton
template <int NMAX, class MODEL>
__device__ __forceinline__ void wf_constraint_newton(const MODEL& m, WaveData& d, int lane, long long* stamp = nullptr, bool have_warm = false) {
  const int nv = m.nv, ne = __builtin_amdgcn_readfirstlane(d.counters[1]);
  if (lane < nv) { d.qfrc_constraint[lane] = 0; d.qacc[lane] = d.qacc_smooth[lane]; }
  WSYNC();
  if (ne == 0) return;
  if (lane < ne) {
    wreal s = -d.efc_aref[lane];
#pragma unroll 6
    for (int k = 0; k < nv; k++) s += d.efc_J[lane * nv + k] * d.qacc[k];
    d.jar[lane] = s;
  }
  WSYNC();
  wreal cost = wf_constraint_cost(d, ne, lane);
  if (have_warm) {  // warm start (mj_fwdConstraint): begin at the previous step's qacc if its cost is lower
    wreal jsave = 0, jw = 0, gauss = 0;
    if (lane < ne) {
      jsave = d.jar[lane];
      wreal s = -d.efc_aref[lane];
#pragma unroll 6
      for (int k = 0; k < nv; k++) s += d.efc_J[lane * nv + k] * d.qacc_warm[k];
      jw = s;
    }
    if (lane < nv) {
      wreal s = 0;
#pragma unroll 6
      for (int b = 0; b < nv; b++) s += d.M[lane * nv + b] * (d.qacc_warm[b] - d.qacc_smooth[b]);
      gauss = WL(0.5) * s * (d.qacc_warm[lane] - d.qacc_smooth[lane]);
    }
    gauss = wave_sum(gauss);
    WSYNC();
    if (lane < ne) d.jar[lane] = jw;
    WSYNC();
    const wreal cw = gauss + wf_constraint_cost(d, ne, lane);
    if (cw < cost) {
      cost = cw;
      if (lane < nv) d.qacc[lane] = d.qacc_warm[lane];
      WSYNC();
    } else {
      if (lane < ne) d.jar[lane] = jsave;
      WSYNC();
      wf_constraint_cost(d, ne, lane);  // restore force / zone of the smooth start
    }
  }
  const wreal scale = WL(1.0) / (m.meaninertia * (nv > 1 ? nv : 1));
  bool factor_valid = false, refresh = true;
  wreal improvement = 0;
  const int my_type = lane < ne ? d.efc_type[lane] : -1;
  int my_zone_prev = -2;
  long long tacc = 0;
#define WACC(k) do { if (stamp && lane == 0) { const long long now_ = (long long)__builtin_readcyclecounter(); stamp[k] += now_ - tacc; tacc = now_; } } while (0)
  for (int iter = 0; iter < m.solver_iterations; iter++) {
    if (stamp && lane == 0) tacc = (long long)__builtin_readcyclecounter();
    // gradient = M (qacc - qacc_smooth) - J' force
    wreal g = 0;
    if (lane < nv) {
      wreal s = 0;
#pragma unroll 6
      for (int b = 0; b < nv; b++) s += d.M[lane * nv + b] * (d.qacc[b] - d.qacc_smooth[b]);
      d.Ma[lane] = s;
      g = s;
#pragma unroll 8
      for (int r = 0; r < ne; r++) g -= d.efc_J[r * nv + lane] * d.efc_force[r];
      d.grad[lane] = g;
      d.search[lane] = -g;
    }
    const wreal gnorm = sqrt(wave_sum(lane < nv ? g * g : WL(0.0)));
    if (gnorm == 0) break;
    // termination as in MuJoCo's primal solvers (engine_solver.c): the test uses the gradient AFTER the update, so it sits
    // between the gradient and the Hessian work of the next pass. float: tolerance floored at what float resolves of a cost
    // of this size
    {
      const wreal tol = sizeof(wreal) == 4 ? fmax((wreal)m.solver_tolerance, WL(1e-7)) : (wreal)m.solver_tolerance;
      if (iter > 0 && (scale * improvement < tol || scale * gnorm < tol)) break;
    }
    if (stamp && lane == 0 && iter == 0) stamp[21] = (long long)__builtin_readcyclecounter();
    WACC(32);
    // H = M + J' (d2s) J depends on the rows' zones only -- and on jar for a cone in its middle (sliding) zone. When no row
    // changed zone since the last factorisation and no cone slides, the factor still in d.H is bit for bit what a
    // recomputation would give (the oracle recomputes): reuse it. Typical for the last iterations of a solve.
    {
      const int zn = (lane < ne && my_type != kEfcConeRow) ? d.efc_zone[lane] : -1;
      const bool moved = zn != my_zone_prev || (my_type == kEfcElliptic && zn == kZoneMiddle);
      my_zone_prev = zn;
      refresh = !factor_valid || __any(moved);
    }
    if (refresh) {
    // cone Hessian blocks (one lane per contact), then H = M + J' (d2s) J over the lower triangle
    if (lane < d.counters[0]) {
      const WaveContact& c = d.con[lane];
      const int r = c.efc, dim = c.dim;
      wreal* Hs = d.coneH + 21 * lane;
      if (c.nrow > 0 && dim > 0 && d.efc_type[r] == kEfcElliptic) {
        // lower triangle only, unrolled to dimension 6 with guards (register-resident)
        const int zone = d.efc_zone[r];
        const wreal mu = c.mu;
        wreal U[6], sc[6], T = 0;
        sc[0] = mu; U[0] = d.jar[r] * mu;
#pragma unroll
        for (int j = 1; j < 6; j++) {
          if (j < dim) { sc[j] = c.friction[j - 1]; U[j] = d.jar[r + j] * sc[j]; T += U[j] * U[j]; } else { sc[j] = 0; U[j] = 0; }
        }
        T = sqrt(T);
        const wreal Dm = d.efc_D[r] / (mu * mu * (1 + mu * mu)), NT = U[0] - mu * T;
        // one reciprocal instead of ~40 divisions per block (differs from the oracle's divisions by a few ulp)
        const wreal iT = zone == kZoneMiddle ? WL(1.0) / T : WL(0.0), iT2 = iT * iT, iT3 = iT2 * iT;
#pragma unroll
        for (int j = 0; j < 6; j++) {
#pragma unroll
          for (int k = 0; k <= j; k++) {
            if (j < dim) {
              wreal hjk = 0;
              if (zone == kZoneBottom) hjk = j == k ? d.efc_D[r + j] : WL(0.0);
              else if (zone == kZoneMiddle) {
                if (j == 0) hjk = Dm;                                   // (0, 0)
                else if (k == 0) hjk = -Dm * mu * U[j] * iT;            // (j, 0)
                else hjk = Dm * mu * mu * U[j] * U[k] * iT2 - Dm * NT * mu * ((j == k ? iT : WL(0.0)) - U[j] * U[k] * iT3);
                hjk *= sc[j] * sc[k];
              }
              Hs[j * (j + 1) / 2 + k] = hjk;
            }
          }
        }
      }
    }
    WSYNC();
    WACC(33);
    // H = M + J' (d2s) J. Friction-loss and limit rows have a single non-zero Jacobian entry: diagonal updates, one lane
    // per row (friction rows have distinct dofs; a limit row may share its dof with a friction row -> two passes).
    // Contact rows: the lower-triangle entries are dealt over the lanes; a contact only touches the dofs on the chain
    // of its body (dofmask), which skips ~3/4 of the (entry, contact) pairs on a legged robot.
    for (int e = lane; e < nv * nv; e += 64) d.H[e] = d.M[e];
    WSYNC();
    WACC(40);
    for (int pass = 0; pass < 2; pass++) {
      if (lane < ne) {
        const int t = d.efc_type[lane];
        if (t == (pass == 0 ? kEfcFriction : kEfcLimit) && d.efc_zone[lane] == kZoneBottom) {
          const int dof = t == kEfcFriction ? d.efc_id[lane] : m.jnt_dofadr[d.efc_id[lane]];
          d.H[dof * nv + dof] += d.efc_D[lane];
        }
      }
      WSYNC();
    }
    // Per-row facts in registers, lane = row, fetched in the entry loop by v_readlane (no dependent LDS chain per row
    // visit): kind 0 = nothing to add (inactive, cone member row, diagonal-only row), 1 = active simple inequality row
    // (frictionless contact, pyramid edge, tendon limit), 2 = head of an elliptic cone outside the top zone.
    WACC(41);
    int my_kind = 0, my_id = 0, my_dim = 1;
    unsigned my_mask = 0;
    wreal my_D = 0;
    {
      const int t = lane < ne ? d.efc_type[lane] : -1;
      const bool is_contact = t >= kEfcNormal;
      if (is_contact) {
        const int id = d.efc_id[lane], zone = d.efc_zone[lane];
        if (t == kEfcElliptic) { my_kind = zone != kZoneTop ? 2 : 0; my_mask = d.con[id].dofmask; my_id = id; my_dim = d.con[id].dim; }
        else if (t != kEfcConeRow && zone == kZoneBottom) { my_kind = 1; my_mask = t == kEfcTendon ? m.tendon_dofmask[id] : d.con[id].dofmask; my_D = d.efc_D[lane]; }
      }
    }
    const unsigned long long simple_rows = __ballot(my_kind == 1), cone_rows = __ballot(my_kind == 2);
    WACC(42);
    if (simple_rows) {  // (nothing to add on a step whose only active rows are cones and diagonal rows)
      // Simple inequality rows (frictionless contacts, pyramid edges, tendon limits): H += J' diag(s) J with s_r = D_r on the
      // active rows and 0 elsewhere -- a (nv x ne)(ne x nv) product on the matrix cores, 16 x 16 output tiles, K = 4 rows per
      // MFMA (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32). A fragment: lane l holds J'[a = tile_i + (l & 15)][r = 4u + (l >> 4)],
      // B fragment: s_r J[r][b = tile_j + (l & 15)]. The row scales go through LDS (jv is dead until the line search).
      if (lane < kWaveMaxEfc) d.jv[lane] = my_kind == 1 ? my_D : WL(0.0);
      WSYNC();
      const int li = lane & 15, lk = lane >> 4;
      const int ksteps = (ne + 3) >> 2;
      for (int ti = 0; ti < nv; ti += 16)
        for (int tj = 0; tj < nv; tj += 16) {
          const int ia = ti + li, jb = tj + li;
          const bool av = ia < nv, bv = jb < nv;
          w_acc4 acc = {0, 0, 0, 0};
          for (int u = 0; u < ksteps; u++) {
            const int r = 4 * u + lk;
            const bool rv = r < ne;
            const wreal sr = rv ? d.jv[r] : WL(0.0);
            const wreal fa = (rv && av) ? d.efc_J[r * nv + ia] : WL(0.0);
            const wreal fb = (rv && bv) ? sr * d.efc_J[r * nv + jb] : WL(0.0);
            acc = w_mfma_16x16x4(fa, fb, acc);
          }
#pragma unroll
          for (int rg = 0; rg < 4; rg++) {
            const int ci = ti + w_mfma_row(lk, rg);
            if (ci < nv && bv) d.H[ci * nv + jb] += acc[rg];
          }
        }
    }
    WSYNC();
    WACC(43);
    // elliptic cones, one at a time (in row order): the cone's block J_c' Hc J_c only touches the nd <= ~15 dofs of its chains,
    // so one lane per entry of that nd x nd triangle adds its contribution into H. (The entry-major loop above left lanes
    // holding trunk entries with one update per cone while most other lanes idled.)
    for (unsigned long long todo = cone_rows; todo; todo &= todo - 1) {
      const int r = __ffsll((long long)todo) - 1;
      const unsigned mask = (unsigned)__builtin_amdgcn_readlane((int)my_mask, r);
      const int ci = __builtin_amdgcn_readlane(my_id, r);
      const int dim = __builtin_amdgcn_readlane(my_dim, r);
      const int nd = __popc(mask);
      for (int e = lane; e < nd * (nd + 1) / 2; e += 64) {
        int ia = (int)((sqrtf(8.0f * e + 1.0f) - 1.0f) * 0.5f);
        while ((ia + 1) * (ia + 2) / 2 <= e) ia++;
        while (ia * (ia + 1) / 2 > e) ia--;
        const int ib = e - ia * (ia + 1) / 2;
        const int a = w_nth_bit(mask, ia), b = w_nth_bit(mask, ib);  // a >= b
        const wreal* Hs = d.coneH + 21 * ci;
        wreal Ja[6], Jb[6], Hl[21];  // fully unrolled with guards: static indices keep these in registers
#pragma unroll
        for (int j = 0; j < 6; j++) { Ja[j] = j < dim ? d.efc_J[(r + j) * nv + a] : WL(0.0); Jb[j] = j < dim ? d.efc_J[(r + j) * nv + b] : WL(0.0); }
#pragma unroll
        for (int q = 0; q < 21; q++) Hl[q] = q < dim * (dim + 1) / 2 ? Hs[q] : WL(0.0);
        wreal h = 0;
#pragma unroll
        for (int j = 0; j < 6; j++) {  // row j of the symmetric block from its packed lower triangle
          wreal sj = 0;
#pragma unroll
          for (int k = 0; k < 6; k++) sj += Hl[j >= k ? j * (j + 1) / 2 + k : k * (k + 1) / 2 + j] * Jb[k];
          h += Ja[j] * sj;
        }
        const wreal hn = d.H[a * nv + b] + h;
        d.H[a * nv + b] = hn;
        d.H[b * nv + a] = hn;
      }
      WSYNC();
    }
    if (stamp && lane == 0 && iter == 0) stamp[22] = (long long)__builtin_readcyclecounter();
    WACC(34);
    if (!wave_chol<NMAX>(d.H, d.dinv, nv, lane)) { if (lane == 0) d.counters[2] |= 16; WSYNC(); break; }
    factor_valid = true;
    }  // refresh
    wave_chol_solve<NMAX>(d.search, d.H, d.dinv, nv, lane);
    if (stamp && lane == 0 && iter == 0) stamp[23] = (long long)__builtin_readcyclecounter();
    WACC(35);
    // jv = J search; Gauss part along the ray
    if (lane < ne) {
      wreal s = 0;
#pragma unroll 6
      for (int k = 0; k < nv; k++) s += d.efc_J[lane * nv + k] * d.search[k];
      d.jv[lane] = s;
    }
    wreal q1 = 0, q2 = 0;
    if (lane < nv) {
      wreal s = 0;
#pragma unroll 6
      for (int b = 0; b < nv; b++) s += d.M[lane * nv + b] * d.search[b];
      q1 = d.search[lane] * d.Ma[lane];
      q2 = d.search[lane] * s;
    }
    q1 = wave_sum(q1); q2 = wave_sum(q2);
    WSYNC();
    WACC(36);
    // exact line search: safeguarded 1-D Newton on the (convex, piecewise quadratic) restriction
    wreal lo = 0, hi = -1, alpha = 0, d1, d2, g0, h0;
    const LsRow lsrow = ls_load(d, lane, ne);
    ls_eval(lsrow, WL(0.0), g0, h0);
    d1 = wave_sum(g0) + q1; d2 = wave_sum(h0) + q2;
    const wreal d10 = fabs(d1);
    // termination as in MuJoCo's PrimalSearch: |derivative| < tolerance * ls_tolerance * |search| / scale
    wreal gtol = m.solver_tolerance * kLsTolerance * sqrt(wave_sum(lane < nv ? d.search[lane] * d.search[lane] : WL(0.0))) / scale;
    if (sizeof(wreal) == 4) gtol = fmax(gtol, WL(1e-4) * d10);  // float: the slope's rounding floor is far above MuJoCo's tolerance
    wreal step1 = WL(1e30), step2 = WL(1e30);  // the last step and the one before (rtsafe safeguard, oracle/contact.inc)
    for (int ls = 0; ls < 50 && d10 >= gtol; ls++) {
      wreal an = alpha - d1 / d2;
      if (!(an > lo) || (hi >= 0 && !(an < hi))) an = hi >= 0 ? WL(0.5) * (lo + hi) : 2 * alpha + 1;
      else if (hi >= 0 && fabs(an - alpha) > WL(0.5) * step2) an = WL(0.5) * (lo + hi);
      if (an == alpha) break;
      step2 = step1; step1 = fabs(an - alpha);
      alpha = an;
      ls_eval(lsrow, alpha, g0, h0);
      d1 = wave_sum(g0) + q1 + alpha * q2; d2 = wave_sum(h0) + q2;
      if (fabs(d1) < gtol) break;
      if (d1 < 0) lo = alpha; else hi = alpha;
      if (stamp && lane == 0) stamp[39]++;
    }
    WACC(37);
    if (stamp && lane == 0 && iter == 0) stamp[24] = (long long)__builtin_readcyclecounter();
    if (lane < nv) d.qacc[lane] += alpha * d.search[lane];
    if (lane < ne) d.jar[lane] += alpha * d.jv[lane];
    WSYNC();
    wreal gauss = 0;
    if (lane < nv) {
      wreal s = 0;
#pragma unroll 6
      for (int b = 0; b < nv; b++) s += d.M[lane * nv + b] * (d.qacc[b] - d.qacc_smooth[b]);
      gauss = WL(0.5) * s * (d.qacc[lane] - d.qacc_smooth[lane]);
    }
    gauss = wave_sum(gauss);
    const wreal newcost = gauss + wf_constraint_cost(d, ne, lane);
    improvement = cost - newcost;
    cost = newcost;
    WACC(38);
    if (stamp && lane == 0) stamp[20] = iter + 1;
  }
  if (lane < nv) {
    wreal s = 0;
#pragma unroll 8
    for (int r = 0; r < ne; r++) s += d.efc_J[r * nv + lane] * d.efc_force[r];
    d.qfrc_constraint[lane] = s;
  }
  WSYNC();
}

// ---- mj_forward up to the constraint solve
#define WSTAMP(k) do { if (stamp && lane == 0) stamp[k] = (long long)__builtin_readcyclecounter(); } while (0)
template <int NMAX, class MODEL, class TASK>
__device__ __forceinline__ void wf_forward(const MODEL& m, const TASK& tk, WaveData& d, int lane, bool& bad_ctrl,
                                           long long* stamp, bool have_warm) {
  const int nv = m.nv;
  WSTAMP(1);
  wf_kinematics(m, tk, d, lane);
  WSYNC();
  WSTAMP(2);
  wf_compos(m, d, lane);
  WSTAMP(3);
  wf_crb(m, d, lane);
  WSTAMP(4);
  for (int e = lane; e < nv * nv; e += 64) d.L[e] = d.M[e];
  WSYNC();
  if (!wave_chol<NMAX>(d.L, d.Ldinv, nv, lane)) { if (lane == 0) d.counters[2] |= 16; }
  WSTAMP(5);
  wf_collision(m, d, lane);
  WSTAMP(6);
  wf_comvel(m, d, lane);
  WSTAMP(7);
  wf_make_constraint(m, d, lane);
  WSTAMP(8);
  wf_smooth_forces(m, d, lane, bad_ctrl);
  WSTAMP(9);
  wave_chol_solve<NMAX>(d.qacc_smooth, d.L, d.Ldinv, nv, lane);
  WSTAMP(10);
  wf_constraint_newton<NMAX>(m, d, lane, stamp, have_warm);
  WSTAMP(11);
}

// ---- mj_integratePos: qpos advanced by h along the generalized velocity `vel` (one lane per joint; quaternions by the exponential map)
template <class MODEL>
__device__ __forceinline__ void w_integrate_pos(const MODEL& m, WaveData& d, const wreal* vel, wreal h, int lane) {
  if (lane < m.njnt) {
    const int j = lane;
    int qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j];
    const int jt = m.jnt_type[j];
    if (jt == kJntFree) {
      for (int k = 0; k < 3; k++) d.qpos[qa + k] += h * vel[da + k];
      qa += 3; da += 3;
    }
    if (jt == kJntFree || jt == kJntBall) {
      wreal ax[3] = {vel[da], vel[da + 1], vel[da + 2]};
      const wreal n = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
      if (n < kMinVal) { ax[0] = 1; ax[1] = ax[2] = 0; }
      else { ax[0] /= n; ax[1] /= n; ax[2] /= n; }
      wreal qrot[4], q[4];
      aa2quat(qrot, ax, h * n);
      for (int k = 0; k < 4; k++) q[k] = d.qpos[qa + k];
      q_norm(q);
      q_mul(q, q, qrot);
      for (int k = 0; k < 4; k++) d.qpos[qa + k] = q[k];
    } else {
      d.qpos[qa] += h * vel[da];
    }
  }
}

// ---- o_euler: implicit joint damping, then integrate positions
template <int NMAX, class MODEL>
__device__ __forceinline__ void wf_euler(const MODEL& m, WaveData& d, int lane, wreal& time) {
  const int nv = m.nv;
  const wreal h = m.timestep;
  if (m.any_damping && !(m.disableflags & MJPCX_DSBL_EULERDAMP)) {
    for (int e = lane; e < nv * nv; e += 64) d.H[e] = d.M[e];
    WSYNC();
    if (lane < nv) { d.H[lane * nv + lane] += h * m.dof_damping[lane]; d.tmpv[lane] = d.qfrc_smooth[lane] + d.qfrc_constraint[lane]; }
    WSYNC();
    if (wave_chol<NMAX>(d.H, d.dinv, nv, lane)) wave_chol_solve<NMAX>(d.tmpv, d.H, d.dinv, nv, lane);
    else { if (lane < nv) d.tmpv[lane] = d.qacc[lane]; WSYNC(); }
  } else {
    if (lane < nv) d.tmpv[lane] = d.qacc[lane];
    WSYNC();
  }
  if (lane < nv) d.qvel[lane] += h * d.tmpv[lane];
  WSYNC();
  w_integrate_pos(m, d, d.qvel, h, lane);
  time += h;
  WSYNC();
}

} }  // namespace mjpcx::WAVE_NS
