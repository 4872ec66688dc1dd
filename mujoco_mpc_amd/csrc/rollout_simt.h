// rollout_simt.h -- lane-per-candidate rollout kernel for the contact models (the throughput path of the Quadruped
// class; rollout_wave.h is the low-latency path).
//
// Why: one step of the A1 is ~40 k dependent fp64 instructions. A wavefront that spreads ONE candidate over its
// lanes (rollout_wave_kernel) keeps a handful of lanes busy per instruction and is issue-bound at ~0.3-0.8 ms per
// step; here every lane integrates its OWN candidate with the oracle's scalar algorithm, so each instruction does
// useful work for 64 candidates. Per-lane state (~44 KB: mjData arrays, constraint rows, the Newton system) lives
// in a global scratch slab laid out [array element][lane]: the 64 lanes of a wavefront touch 64 consecutive
// doubles, i.e. one fully coalesced 512-byte access per instruction. Model constants are wave-uniform loads
// (scalar cache). Control flow follows the data per lane (contact counts differ): SIMT divergence, not branches
// around cooperative phases.
//
// The arithmetic is oracle/physics.c + oracle/contact.inc statement by statement (same operation order per
// candidate), so agreement with the oracle is limited only by FMA contraction and libm.
#pragma once
#include "device_common.h"
#include "rollout_lane.h"
#include "rollout_wave.h"  // WaveModel / WaveTask / small math helpers / residual constants

namespace mjpcx {
using namespace w64;  // the fp64 instantiation of the wave helpers (rollout_wave.h)

struct SimtLayout {  // offsets (in elements) of the per-lane arrays inside a wavefront's slab
  int qpos, qvel, ctrl, xpos, xquat, xmat, xipos, ximat, xanchor, xaxis, site_xpos, subtree_com, cinert, crb, cdof, cdof_dot,
      cvel, cacc, cfrc, M, L, H, qfrc_passive, qfrc_bias, qfrc_actuator, qfrc_smooth, qacc_smooth, qacc, qfrc_constraint,
      actuator_force, grad, search, Ma, tmpv, qacc_warm, efc_J, efc_pos, efc_margin, efc_D, efc_R, efc_aref, efc_floss, efc_force, jar, jv,
      efc_type, efc_id, efc_zone, con, residual, nodes, total;
};
constexpr int kSimtConDoubles = 32;  // per contact: dist margin includemargin mu pos[3] frame[9] friction[5] solref[2] solimp[5] g1 g2 dim efc

inline SimtLayout simt_layout(int nq, int nv, int nu, int nb, int nj, int ns, int nr, int P) {
  SimtLayout o{};
  int at = 0;
  auto seg = [&](int n) { int r = at; at += n; return r; };
  o.qpos = seg(nq); o.qvel = seg(nv); o.ctrl = seg(nu);
  o.xpos = seg(3 * nb); o.xquat = seg(4 * nb); o.xmat = seg(9 * nb); o.xipos = seg(3 * nb); o.ximat = seg(9 * nb);
  o.xanchor = seg(3 * nj); o.xaxis = seg(3 * nj); o.site_xpos = seg(3 * ns); o.subtree_com = seg(3 * nb);
  o.cinert = seg(10 * nb); o.crb = seg(10 * nb); o.cdof = seg(6 * nv); o.cdof_dot = seg(6 * nv);
  o.cvel = seg(6 * nb); o.cacc = seg(6 * nb); o.cfrc = seg(6 * nb);
  o.M = seg(nv * nv); o.L = seg(nv * nv); o.H = seg(nv * nv);
  o.qfrc_passive = seg(nv); o.qfrc_bias = seg(nv); o.qfrc_actuator = seg(nv); o.qfrc_smooth = seg(nv); o.qacc_smooth = seg(nv);
  o.qacc = seg(nv); o.qfrc_constraint = seg(nv); o.actuator_force = seg(nu); o.grad = seg(nv); o.search = seg(nv); o.Ma = seg(nv);
  o.tmpv = seg(nv); o.qacc_warm = seg(nv);
  o.efc_J = seg(kWaveMaxEfc * nv);
  o.efc_pos = seg(kWaveMaxEfc); o.efc_margin = seg(kWaveMaxEfc); o.efc_D = seg(kWaveMaxEfc); o.efc_R = seg(kWaveMaxEfc);
  o.efc_aref = seg(kWaveMaxEfc); o.efc_floss = seg(kWaveMaxEfc); o.efc_force = seg(kWaveMaxEfc); o.jar = seg(kWaveMaxEfc);
  o.jv = seg(kWaveMaxEfc); o.efc_type = seg(kWaveMaxEfc); o.efc_id = seg(kWaveMaxEfc); o.efc_zone = seg(kWaveMaxEfc);
  o.con = seg(kWaveMaxCon * kSimtConDoubles);
  o.residual = seg(nr);
  o.nodes = seg(P * nu);
  o.total = at;
  return o;
}

// per-lane view of the slab
#ifndef MJPCX_SIMT_PRIVATE
#define MJPCX_SIMT_PRIVATE 1  // 1: per-lane state in the hardware-swizzled private segment (scratch); 0: explicit global slab
#endif
constexpr int kSimtPrivateDoubles = 6144;
struct SimtData {
  double* slab;  // MJPCX_SIMT_PRIVATE: the lane's private array; else wave slab + lane (stride 64)
  SimtLayout o;
#if MJPCX_SIMT_PRIVATE
  __device__ __forceinline__ double& at(int off, int i) const { return slab[off + i]; }
  __device__ __forceinline__ void ld(double* dst, int off, int i, int n) const { for (int k = 0; k < n; k++) dst[k] = slab[off + i + k]; }
  __device__ __forceinline__ void st(const double* src, int off, int i, int n) const { for (int k = 0; k < n; k++) slab[off + i + k] = src[k]; }
#else
  __device__ __forceinline__ double& at(int off, int i) const { return slab[(size_t)(off + i) * 64]; }
  __device__ __forceinline__ void ld(double* dst, int off, int i, int n) const { for (int k = 0; k < n; k++) dst[k] = slab[(size_t)(off + i + k) * 64]; }
  __device__ __forceinline__ void st(const double* src, int off, int i, int n) const { for (int k = 0; k < n; k++) slab[(size_t)(off + i + k) * 64] = src[k]; }
#endif
};

struct SimtContact {  // registers while in use; packed into kSimtConDoubles doubles in the slab
  int g1, g2, dim, efc;
  double dist, margin, includemargin, mu, pos[3], frame[9], friction[5], solref[2], solimp[5];
};
__device__ __forceinline__ void simt_con_store(const SimtData& d, int ci, const SimtContact& c) {
  const int b = d.o.con + ci * kSimtConDoubles;
  d.at(b, 0) = c.dist; d.at(b, 1) = c.margin; d.at(b, 2) = c.includemargin; d.at(b, 3) = c.mu;
  d.st(c.pos, b, 4, 3); d.st(c.frame, b, 7, 9); d.st(c.friction, b, 16, 5); d.st(c.solref, b, 21, 2); d.st(c.solimp, b, 23, 5);
  d.at(b, 28) = c.g1; d.at(b, 29) = c.g2; d.at(b, 30) = c.dim; d.at(b, 31) = c.efc;
}
__device__ __forceinline__ void simt_con_load(const SimtData& d, int ci, SimtContact& c) {
  const int b = d.o.con + ci * kSimtConDoubles;
  c.dist = d.at(b, 0); c.margin = d.at(b, 1); c.includemargin = d.at(b, 2); c.mu = d.at(b, 3);
  d.ld(c.pos, b, 4, 3); d.ld(c.frame, b, 7, 9); d.ld(c.friction, b, 16, 5); d.ld(c.solref, b, 21, 2); d.ld(c.solimp, b, 23, 5);
  c.g1 = (int)d.at(b, 28); c.g2 = (int)d.at(b, 29); c.dim = (int)d.at(b, 30); c.efc = (int)d.at(b, 31);
}

// ---------------------------------------------------------------- position stage
__device__ __forceinline__ void simt_kinematics(const WaveModel& m, const WaveTask& tk, const SimtData& d) {
  const SimtLayout& o = d.o;
  {
    const double z3[3] = {0, 0, 0}, q1[4] = {1, 0, 0, 0}, I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    d.st(z3, o.xpos, 0, 3); d.st(q1, o.xquat, 0, 4); d.st(I, o.xmat, 0, 9); d.st(z3, o.xipos, 0, 3); d.st(I, o.ximat, 0, 9);
  }
  for (int i = 1; i < m.nbody; i++) {
    double xpos[3], xquat[4];
    const int pid = m.body_parentid[i], jn = m.body_jntnum[i], ja = m.body_jntadr[i];
    if (m.body_mocapid[i] >= 0) {
      const double* mp = tk.blob + tk.off_mocap + 7 * m.body_mocapid[i];
      for (int k = 0; k < 3; k++) xpos[k] = mp[k];
      for (int k = 0; k < 4; k++) xquat[k] = mp[3 + k];
      q_norm(xquat);
    } else if (jn == 1 && m.jnt_type[ja] == kJntFree) {
      const int qa = m.jnt_qposadr[ja];
      d.ld(xpos, o.qpos, qa, 3); d.ld(xquat, o.qpos, qa + 3, 4);
      q_norm(xquat);
      d.st(xpos, o.xanchor, 3 * ja, 3);
      const double ez[3] = {0, 0, 1};
      d.st(ez, o.xaxis, 3 * ja, 3);
    } else {
      double pm[9], pq[4], pp[3];
      d.ld(pm, o.xmat, 9 * pid, 9); d.ld(pq, o.xquat, 4 * pid, 4); d.ld(pp, o.xpos, 3 * pid, 3);
      mv3(xpos, pm, m.body_pos + 3 * i);
      for (int k = 0; k < 3; k++) xpos[k] += pp[k];
      q_mul(xquat, pq, m.body_quat + 4 * i);
      for (int j = ja; j < ja + jn; j++) {
        const int qa = m.jnt_qposadr[j], jt = m.jnt_type[j];
        double anchor[3], axis[3];
        q_rot(anchor, m.jnt_pos + 3 * j, xquat);
        for (int k = 0; k < 3; k++) anchor[k] += xpos[k];
        q_rot(axis, m.jnt_axis + 3 * j, xquat);
        if (jt == kJntSlide) {
          const double s = d.at(o.qpos, qa) - m.qpos0[qa];
          for (int k = 0; k < 3; k++) xpos[k] += axis[k] * s;
        } else if (jt == kJntBall || jt == kJntHinge) {
          double qloc[4], vec[3];
          if (jt == kJntBall) { d.ld(qloc, o.qpos, qa, 4); q_norm(qloc); }
          else aa2quat(qloc, m.jnt_axis + 3 * j, d.at(o.qpos, qa) - m.qpos0[qa]);
          q_mul(xquat, xquat, qloc);
          q_rot(vec, m.jnt_pos + 3 * j, xquat);
          for (int k = 0; k < 3; k++) xpos[k] = anchor[k] - vec[k];
        }
        d.st(anchor, o.xanchor, 3 * j, 3); d.st(axis, o.xaxis, 3 * j, 3);
      }
    }
    q_norm(xquat);
    double xmat[9], v[3], q[4], im[9], ip[3];
    q2mat(xmat, xquat);
    d.st(xpos, o.xpos, 3 * i, 3); d.st(xquat, o.xquat, 4 * i, 4); d.st(xmat, o.xmat, 9 * i, 9);
    mv3(v, xmat, m.body_ipos + 3 * i);
    for (int k = 0; k < 3; k++) ip[k] = xpos[k] + v[k];
    d.st(ip, o.xipos, 3 * i, 3);
    q_mul(q, xquat, m.body_iquat + 4 * i);
    q2mat(im, q);
    d.st(im, o.ximat, 9 * i, 9);
  }
  for (int s = 0; s < m.nsite; s++) {
    const int b = m.site_bodyid[s];
    double bm[9], bp[3], v[3];
    d.ld(bm, o.xmat, 9 * b, 9); d.ld(bp, o.xpos, 3 * b, 3);
    mv3(v, bm, m.site_pos + 3 * s);
    for (int k = 0; k < 3; k++) v[k] += bp[k];
    d.st(v, o.site_xpos, 3 * s, 3);
  }
}
__device__ __forceinline__ void simt_geom_pose(const WaveModel& m, const SimtData& d, int g, double* pos, double* mat) {
  const int b = m.geom_bodyid[g];
  double bm[9], bp[3], bq[4], v[3], q[4];
  d.ld(bm, d.o.xmat, 9 * b, 9); d.ld(bp, d.o.xpos, 3 * b, 3); d.ld(bq, d.o.xquat, 4 * b, 4);
  mv3(v, bm, m.geom_pos + 3 * g);
  for (int k = 0; k < 3; k++) pos[k] = bp[k] + v[k];
  q_mul(q, bq, m.geom_quat + 4 * g);
  q2mat(mat, q);
}

__device__ __forceinline__ void simt_compos(const WaveModel& m, const SimtData& d) {
  const SimtLayout& o = d.o;
  const int nb = m.nbody;
  for (int i = 0; i < nb; i++) {
    const double mi = m.body_mass[i];
    for (int k = 0; k < 3; k++) d.at(o.subtree_com, 3 * i + k) = mi * d.at(o.xipos, 3 * i + k);
  }
  for (int i = nb - 1; i > 0; i--) {
    const int p = m.body_parentid[i];
    for (int k = 0; k < 3; k++) d.at(o.subtree_com, 3 * p + k) += d.at(o.subtree_com, 3 * i + k);
  }
  for (int i = 0; i < nb; i++) {
    const double sm = m.body_subtreemass[i];
    for (int k = 0; k < 3; k++) d.at(o.subtree_com, 3 * i + k) = sm < kMinVal ? d.at(o.xipos, 3 * i + k) : d.at(o.subtree_com, 3 * i + k) / sm;
  }
  for (int k = 0; k < 10; k++) d.at(o.cinert, k) = 0;
  for (int i = 1; i < nb; i++) {
    double off[3], com[3], ip[3], im[9], ci[10];
    d.ld(com, o.subtree_com, 3 * m.body_rootid[i], 3); d.ld(ip, o.xipos, 3 * i, 3); d.ld(im, o.ximat, 9 * i, 9);
    for (int k = 0; k < 3; k++) off[k] = ip[k] - com[k];
    w_inert_com(ci, m.body_inertia + 3 * i, im, off, m.body_mass[i]);
    d.st(ci, o.cinert, 10 * i, 10);
  }
  for (int j = 0; j < m.njnt; j++) {
    const int b = m.jnt_bodyid[j], jt = m.jnt_type[j];
    int da = m.jnt_dofadr[j];
    double off[3], com[3], an[3], xmat[9];
    d.ld(com, o.subtree_com, 3 * m.body_rootid[b], 3); d.ld(an, o.xanchor, 3 * j, 3); d.ld(xmat, o.xmat, 9 * b, 9);
    for (int k = 0; k < 3; k++) off[k] = com[k] - an[k];
    if (jt == kJntFree) {
      for (int k = 0; k < 3; k++) {
        double c[6] = {0, 0, 0, 0, 0, 0};
        c[3 + k] = 1;
        d.st(c, o.cdof, 6 * (da + k), 6);
      }
      da += 3;
    }
    if (jt == kJntFree || jt == kJntBall) {
      for (int k = 0; k < 3; k++) {
        double c[6];
        const double ax[3] = {xmat[k], xmat[3 + k], xmat[6 + k]};
        for (int e = 0; e < 3; e++) c[e] = ax[e];
        cr3(c + 3, ax, off);
        d.st(c, o.cdof, 6 * (da + k), 6);
      }
    } else {
      double c[6], ax[3];
      d.ld(ax, o.xaxis, 3 * j, 3);
      if (jt == kJntSlide) { c[0] = c[1] = c[2] = 0; for (int e = 0; e < 3; e++) c[3 + e] = ax[e]; }
      else { for (int e = 0; e < 3; e++) c[e] = ax[e]; cr3(c + 3, ax, off); }
      d.st(c, o.cdof, 6 * da, 6);
    }
  }
}

__device__ __forceinline__ void simt_crb(const WaveModel& m, const SimtData& d) {
  const SimtLayout& o = d.o;
  const int nv = m.nv, nb = m.nbody;
  for (int e = 0; e < 10 * nb; e++) d.at(o.crb, e) = d.at(o.cinert, e);
  for (int i = nb - 1; i > 0; i--) {
    const int p = m.body_parentid[i];
    if (p > 0) for (int k = 0; k < 10; k++) d.at(o.crb, 10 * p + k) += d.at(o.crb, 10 * i + k);
  }
  for (int e = 0; e < nv * nv; e++) d.at(o.M, e) = 0;
  for (int i = 0; i < nv; i++) {
    double buf[6], crb[10], ci[6];
    d.ld(crb, o.crb, 10 * m.dof_bodyid[i], 10); d.ld(ci, o.cdof, 6 * i, 6);
    w_mul_inert(buf, crb, ci);
    d.at(o.M, i * nv + i) = m.dof_armature[i] + w_dot6(ci, buf);
    for (int j = m.dof_parentid[i]; j >= 0; j = m.dof_parentid[j]) {
      double cj[6];
      d.ld(cj, o.cdof, 6 * j, 6);
      const double v = w_dot6(cj, buf);
      d.at(o.M, i * nv + j) = v;
      d.at(o.M, j * nv + i) = v;
    }
  }
}

// dense Cholesky / solve on slab matrices (oracle chol_factor / chol_solve)
__device__ __forceinline__ bool simt_chol(const SimtData& d, int L, int A, int n) {
  bool ok = true;
  for (int j = 0; j < n; j++) {
    double s = d.at(A, j * n + j);
    for (int k = 0; k < j; k++) { const double l = d.at(L, j * n + k); s -= l * l; }
    if (!(s > kMinVal)) { ok = false; s = 1.0; }  // keep going with a harmless pivot: lanes diverge here, the flag decides
    s = sqrt(s);
    d.at(L, j * n + j) = s;
    for (int i = j + 1; i < n; i++) {
      double v = d.at(A, i * n + j);
      for (int k = 0; k < j; k++) v -= d.at(L, i * n + k) * d.at(L, j * n + k);
      d.at(L, i * n + j) = v / s;
    }
  }
  return ok;
}
__device__ __forceinline__ void simt_chol_solve(const SimtData& d, int x, int L, int n) {  // in place on slab vector x
  for (int i = 0; i < n; i++) {
    double v = d.at(x, i);
    for (int k = 0; k < i; k++) v -= d.at(L, i * n + k) * d.at(x, k);
    d.at(x, i) = v / d.at(L, i * n + i);
  }
  for (int i = n - 1; i >= 0; i--) {
    double v = d.at(x, i);
    for (int k = i + 1; k < n; k++) v -= d.at(L, k * n + i) * d.at(x, k);
    d.at(x, i) = v / d.at(L, i * n + i);
  }
}

// ---------------------------------------------------------------- collision (o_collision)
__device__ __forceinline__ void simt_add_contact(const SimtData& d, int& ncon, int& warning, SimtContact& proto, double dist, const double* pos,
                                                 const double* normal) {
  if (!(dist < proto.margin)) return;
  if (ncon >= kWaveMaxCon) { warning |= 32; return; }
  SimtContact c = proto;
  c.dist = dist;
  for (int k = 0; k < 3; k++) { c.pos[k] = pos[k]; c.frame[k] = normal[k]; }
  w_make_frame(c.frame);
  simt_con_store(d, ncon, c);
  ncon++;
}
__device__ __forceinline__ void simt_contact_param(const WaveModel& m, int g1, int g2, SimtContact& c) {
  WaveContact w;
  wf_contact_param(m, g1, g2, w);
  c.g1 = g1; c.g2 = g2; c.dim = w.dim; c.efc = 0; c.mu = 0; c.dist = 0;
  c.margin = w.margin; c.includemargin = w.includemargin;
  for (int k = 0; k < 5; k++) { c.friction[k] = w.friction[k]; c.solimp[k] = w.solimp[k]; }
  c.solref[0] = w.solref[0]; c.solref[1] = w.solref[1];
}
__device__ __forceinline__ void simt_collision(const WaveModel& m, const SimtData& d, int& ncon, int& warning) {
  ncon = 0;
  if (m.disableflags & (MJPCX_DSBL_CONSTRAINT | MJPCX_DSBL_CONTACT)) return;
  for (int si = 0; si < m.nstatic_geom; si++) {
    const int g1 = m.static_geom[si], t1 = m.geom_type[g1];
    if (t1 != MJPCX_GEOM_PLANE && t1 != MJPCX_GEOM_SPHERE && t1 != MJPCX_GEOM_BOX) continue;
    double p1[3], R1[9];
    simt_geom_pose(m, d, g1, p1, R1);
    for (int di = 0; di < m.ndynamic_geom; di++) {
      const int g2 = m.dynamic_geom[di];
      if (!((m.geom_contype[g1] & m.geom_conaffinity[g2]) || (m.geom_contype[g2] & m.geom_conaffinity[g1]))) continue;
      const int t2 = m.geom_type[g2];
      double p2[3], R2[9];
      simt_geom_pose(m, d, g2, p2, R2);
      const double* s2 = m.geom_size + 3 * g2;
      SimtContact proto;
      simt_contact_param(m, g1, g2, proto);
      if (t1 == MJPCX_GEOM_PLANE) {
        const double n[3] = {R1[2], R1[5], R1[8]};
        auto sphere_plane = [&](const double* c, double r) {
          const double dist = (c[0] - p1[0]) * n[0] + (c[1] - p1[1]) * n[1] + (c[2] - p1[2]) * n[2] - r;
          double pos[3];
          for (int k = 0; k < 3; k++) pos[k] = c[k] - n[k] * (r + 0.5 * dist);
          simt_add_contact(d, ncon, warning, proto, dist, pos, n);
        };
        if (t2 == MJPCX_GEOM_SPHERE) sphere_plane(p2, s2[0]);
        else if (t2 == MJPCX_GEOM_CAPSULE) {
          for (int sgn = -1; sgn <= 1; sgn += 2) {
            double c[3];
            for (int k = 0; k < 3; k++) c[k] = p2[k] + sgn * s2[1] * R2[3 * k + 2];
            sphere_plane(c, s2[0]);
          }
        } else if (t2 == MJPCX_GEOM_BOX) {
          int cnt = 0;
          for (int i = 0; i < 8 && cnt < 4; i++) {
            const double loc[3] = {(i & 1 ? s2[0] : -s2[0]), (i & 2 ? s2[1] : -s2[1]), (i & 4 ? s2[2] : -s2[2])};
            double c[3];
            mv3(c, R2, loc);
            for (int k = 0; k < 3; k++) c[k] += p2[k];
            const double dist = (c[0] - p1[0]) * n[0] + (c[1] - p1[1]) * n[1] + (c[2] - p1[2]) * n[2];
            if (dist < proto.margin) {
              double pos[3];
              for (int k = 0; k < 3; k++) pos[k] = c[k] - 0.5 * dist * n[k];
              simt_add_contact(d, ncon, warning, proto, dist, pos, n);
              cnt++;
            }
          }
        } else if (t2 == MJPCX_GEOM_CYLINDER) {
          const double a[3] = {R2[2], R2[5], R2[8]};
          const double pa = n[0] * a[0] + n[1] * a[1] + n[2] * a[2];
          const double sgn = pa > 0 ? -1.0 : 1.0;
          double v[3], vn = 0;
          for (int k = 0; k < 3; k++) { v[k] = -(n[k] - pa * a[k]); vn += v[k] * v[k]; }
          vn = sqrt(vn);
          if (vn < 1e-10) { v[0] = R2[0]; v[1] = R2[3]; v[2] = R2[6]; vn = 1; }
          for (int k = 0; k < 3; k++) v[k] /= vn;
          double w[3];
          cr3(w, a, v);
          const double cs[3] = {1.0, -0.5, -0.5}, sn[3] = {0.0, 0.8660254037844386, -0.8660254037844386};
          for (int i = 0; i < 4; i++) {
            const double side = i < 3 ? sgn : -sgn, cc = i < 3 ? cs[i] : 1.0, ss = i < 3 ? sn[i] : 0.0;
            double c[3], pos[3];
            for (int k = 0; k < 3; k++) c[k] = p2[k] + side * s2[1] * a[k] + s2[0] * (cc * v[k] + ss * w[k]);
            const double dist = (c[0] - p1[0]) * n[0] + (c[1] - p1[1]) * n[1] + (c[2] - p1[2]) * n[2];
            for (int k = 0; k < 3; k++) pos[k] = c[k] - 0.5 * dist * n[k];
            simt_add_contact(d, ncon, warning, proto, dist, pos, n);
          }
        }
      } else if (t1 == MJPCX_GEOM_SPHERE && t2 == MJPCX_GEOM_SPHERE) {
        double n[3], len = 0, pos[3];
        for (int k = 0; k < 3; k++) { n[k] = p2[k] - p1[k]; len += n[k] * n[k]; }
        len = sqrt(len);
        if (len < kMinVal) { n[0] = 1; n[1] = n[2] = 0; } else for (int k = 0; k < 3; k++) n[k] /= len;
        const double r1 = m.geom_size[3 * g1], dist = len - r1 - s2[0];
        for (int k = 0; k < 3; k++) pos[k] = p1[k] + n[k] * (r1 + 0.5 * dist);
        simt_add_contact(d, ncon, warning, proto, dist, pos, n);
      } else if (t1 == MJPCX_GEOM_BOX && t2 == MJPCX_GEOM_SPHERE) {
        const double* s1 = m.geom_size + 3 * g1;
        double rel[3], loc[3], clamped[3];
        for (int k = 0; k < 3; k++) rel[k] = p2[k] - p1[k];
        for (int k = 0; k < 3; k++) loc[k] = R1[k] * rel[0] + R1[3 + k] * rel[1] + R1[6 + k] * rel[2];
        bool inside = true;
        for (int k = 0; k < 3; k++) {
          clamped[k] = loc[k] < -s1[k] ? -s1[k] : (loc[k] > s1[k] ? s1[k] : loc[k]);
          if (clamped[k] != loc[k]) inside = false;
        }
        double nl[3] = {0, 0, 0}, dist;
        if (!inside) {
          double len = 0;
          for (int k = 0; k < 3; k++) { nl[k] = loc[k] - clamped[k]; len += nl[k] * nl[k]; }
          len = sqrt(len);
          for (int k = 0; k < 3; k++) nl[k] /= len;
          dist = len - s2[0];
        } else {
          int best = 0; double bd = 1e300;
          for (int k = 0; k < 3; k++) { const double dd = s1[k] - fabs(loc[k]); if (dd < bd) { bd = dd; best = k; } }
          nl[best] = loc[best] >= 0 ? 1 : -1;
          clamped[best] = nl[best] * s1[best];
          dist = -bd - s2[0];
        }
        double n[3], surf[3], pos[3];
        mv3(n, R1, nl);
        mv3(surf, R1, clamped);
        for (int k = 0; k < 3; k++) pos[k] = p1[k] + surf[k] + 0.5 * dist * n[k];
        simt_add_contact(d, ncon, warning, proto, dist, pos, n);
      }
    }
  }
}

// ---------------------------------------------------------------- velocity stage
__device__ __forceinline__ void simt_comvel(const WaveModel& m, const SimtData& d) {
  const SimtLayout& o = d.o;
  for (int k = 0; k < 6; k++) d.at(o.cvel, k) = 0;
  for (int i = 1; i < m.nbody; i++) {
    double cvel[6];
    d.ld(cvel, o.cvel, 6 * m.body_parentid[i], 6);
    for (int j = m.body_jntadr[i]; j < m.body_jntadr[i] + m.body_jntnum[i]; j++) {
      int da = m.jnt_dofadr[j];
      const int jt = m.jnt_type[j];
      if (jt == kJntFree) {
        for (int e = 0; e < 18; e++) d.at(o.cdof_dot, 6 * da + e) = 0;
        for (int k = 0; k < 3; k++) {
          const double qv = d.at(o.qvel, da + k);
          for (int c = 0; c < 6; c++) cvel[c] += d.at(o.cdof, 6 * (da + k) + c) * qv;
        }
        da += 3;
      }
      if (jt == kJntFree || jt == kJntBall) {
        for (int k = 0; k < 3; k++) {
          double cd[6], dot[6];
          d.ld(cd, o.cdof, 6 * (da + k), 6);
          w_cross_motion(dot, cvel, cd);
          d.st(dot, o.cdof_dot, 6 * (da + k), 6);
        }
        for (int k = 0; k < 3; k++) {
          const double qv = d.at(o.qvel, da + k);
          for (int c = 0; c < 6; c++) cvel[c] += d.at(o.cdof, 6 * (da + k) + c) * qv;
        }
      } else {
        double cd[6], dot[6];
        d.ld(cd, o.cdof, 6 * da, 6);
        w_cross_motion(dot, cvel, cd);
        d.st(dot, o.cdof_dot, 6 * da, 6);
        const double qv = d.at(o.qvel, da);
        for (int c = 0; c < 6; c++) cvel[c] += cd[c] * qv;
      }
    }
    d.st(cvel, o.cvel, 6 * i, 6);
  }
}

// ---------------------------------------------------------------- constraint rows (o_make_constraint_full)
__device__ __forceinline__ int simt_new_row(const SimtData& d, int& nefc, int& warning, int type, int id, int nv) {
  if (nefc >= kWaveMaxEfc) { warning |= 64; return -1; }
  const int r = nefc++;
  for (int k = 0; k < nv; k++) d.at(d.o.efc_J, r * nv + k) = 0;
  d.at(d.o.efc_type, r) = type; d.at(d.o.efc_id, r) = id;
  d.at(d.o.efc_pos, r) = 0; d.at(d.o.efc_margin, r) = 0; d.at(d.o.efc_floss, r) = 0;
  return r;
}
__device__ __forceinline__ void simt_make_constraint(const WaveModel& m, const SimtData& d, int ncon, int& nefc, int& warning) {
  const SimtLayout& o = d.o;
  const int nv = m.nv;
  nefc = 0;
  if (m.disableflags & MJPCX_DSBL_CONSTRAINT) return;
  if (!(m.disableflags & MJPCX_DSBL_FRICTIONLOSS))
    for (int i = 0; i < nv; i++)
      if (m.dof_frictionloss[i] > 0) {
        const int r = simt_new_row(d, nefc, warning, kEfcFriction, i, nv);
        if (r < 0) break;
        d.at(o.efc_J, r * nv + i) = 1;
        d.at(o.efc_floss, r) = m.dof_frictionloss[i];
      }
  if (!(m.disableflags & MJPCX_DSBL_LIMIT))
    for (int j = 0; j < m.njnt; j++) {
      if (!m.jnt_limited[j]) continue;
      if (m.jnt_type[j] != kJntSlide && m.jnt_type[j] != kJntHinge) continue;
      const double value = d.at(o.qpos, m.jnt_qposadr[j]), margin = m.jnt_margin[j];
      for (int side = -1; side <= 1; side += 2) {
        const double dist = side * (m.jnt_range[2 * j + (side + 1) / 2] - value);
        if (dist < margin) {
          const int r = simt_new_row(d, nefc, warning, kEfcLimit, j, nv);
          if (r < 0) break;
          d.at(o.efc_J, r * nv + m.jnt_dofadr[j]) = -side;
          d.at(o.efc_pos, r) = dist; d.at(o.efc_margin, r) = margin;
        }
      }
    }
  for (int ci = 0; ci < ncon; ci++) {
    SimtContact c;
    simt_con_load(d, ci, c);
    const int b2 = m.geom_bodyid[c.g2];
    int dim = c.dim;
    if (dim > 1 && m.cone != 1) { warning |= 128; dim = c.dim = 1; }
    c.efc = nefc;
    c.mu = c.friction[0] / sqrt(m.impratio > kMinVal ? m.impratio : 1.0);
    double com[3];
    d.ld(com, o.subtree_com, 3 * m.body_rootid[b2], 3);
    const double off[3] = {c.pos[0] - com[0], c.pos[1] - com[1], c.pos[2] - com[2]};
    const unsigned mask = m.body_dofmask[b2];
    for (int row = 0; row < dim; row++) {
      const int r = simt_new_row(d, nefc, warning, dim == 1 ? kEfcNormal : (row == 0 ? kEfcElliptic : kEfcConeRow), ci, nv);
      if (r < 0) { c.dim = row; break; }
      const double* ax = c.frame + 3 * (row < 3 ? row : row - 3);
      for (int k = 0; k < nv; k++) {
        if (!((mask >> k) & 1u)) continue;
        double cd[6];
        d.ld(cd, o.cdof, 6 * k, 6);
        double v;
        if (row < 3) {
          double lin[3];
          cr3(lin, cd, off);
          v = ax[0] * (cd[3] + lin[0]) + ax[1] * (cd[4] + lin[1]) + ax[2] * (cd[5] + lin[2]);
        } else {
          v = ax[0] * cd[0] + ax[1] * cd[1] + ax[2] * cd[2];
        }
        d.at(o.efc_J, r * nv + k) = v;
      }
      if (row == 0) { d.at(o.efc_pos, r) = c.dist; d.at(o.efc_margin, r) = c.includemargin; }
    }
    simt_con_store(d, ci, c);
  }
  for (int r = 0; r < nefc; r++) {
    const int type = (int)d.at(o.efc_type, r), id = (int)d.at(o.efc_id, r);
    double solref[2], solimp[5], diag;
    SimtContact c;
    if (type == kEfcFriction) { for (int k = 0; k < 2; k++) solref[k] = m.dof_solref[2 * id + k]; for (int k = 0; k < 5; k++) solimp[k] = m.dof_solimp[5 * id + k]; diag = m.dof_invweight0[id]; }
    else if (type == kEfcLimit) { for (int k = 0; k < 2; k++) solref[k] = m.jnt_solref[2 * id + k]; for (int k = 0; k < 5; k++) solimp[k] = m.jnt_solimp[5 * id + k]; diag = m.dof_invweight0[m.jnt_dofadr[id]]; }
    else {
      simt_con_load(d, id, c);
      solref[0] = c.solref[0]; solref[1] = c.solref[1];
      for (int k = 0; k < 5; k++) solimp[k] = c.solimp[k];
      diag = m.body_invweight0[2 * m.geom_bodyid[c.g1]] + m.body_invweight0[2 * m.geom_bodyid[c.g2]];
    }
    double vel = 0;
    for (int k = 0; k < nv; k++) vel += d.at(o.efc_J, r * nv + k) * d.at(o.qvel, k);
    double kk, bb;
    w_solref_kb(m, solref, solimp, kk, bb);
    if (type == kEfcConeRow) {
      const double f = c.friction[r - c.efc - 1];
      d.at(o.efc_R, r) = d.at(o.efc_R, c.efc) * (c.mu * c.mu) / (f * f);
      d.at(o.efc_aref, r) = -bb * vel;
    } else {
      const double pos = d.at(o.efc_pos, r) - d.at(o.efc_margin, r);
      const double imp = w_impedance(solimp, pos);
      const double R = (1 - imp) / imp * diag;
      d.at(o.efc_R, r) = R < kMinVal ? kMinVal : R;
      d.at(o.efc_aref, r) = -bb * vel - kk * imp * pos;
    }
    d.at(o.efc_D, r) = 1.0 / d.at(o.efc_R, r);
  }
}

// ---------------------------------------------------------------- smooth forces (o_passive, o_rne, o_actuation)
__device__ __forceinline__ void simt_smooth(const WaveModel& m, const SimtData& d, int& warning) {
  const SimtLayout& o = d.o;
  const int nv = m.nv, nb = m.nbody, nu = m.nu;
  for (int i = 0; i < nv; i++) d.at(o.qfrc_passive, i) = 0;
  if (!(m.disableflags & MJPCX_DSBL_PASSIVE)) {
    for (int j = 0; j < m.njnt; j++) {
      const double k = m.jnt_stiffness[j];
      if (k == 0) continue;
      const int qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j];
      if (m.jnt_type[j] == kJntSlide || m.jnt_type[j] == kJntHinge) d.at(o.qfrc_passive, da) -= k * (d.at(o.qpos, qa) - m.qpos_spring[qa]);
    }
    for (int i = 0; i < nv; i++) d.at(o.qfrc_passive, i) -= m.dof_damping[i] * d.at(o.qvel, i);
  }
  for (int k = 0; k < 6; k++) { d.at(o.cacc, k) = (k >= 3 && !(m.disableflags & MJPCX_DSBL_GRAVITY)) ? -m.gravity[k - 3] : 0.0; d.at(o.cfrc, k) = 0; }
  for (int i = 1; i < nb; i++) {
    double cacc[6], t1[6], t2[6], t3[6], ci[10], cv[6];
    d.ld(cacc, o.cacc, 6 * m.body_parentid[i], 6);
    const int da = m.body_dofadr[i];
    for (int k = da; k >= 0 && k < da + m.body_dofnum[i]; k++) {
      const double qv = d.at(o.qvel, k);
      for (int c = 0; c < 6; c++) cacc[c] += d.at(o.cdof_dot, 6 * k + c) * qv;
    }
    d.st(cacc, o.cacc, 6 * i, 6);
    d.ld(ci, o.cinert, 10 * i, 10); d.ld(cv, o.cvel, 6 * i, 6);
    w_mul_inert(t1, ci, cacc);
    w_mul_inert(t2, ci, cv);
    w_cross_force(t3, cv, t2);
    for (int c = 0; c < 6; c++) d.at(o.cfrc, 6 * i + c) = t1[c] + t3[c];
  }
  for (int i = nb - 1; i > 0; i--) {
    const int p = m.body_parentid[i];
    if (p > 0) for (int c = 0; c < 6; c++) d.at(o.cfrc, 6 * p + c) += d.at(o.cfrc, 6 * i + c);
  }
  for (int k = 0; k < nv; k++) {
    double cd[6], cf[6];
    d.ld(cd, o.cdof, 6 * k, 6); d.ld(cf, o.cfrc, 6 * m.dof_bodyid[k], 6);
    d.at(o.qfrc_bias, k) = w_dot6(cd, cf);
  }
  for (int i = 0; i < nv; i++) d.at(o.qfrc_actuator, i) = 0;
  for (int i = 0; i < nu; i++) d.at(o.actuator_force, i) = 0;
  if (!(m.disableflags & MJPCX_DSBL_ACTUATION)) {
    bool bad = false;
    for (int i = 0; i < nu; i++) bad |= is_bad(d.at(o.ctrl, i));
    if (bad) { warning |= 8; for (int i = 0; i < nu; i++) d.at(o.ctrl, i) = 0; }
    for (int i = 0; i < nu; i++) {
      double ctrl = d.at(o.ctrl, i);
      if (m.actuator_ctrllimited[i] && !(m.disableflags & MJPCX_DSBL_CLAMPCTRL)) ctrl = clampv(ctrl, m.actuator_ctrlrange[2 * i], m.actuator_ctrlrange[2 * i + 1]);
      const int j = m.actuator_trnid[i], qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j];
      const double gear = m.actuator_gear[i];
      double force = m.actuator_gainprm[3 * i] * ctrl;
      if (m.actuator_biastype[i] == 1)
        force += m.actuator_biasprm[3 * i] + m.actuator_biasprm[3 * i + 1] * gear * d.at(o.qpos, qa) + m.actuator_biasprm[3 * i + 2] * gear * d.at(o.qvel, da);
      if (m.actuator_forcelimited[i]) force = clampv(force, m.actuator_forcerange[2 * i], m.actuator_forcerange[2 * i + 1]);
      d.at(o.actuator_force, i) = force;
      d.at(o.qfrc_actuator, da) += gear * force;
    }
  }
  for (int i = 0; i < nv; i++) {
    const double f = d.at(o.qfrc_passive, i) - d.at(o.qfrc_bias, i) + d.at(o.qfrc_actuator, i);
    d.at(o.qfrc_smooth, i) = f;
    d.at(o.qacc_smooth, i) = f;
  }
}

// ---------------------------------------------------------------- Newton solver (o_constraint_newton)
// penalty of all rows at jar (+ alpha jv): cost, optional forces/zones, directional derivatives
template <bool WRITE, bool LINE>
__device__ __forceinline__ double simt_rows(const SimtData& d, int nefc, double alpha, double& g1, double& h2) {
  const SimtLayout& o = d.o;
  double cost = 0;
  g1 = 0; h2 = 0;
  for (int r = 0; r < nefc; r++) {
    const int type = (int)d.at(o.efc_type, r);
    const double D = d.at(o.efc_D, r);
    const double v = LINE ? d.at(o.jv, r) : 0.0;
    const double x = d.at(o.jar, r) + (LINE ? alpha * v : 0.0);
    if (type == kEfcFriction) {
      const double f = d.at(o.efc_floss, r), R = d.at(o.efc_R, r);
      if (x <= -R * f) { cost += -0.5 * R * f * f - f * x; g1 += -f * v; if (WRITE) { d.at(o.efc_force, r) = f; d.at(o.efc_zone, r) = kZoneTop; } }
      else if (x >= R * f) { cost += -0.5 * R * f * f + f * x; g1 += f * v; if (WRITE) { d.at(o.efc_force, r) = -f; d.at(o.efc_zone, r) = kZoneTop; } }
      else { cost += 0.5 * D * x * x; g1 += D * x * v; h2 += D * v * v; if (WRITE) { d.at(o.efc_force, r) = -D * x; d.at(o.efc_zone, r) = kZoneBottom; } }
    } else if (type == kEfcLimit || type == kEfcNormal) {
      if (x < 0) { cost += 0.5 * D * x * x; g1 += D * x * v; h2 += D * v * v; if (WRITE) { d.at(o.efc_force, r) = -D * x; d.at(o.efc_zone, r) = kZoneBottom; } }
      else if (WRITE) { d.at(o.efc_force, r) = 0; d.at(o.efc_zone, r) = kZoneTop; }
    } else if (type == kEfcElliptic) {
      const int cb = o.con + (int)d.at(o.efc_id, r) * kSimtConDoubles;
      const int dim = (int)d.at(cb, 30);
      const double mu = d.at(cb, 3);
      double fr[5], U[6], V[6], X[6], T = 0;
      d.ld(fr, cb, 16, 5);
      X[0] = x; U[0] = x * mu; V[0] = v * mu;
      for (int j = 1; j < dim; j++) {
        const double vj = LINE ? d.at(o.jv, r + j) : 0.0;
        X[j] = d.at(o.jar, r + j) + (LINE ? alpha * vj : 0.0);
        U[j] = X[j] * fr[j - 1];
        V[j] = vj * fr[j - 1];
        T += U[j] * U[j];
      }
      T = sqrt(T);
      const double N = U[0];
      if (N >= mu * T || (T <= 0 && N >= 0)) {
        if (WRITE) { for (int j = 0; j < dim; j++) d.at(o.efc_force, r + j) = 0; d.at(o.efc_zone, r) = kZoneTop; }
      } else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
        for (int j = 0; j < dim; j++) {
          const double Dj = d.at(o.efc_D, r + j), vj = LINE ? d.at(o.jv, r + j) : 0.0;
          cost += 0.5 * Dj * X[j] * X[j]; g1 += Dj * X[j] * vj; h2 += Dj * vj * vj;
          if (WRITE) d.at(o.efc_force, r + j) = -Dj * X[j];
        }
        if (WRITE) d.at(o.efc_zone, r) = kZoneBottom;
      } else {
        const double Dm = D / (mu * mu * (1 + mu * mu)), NT = N - mu * T;
        cost += 0.5 * Dm * NT * NT;
        if (LINE) {
          double UV = 0, VV = 0;
          for (int j = 1; j < dim; j++) { UV += U[j] * V[j]; VV += V[j] * V[j]; }
          const double dNT = V[0] - mu * UV / T;
          const double d2NT = -mu * (VV / T - UV * UV / (T * T * T));
          g1 += Dm * NT * dNT;
          h2 += Dm * (dNT * dNT + NT * d2NT);
        }
        if (WRITE) {
          d.at(o.efc_force, r) = -Dm * NT * mu;
          for (int j = 1; j < dim; j++) d.at(o.efc_force, r + j) = Dm * NT * mu * U[j] * fr[j - 1] / T;
          d.at(o.efc_zone, r) = kZoneMiddle;
        }
      }
      r += dim - 1;
    }
  }
  return cost;
}

__device__ __forceinline__ void simt_hessian(const WaveModel& m, const SimtData& d, int nefc) {
  const SimtLayout& o = d.o;
  const int nv = m.nv;
  for (int e = 0; e < nv * nv; e++) d.at(o.H, e) = d.at(o.M, e);
  for (int r = 0; r < nefc; r++) {
    const int t = (int)d.at(o.efc_type, r);
    if (t == kEfcFriction || t == kEfcLimit || t == kEfcNormal) {
      if ((int)d.at(o.efc_zone, r) != kZoneBottom) continue;
      const double D = d.at(o.efc_D, r);
      for (int a = 0; a < nv; a++) {
        const double ja = d.at(o.efc_J, r * nv + a);
        if (ja == 0) continue;
        for (int b = 0; b < nv; b++) d.at(o.H, a * nv + b) += D * ja * d.at(o.efc_J, r * nv + b);
      }
    } else if (t == kEfcElliptic) {
      const int cb = o.con + (int)d.at(o.efc_id, r) * kSimtConDoubles;
      const int dim = (int)d.at(cb, 30);
      const int zone = (int)d.at(o.efc_zone, r);
      if (zone != kZoneTop) {
        double Hc[36];
        for (int e = 0; e < 36; e++) Hc[e] = 0;
        if (zone == kZoneBottom) {
          for (int j = 0; j < dim; j++) Hc[j * dim + j] = d.at(o.efc_D, r + j);
        } else {
          const double mu = d.at(cb, 3);
          double U[6], s[6], T = 0, fr[5];
          d.ld(fr, cb, 16, 5);
          s[0] = mu; U[0] = d.at(o.jar, r) * mu;
          for (int j = 1; j < dim; j++) { s[j] = fr[j - 1]; U[j] = d.at(o.jar, r + j) * s[j]; T += U[j] * U[j]; }
          T = sqrt(T);
          const double Dm = d.at(o.efc_D, r) / (mu * mu * (1 + mu * mu)), NT = U[0] - mu * T;
          Hc[0] = Dm;
          for (int j = 1; j < dim; j++) {
            Hc[j] = Hc[j * dim] = -Dm * mu * U[j] / T;
            for (int k = 1; k < dim; k++)
              Hc[j * dim + k] = Dm * mu * mu * U[j] * U[k] / (T * T) - Dm * NT * mu * ((j == k ? 1.0 / T : 0.0) - U[j] * U[k] / (T * T * T));
          }
          for (int j = 0; j < dim; j++) for (int k = 0; k < dim; k++) Hc[j * dim + k] *= s[j] * s[k];
        }
        for (int j = 0; j < dim; j++)
          for (int k = 0; k < dim; k++) {
            const double w = Hc[j * dim + k];
            if (w == 0) continue;
            for (int a = 0; a < nv; a++) {
              const double ja = d.at(o.efc_J, (r + j) * nv + a);
              if (ja == 0) continue;
              for (int b = 0; b < nv; b++) d.at(o.H, a * nv + b) += w * ja * d.at(o.efc_J, (r + k) * nv + b);
            }
          }
      }
      r += dim - 1;
    }
  }
}

__device__ __forceinline__ void simt_newton(const WaveModel& m, const SimtData& d, int nefc, int& warning, bool have_warm) {
  const SimtLayout& o = d.o;
  const int nv = m.nv, ne = nefc;
  for (int i = 0; i < nv; i++) { d.at(o.qfrc_constraint, i) = 0; d.at(o.qacc, i) = d.at(o.qacc_smooth, i); }
  if (ne == 0) return;
  for (int r = 0; r < ne; r++) {
    double s = -d.at(o.efc_aref, r);
    for (int k = 0; k < nv; k++) s += d.at(o.efc_J, r * nv + k) * d.at(o.qacc, k);
    d.at(o.jar, r) = s;
  }
  double g1, h2;
  double cost = simt_rows<true, false>(d, ne, 0.0, g1, h2);
  if (have_warm) {
    double gauss = 0;
    for (int r = 0; r < ne; r++) {
      double s = -d.at(o.efc_aref, r);
      for (int k = 0; k < nv; k++) s += d.at(o.efc_J, r * nv + k) * d.at(o.qacc_warm, k);
      d.at(o.jv, r) = d.at(o.jar, r);  // keep the smooth-start residual
      d.at(o.jar, r) = s;
    }
    for (int a = 0; a < nv; a++) {
      double s = 0;
      for (int b = 0; b < nv; b++) s += d.at(o.M, a * nv + b) * (d.at(o.qacc_warm, b) - d.at(o.qacc_smooth, b));
      gauss += 0.5 * s * (d.at(o.qacc_warm, a) - d.at(o.qacc_smooth, a));
    }
    const double cw = gauss + simt_rows<true, false>(d, ne, 0.0, g1, h2);
    if (cw < cost) {
      cost = cw;
      for (int a = 0; a < nv; a++) d.at(o.qacc, a) = d.at(o.qacc_warm, a);
    } else {
      for (int r = 0; r < ne; r++) d.at(o.jar, r) = d.at(o.jv, r);
      simt_rows<true, false>(d, ne, 0.0, g1, h2);
    }
  }
  const double scale = 1.0 / (m.meaninertia * (nv > 1 ? nv : 1));
  double improvement = 0;
  // every lane runs the same number of trips as the slowest lane of its wavefront would anyway (SIMT); a finished
  // lane just stops updating
  for (int iter = 0; iter < m.solver_iterations; iter++) {
    double gnorm = 0;
    for (int a = 0; a < nv; a++) {
      double s = 0;
      for (int b = 0; b < nv; b++) s += d.at(o.M, a * nv + b) * (d.at(o.qacc, b) - d.at(o.qacc_smooth, b));
      d.at(o.Ma, a) = s;
      double g = s;
      for (int r = 0; r < ne; r++) g -= d.at(o.efc_J, r * nv + a) * d.at(o.efc_force, r);
      d.at(o.grad, a) = g;
      gnorm += g * g;
    }
    gnorm = sqrt(gnorm);
    if (gnorm == 0) break;
    if (iter > 0 && (scale * improvement < m.solver_tolerance || scale * gnorm < m.solver_tolerance)) break;  // MuJoCo's order
    simt_hessian(m, d, ne);
    if (!simt_chol(d, o.L, o.H, nv)) { warning |= 16; break; }
    for (int a = 0; a < nv; a++) d.at(o.search, a) = -d.at(o.grad, a);
    simt_chol_solve(d, o.search, o.L, nv);
    for (int r = 0; r < ne; r++) {
      double s = 0;
      for (int k = 0; k < nv; k++) s += d.at(o.efc_J, r * nv + k) * d.at(o.search, k);
      d.at(o.jv, r) = s;
    }
    double q1 = 0, q2 = 0;
    for (int a = 0; a < nv; a++) {
      const double sa = d.at(o.search, a);
      q1 += sa * d.at(o.Ma, a);
      double s = 0;
      for (int b = 0; b < nv; b++) s += d.at(o.M, a * nv + b) * d.at(o.search, b);
      q2 += sa * s;
    }
    double lo = 0, hi = -1, alpha = 0, d1, d2;
    simt_rows<false, true>(d, ne, 0.0, d1, d2);
    d1 += q1; d2 += q2;
    const double d10 = fabs(d1);
    double snorm = 0;
    for (int a = 0; a < nv; a++) snorm += d.at(o.search, a) * d.at(o.search, a);
    const double gtol = m.solver_tolerance * kLsTolerance * sqrt(snorm) / scale;
    for (int ls = 0; ls < 50 && d10 >= gtol; ls++) {
      double an = alpha - d1 / d2;
      if (!(an > lo) || (hi >= 0 && !(an < hi))) an = hi >= 0 ? 0.5 * (lo + hi) : 2 * alpha + 1;
      if (an == alpha) break;
      alpha = an;
      simt_rows<false, true>(d, ne, alpha, d1, d2);
      d1 += q1 + alpha * q2; d2 += q2;
      if (fabs(d1) < gtol) break;
      if (d1 < 0) lo = alpha; else hi = alpha;
    }
    for (int a = 0; a < nv; a++) d.at(o.qacc, a) += alpha * d.at(o.search, a);
    for (int r = 0; r < ne; r++) d.at(o.jar, r) += alpha * d.at(o.jv, r);
    double gauss = 0;
    for (int a = 0; a < nv; a++) {
      double s = 0;
      for (int b = 0; b < nv; b++) s += d.at(o.M, a * nv + b) * (d.at(o.qacc, b) - d.at(o.qacc_smooth, b));
      gauss += 0.5 * s * (d.at(o.qacc, a) - d.at(o.qacc_smooth, a));
    }
    const double newcost = gauss + simt_rows<true, false>(d, ne, 0.0, g1, h2);
    improvement = cost - newcost;
    cost = newcost;
  }
  for (int c = 0; c < nv; c++) {
    double s = 0;
    for (int r = 0; r < ne; r++) s += d.at(o.efc_J, r * nv + c) * d.at(o.efc_force, r);
    d.at(o.qfrc_constraint, c) = s;
  }
}

// ---------------------------------------------------------------- mj_forward / Euler
__device__ __forceinline__ void simt_forward(const WaveModel& m, const WaveTask& tk, const SimtData& d, int& ncon, int& nefc, int& warning, bool have_warm) {
  const SimtLayout& o = d.o;
  const int nv = m.nv;
  simt_kinematics(m, tk, d);
  simt_compos(m, d);
  simt_crb(m, d);
  if (!simt_chol(d, o.L, o.M, nv)) warning |= 16;
  // (the factor of M stays in L until qacc_smooth is solved below; Newton then reuses L for the factor of H)
  simt_collision(m, d, ncon, warning);
  simt_comvel(m, d);
  simt_make_constraint(m, d, ncon, nefc, warning);
  simt_smooth(m, d, warning);
  simt_chol_solve(d, o.qacc_smooth, o.L, nv);
  simt_newton(m, d, nefc, warning, have_warm);
}

__device__ __forceinline__ void simt_euler(const WaveModel& m, const SimtData& d, double& time) {
  const SimtLayout& o = d.o;
  const int nv = m.nv;
  const double h = m.timestep;
  if (m.any_damping && !(m.disableflags & MJPCX_DSBL_EULERDAMP)) {
    for (int e = 0; e < nv * nv; e++) d.at(o.H, e) = d.at(o.M, e);
    for (int i = 0; i < nv; i++) { d.at(o.H, i * nv + i) += h * m.dof_damping[i]; d.at(o.tmpv, i) = d.at(o.qfrc_smooth, i) + d.at(o.qfrc_constraint, i); }
    if (simt_chol(d, o.L, o.H, nv)) simt_chol_solve(d, o.tmpv, o.L, nv);
    else for (int i = 0; i < nv; i++) d.at(o.tmpv, i) = d.at(o.qacc, i);
  } else {
    for (int i = 0; i < nv; i++) d.at(o.tmpv, i) = d.at(o.qacc, i);
  }
  for (int i = 0; i < nv; i++) d.at(o.qvel, i) += h * d.at(o.tmpv, i);
  for (int j = 0; j < m.njnt; j++) {
    int qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j];
    const int jt = m.jnt_type[j];
    if (jt == kJntFree) {
      for (int k = 0; k < 3; k++) d.at(o.qpos, qa + k) += h * d.at(o.qvel, da + k);
      qa += 3; da += 3;
    }
    if (jt == kJntFree || jt == kJntBall) {
      double ax[3] = {d.at(o.qvel, da), d.at(o.qvel, da + 1), d.at(o.qvel, da + 2)};
      const double n = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
      if (n < kMinVal) { ax[0] = 1; ax[1] = ax[2] = 0; }
      else { ax[0] /= n; ax[1] /= n; ax[2] /= n; }
      double qrot[4], q[4];
      aa2quat(qrot, ax, h * n);
      d.ld(q, o.qpos, qa, 4);
      q_norm(q);
      q_mul(q, q, qrot);
      d.st(q, o.qpos, qa, 4);
    } else {
      d.at(o.qpos, qa) += h * d.at(o.qvel, da);
    }
  }
  time += h;
}

}  // namespace mjpcx

#include "simt_kernel.h"
