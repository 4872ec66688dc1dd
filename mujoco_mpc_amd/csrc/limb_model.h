// limb_model.h -- model constants of the LIMB kernel family (limb_step.h, limb_kernel.h): FOUR LANES PER CANDIDATE, one per LIMB of a
// floating-base biped -- the Humanoid of BASELINE configs[3] (mjpc/tasks/humanoid/tracking): a trunk CHAIN (torso with the free joint,
// lower waist with two hinges, pelvis with one: 9 dofs, replicated in the four lanes) and four limb chains of at most three jointed bodies
// with (3, 1, 2) hinges -- the legs (thigh, shin, foot) use all six slots, the arms (upper arm with two hinges, lower arm with one) three.
// Sixteen candidates share a wavefront. The inertia matrix of such a tree is a (nested) ARROWHEAD: four 6 x 6 limb blocks, coupled only
// through the 9 x 9 trunk block; limits, the hamstring tendons and contacts with the floor keep that shape, and the few contacts between
// two moving bodies (a hand on a thigh) enter the Newton Hessian as rank-one terms through the Woodbury identity (limb_step.h).
//
// Bodies without joints (head, hands, heels, toes) are FOLDED into the jointed body they are welded to when the image is built: mass,
// centre of mass and inertia tensor combined in that body's frame, their geoms and sites re-expressed there (the same rigid body; the
// oracle carries them as separate bodies, which differs by rounding only).
//
// Plain C++ (no HIP): the host builds the image once per context (mjpcx_create) and the CPU emulator of the kernel (tests/limbemu, test
// infrastructure) builds the same one. R is the working precision of the kernel (float for configs[3]; double for the parity tests).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/mjpcx.h"
#include "pair_cull.h"

namespace mjpcx { namespace limb {

constexpr int kLimbs = 4;
constexpr int kLB = 3, kLD = 6;     // jointed bodies / hinge slots of a limb: slots 0..2 on body 0, slot 3 on body 1, slots 4..5 on body 2
constexpr int kTB = 3, kTD = 9;     // trunk chain: dofs 0..5 the free joint (body 0), 6..7 hinges of body 1, 8 the hinge of body 2
constexpr int kLG = 4;              // collidable geoms on a limb's own bodies
constexpr int kTGL = 2;             // trunk geoms dealt to a lane for the test against the floor
constexpr int kLS = 4;              // tracking sites per lane (the limb's own and the trunk's, dealt)
constexpr int kNG = 24;             // moving geoms in the candidate's shared pose table
constexpr int kMaxPC = 8;           // contacts with the floor per lane and step
constexpr int kMaxX = 8;            // contacts between two moving geoms per candidate and step
constexpr int kMaxPair = 192, kMaxPSet = 8, kMaxResid = 160, kMaxTerm = 32, kMaxTrace = 2;
constexpr int kPairsPerLane = kMaxPair / kLimbs;
inline constexpr int slot_body(int j) { return j < 3 ? 0 : (j < 4 ? 1 : 2); }
inline constexpr int trunk_dof_body(int k) { return k < 6 ? 0 : (k < 8 ? 1 : 2); }

template <typename R> struct LBodyT { R pos[3], quat[4], ipos[3], inertia[6], mass; };  // inertia: xx yy zz xy xz yz about ipos, body axes
template <typename R> struct LJointT {
  R pos[3], axis[3];
  R qpos0, qspring, stiffness, damping, armature;
  R range[2], margin, invw, lim_k, lim_b, lim_imp[5];
  R gear_gain;        // gear * gainprm[0] of the joint's motor (0: none)
  R ctrl_lo, ctrl_hi;
  int on, limited, dof, qadr, act;
};
template <typename R> struct LGeomT {
  R pos[3], axis[3], radius, half;  // in the jointed body's frame; half = 0: sphere
  R pdiag, pmu;                     // against the floor: invweight0 sum of the pair, regularised sliding friction
  int on, body, gslot, pset, pdim;  // body: limb body slot 0..2 (a limb's geom) | trunk body 0..2 (a dealt trunk geom); pset: LPSetT index, pdim: condim (0: no pair)
};
template <typename R> struct LPSetT { R margin, includemargin, k, b, imp[5]; };
template <typename R> struct LSiteT { R pos[3]; int on, body, marker, mocap, tpos, tvel; };  // tpos / tvel: the cost terms of the marker's position / velocity entries  // body: 0..2 limb body, 3..5 trunk body
template <typename R> struct LTendonT { R coef[2], range[2], margin, invw, k, b, imp[5]; int on, slot[2]; };
template <typename R> struct LPairT { R diag, reach; unsigned char ga, gb, pset, pad; };
template <typename R> struct LTraceT { R pos[3]; int lane, body; };  // lane 0..3 + limb body slot, or lane 4 + trunk body

template <typename R> struct LimbT {
  LBodyT<R> body[kLB];
  LJointT<R> jnt[kLD];
  LGeomT<R> geom[kLG];
  LGeomT<R> tgeom[kTGL];
  LSiteT<R> site[kLS];
  LTendonT<R> tendon;
  int attach, nanc;      // the trunk body the limb hangs on; trunk dofs that move it (6, 8 or 9)
  int npair, pair0;      // the lane's share of the moving-geom pairs: pair[pair0 .. pair0 + npair)
  int tact[3];           // the trunk actuators whose spline this lane evaluates (ctrl index or -1), by trunk hinge 0..2
  int owns_trunk_rows;   // 1 in the lane that records the trunk's entries (state, waist joints' residual entries, the averages)
};
template <typename R> struct LimbModelT {
  LimbT<R> limb[kLimbs];
  LBodyT<R> tbody[kTB];
  LJointT<R> tjnt[3];    // trunk hinges: dofs 6, 7 (body 1), 8 (body 2)
  R tdamp[6], tarm[6];   // damping / armature of the free joint's dofs
  R plane_pos[3], plane_n[3], plane_t1[3], plane_t2[3];
  R gravity[3], timestep, tolerance, meaninertia;
  R grad[kNG], ghalf[kNG];          // radius, half length of the moving geoms by slot
  unsigned char glane[kNG], gbody[kNG];  // owner: lane 0..3 (gbody: limb body slot) or 4 (gbody: trunk body)
  LPSetT<R> pset[kMaxPSet];
  LPairT<R> pair[kMaxPair];
  LTraceT<R> trace[kMaxTrace];
  unsigned char term_of[kMaxResid];
  int term_norm[kMaxTerm];
  int iterations, nv, nq, nu, nr, nterm, ntrace, nmocap, ngeom, npair;
  int t_qvel, t_ctrl, t_avg;        // the cost terms of the joint-velocity, control and marker-average entries
  int key_start, key_last;          // residual_int[0..1] at build time (the kernel reads the current ones from the plan blob)
  int nattach[kTB];                 // limbs hanging on each trunk body
};

// one record per model: what the two precisions' images are cast from
typedef LimbModelT<double> LimbModelD;

namespace detail {
inline void q2m(double* m, const double* q) {
  const double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  const double q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3], q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2 * (q12 - q03); m[2] = 2 * (q13 + q02); m[3] = 2 * (q12 + q03);
  m[5] = 2 * (q23 - q01); m[6] = 2 * (q13 - q02); m[7] = 2 * (q23 + q01);
}
inline void qmul(double* r, const double* a, const double* b) {
  const double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  const double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
inline void mv(double* r, const double* m, const double* v) {
  const double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2], z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
inline double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
inline void digest_solimp(double* out, const double* s) {
  out[0] = clipd(s[0], 0.0001, 0.9999); out[1] = clipd(s[1], 0.0001, 0.9999); out[2] = s[2]; out[3] = clipd(s[3], 0.0001, 0.9999); out[4] = s[4] < 1 ? 1 : s[4];
}
inline void solref_kb(const mjpcx_model* m, const double* solref, const double* solimp, double* k, double* b) {
  const double dmax = clipd(solimp[1], 0.0001, 0.9999);
  if (solref[0] > 0) {
    double tc = solref[0];
    if (!(m->disableflags & MJPCX_DSBL_REFSAFE) && tc < 2 * m->timestep) tc = 2 * m->timestep;
    *k = 1.0 / (dmax * dmax * tc * tc * solref[1] * solref[1]); *b = 2.0 / (dmax * tc);
  } else { *k = -solref[0] / (dmax * dmax); *b = -solref[1] / dmax; }
}
// pose of body `b` in the frame of its ancestor `anc` (both welded together: no joints between): pos, rotation matrix
inline void rel_pose(const mjpcx_model* m, int b, int anc, double* pos, double* quat) {
  double p[3] = {0, 0, 0}, q[4] = {1, 0, 0, 0};
  std::vector<int> chain;
  for (int x = b; x != anc; x = m->body_parentid[x]) chain.push_back(x);
  for (int i = (int)chain.size() - 1; i >= 0; i--) {
    const int x = chain[i];
    double R[9], v[3], q2[4];
    q2m(R, q);
    mv(v, R, m->body_pos + 3 * x);
    for (int k = 0; k < 3; k++) p[k] += v[k];
    qmul(q2, q, m->body_quat + 4 * x);
    memcpy(q, q2, sizeof q2);
  }
  memcpy(pos, p, sizeof p); memcpy(quat, q, sizeof q);
}
}  // namespace detail

// Builds the limb kernel's view of a model + task. Returns "" on success, otherwise why the model is outside the class the kernel covers
// (the caller then keeps the wavefront-per-candidate kernels: not an error).
inline std::string limb_build(const mjpcx_model* m, const mjpcx_task* task, LimbModelD* out) {
  using namespace detail;
  LimbModelD& L = *out;
  memset(&L, 0, sizeof L);
  if (!task || task->residual_id != MJPCX_RESIDUAL_HUMANOID_TRACK) return "the limb kernel carries the humanoid tracking residual only";
  if (m->integrator != MJPCX_INT_EULER) return "integrator other than Euler";
  if (m->disableflags != 0) return "disable flags set";
  if (m->na != 0 || m->nuserdata != 0) return "activations / userdata";
  if (m->nv > kTD + kLimbs * kLD) return "more dofs than a 9-dof trunk and four 6-dof limbs";
  if (!m->body_weldid || !m->body_invweight0) return "no contact tables";
  for (int i = 0; i < m->nv; i++) if (m->dof_frictionloss[i] > 0) return "friction loss";
  const int nb = m->nbody;
  // ---- the moving tree: one free-joint root under the world
  int root = -1;
  for (int b = 1; b < nb; b++) {
    if (m->body_dofnum[b] == 0) continue;
    int r = b;
    while (m->body_parentid[r] != 0) r = m->body_parentid[r];
    if (root >= 0 && r != root) return "more than one moving tree";
    root = r;
  }
  if (root < 0 || m->body_jntnum[root] != 1 || m->jnt_type[m->body_jntadr[root]] != MJPCX_JNT_FREE) return "no free-joint root";
  std::vector<char> in_tree(nb, 0), jointed(nb, 0);
  for (int b = 1; b < nb; b++) { int r = b; while (r != 0 && r != root) r = m->body_parentid[r]; in_tree[b] = r == root; }
  for (int b = 1; b < nb; b++) if (in_tree[b]) jointed[b] = m->body_jntnum[b] > 0;
  std::vector<int> jpar(nb, -1), host(nb, -1);  // nearest jointed proper ancestor; the jointed body a body is welded to (itself if jointed)
  for (int b = 1; b < nb; b++) {
    if (!in_tree[b]) continue;
    int h = b;
    while (!jointed[h]) h = m->body_parentid[h];
    host[b] = h;
    if (h != root) { int p = m->body_parentid[h]; while (!jointed[p]) p = m->body_parentid[p]; jpar[h] = p; }
  }
  std::vector<std::vector<int>> jkids(nb);
  for (int b = 1; b < nb; b++) if (in_tree[b] && jointed[b] && b != root) jkids[jpar[b]].push_back(b);
  // limb chains: a jointed body with at most one jointed child, itself a limb-chain body
  std::vector<char> chain(nb, 0);
  for (int b = nb - 1; b >= 1; b--) if (in_tree[b] && jointed[b]) chain[b] = jkids[b].empty() || (jkids[b].size() == 1 && chain[jkids[b][0]]);
  if (chain[root]) return "the tree is a single chain";
  std::vector<int> trunk, limb_root;
  for (int b = root; b >= 0;) {
    trunk.push_back(b);
    int next = -1;
    for (int c : jkids[b]) { if (chain[c]) limb_root.push_back(c); else { if (next >= 0) return "the trunk branches"; next = c; } }
    b = next;
  }
  if ((int)trunk.size() > kTB) return "a trunk chain of more than three jointed bodies";
  if ((int)limb_root.size() != kLimbs) return "not four limbs";
  std::sort(limb_root.begin(), limb_root.end(), [&](int a, int b) { return m->body_dofadr[a] < m->body_dofadr[b]; });
  const int tj_max[kTB] = {1, 2, 1};
  for (size_t i = 1; i < trunk.size(); i++) {
    if (m->body_jntnum[trunk[i]] > tj_max[i]) return "more hinges on a trunk body than staged (2 on the second, 1 on the third)";
    for (int j = 0; j < m->body_jntnum[trunk[i]]; j++) if (m->jnt_type[m->body_jntadr[trunk[i]] + j] != MJPCX_JNT_HINGE) return "a trunk joint other than a hinge";
  }
  // ---- folded bodies: mass, centre of mass, inertia of a jointed body with everything welded to it, in its frame
  auto fold = [&](int B, LBodyT<double>& o) {
    memcpy(o.pos, m->body_pos + 3 * B, 24); memcpy(o.quat, m->body_quat + 4 * B, 32);
    double M = 0, c[3] = {0, 0, 0};
    std::vector<int> parts;
    for (int b = 1; b < nb; b++) if (in_tree[b] && host[b] == B) parts.push_back(b);
    std::vector<double> pc(3 * parts.size()), pI(9 * parts.size());
    for (size_t i = 0; i < parts.size(); i++) {
      const int b = parts[i];
      double p[3], q[4], Rm[9], v[3], qi[4], Ri[9];
      rel_pose(m, b, B, p, q);
      q2m(Rm, q);
      mv(v, Rm, m->body_ipos + 3 * b);
      for (int k = 0; k < 3; k++) pc[3 * i + k] = p[k] + v[k];
      qmul(qi, q, m->body_iquat + 4 * b);
      q2m(Ri, qi);
      for (int r = 0; r < 3; r++) for (int s = 0; s < 3; s++) {
        double a = 0;
        for (int k = 0; k < 3; k++) a += Ri[3 * r + k] * m->body_inertia[3 * b + k] * Ri[3 * s + k];
        pI[9 * i + 3 * r + s] = a;
      }
      M += m->body_mass[b];
      for (int k = 0; k < 3; k++) c[k] += m->body_mass[b] * pc[3 * i + k];
    }
    if (M > 1e-15) for (int k = 0; k < 3; k++) c[k] /= M; else for (int k = 0; k < 3; k++) c[k] = m->body_ipos[3 * B + k];
    double I[9] = {0};
    for (size_t i = 0; i < parts.size(); i++) {
      const double mb = m->body_mass[parts[i]], d[3] = {pc[3 * i] - c[0], pc[3 * i + 1] - c[1], pc[3 * i + 2] - c[2]}, d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
      for (int r = 0; r < 3; r++) for (int s = 0; s < 3; s++) I[3 * r + s] += pI[9 * i + 3 * r + s] + mb * ((r == s ? d2 : 0.0) - d[r] * d[s]);
    }
    o.mass = M; memcpy(o.ipos, c, 24);
    o.inertia[0] = I[0]; o.inertia[1] = I[4]; o.inertia[2] = I[8]; o.inertia[3] = I[1]; o.inertia[4] = I[2]; o.inertia[5] = I[5];
  };
  auto fill_joint = [&](int j, LJointT<double>& o) -> std::string {
    if (m->jnt_type[j] != MJPCX_JNT_HINGE) return "a limb / trunk joint other than a hinge";
    const int d = m->jnt_dofadr[j], qa = m->jnt_qposadr[j];
    o.on = 1; o.dof = d; o.qadr = qa; o.act = -1;
    memcpy(o.pos, m->jnt_pos + 3 * j, 24); memcpy(o.axis, m->jnt_axis + 3 * j, 24);
    o.qpos0 = m->qpos0[qa]; o.qspring = m->qpos_spring[qa]; o.stiffness = m->jnt_stiffness[j]; o.damping = m->dof_damping[d]; o.armature = m->dof_armature[d];
    o.limited = m->jnt_limited[j]; o.range[0] = m->jnt_range[2 * j]; o.range[1] = m->jnt_range[2 * j + 1]; o.margin = m->jnt_margin[j];
    o.invw = m->dof_invweight0[d];
    solref_kb(m, m->jnt_solref + 2 * j, m->jnt_solimp + 5 * j, &o.lim_k, &o.lim_b);
    digest_solimp(o.lim_imp, m->jnt_solimp + 5 * j);
    o.ctrl_lo = -1e30; o.ctrl_hi = 1e30;
    for (int u = 0; u < m->nu; u++) {
      if (m->actuator_trnid[u] != j) continue;
      if (o.act >= 0) return "two actuators on one joint";
      if (m->actuator_gaintype[u] != MJPCX_GAIN_FIXED || m->actuator_biastype[u] != MJPCX_BIAS_NONE || m->actuator_forcelimited[u]) return "an actuator other than a plain motor";
      o.act = u; o.gear_gain = m->actuator_gear[u] * m->actuator_gainprm[3 * u];
      o.ctrl_lo = m->actuator_ctrlrange[2 * u]; o.ctrl_hi = m->actuator_ctrlrange[2 * u + 1];
      if (!m->actuator_ctrllimited[u]) return "an actuator without a control range";
    }
    return "";
  };
  int n_act = 0;
  for (size_t i = 0; i < trunk.size(); i++) fold(trunk[i], L.tbody[i]);
  for (size_t i = 1; i < trunk.size(); i++)
    for (int j = 0; j < m->body_jntnum[trunk[i]]; j++) {
      LJointT<double>& o = L.tjnt[i == 1 ? j : 2];
      const std::string e = fill_joint(m->body_jntadr[trunk[i]] + j, o);
      if (!e.empty()) return e;
      if (o.dof != 6 + (i == 1 ? j : 2) && !(i == 2 && m->body_jntnum[trunk[1]] < 2)) return "trunk dofs not in chain order";
      n_act += o.act >= 0;
    }
  // (a trunk with fewer hinges keeps the unused slots off: armature 1, no coupling -- handled by `on` in the step function)
  {
    const int jf = m->body_jntadr[root];
    if (m->jnt_dofadr[jf] != 0 || m->jnt_qposadr[jf] != 0) return "the free joint is not the first joint";
    for (int k = 0; k < 6; k++) { L.tdamp[k] = m->dof_damping[k]; L.tarm[k] = m->dof_armature[k]; }
    for (int u = 0; u < m->nu; u++) if (m->actuator_trnid[u] == jf) return "an actuator on the free joint";
  }
  const int lj_max[kLB] = {3, 1, 2}, lj_off[kLB] = {0, 3, 4};
  std::vector<int> lane_of(nb, -1), slot_of(nb, -1);  // by jointed body: lane 0..3 / 4 (trunk), body slot
  for (size_t i = 0; i < trunk.size(); i++) { lane_of[trunk[i]] = 4; slot_of[trunk[i]] = (int)i; }
  for (int l = 0; l < kLimbs; l++) {
    LimbT<double>& Lb = L.limb[l];
    int b = limb_root[l], s = 0;
    const int at = jpar[b];
    Lb.attach = slot_of[at];
    if (lane_of[at] != 4) return "a limb that hangs on another limb";
    Lb.nanc = Lb.attach == 0 ? 6 : (Lb.attach == 1 ? 8 : 9);
    L.nattach[Lb.attach]++;
    for (; b >= 0; b = jkids[b].empty() ? -1 : jkids[b][0], s++) {
      if (s >= kLB) return "a limb of more than three jointed bodies";
      if (m->body_jntnum[b] > lj_max[s]) return "more hinges on a limb body than the (3, 1, 2) slots hold";
      lane_of[b] = l; slot_of[b] = s;
      fold(b, Lb.body[s]);
      if (s == 0) {  // pose relative to the trunk body it hangs on (through welded bodies, if any)
        double p[3], q[4];
        rel_pose(m, b, at, p, q);
        memcpy(Lb.body[0].pos, p, 24); memcpy(Lb.body[0].quat, q, 32);
      } else {
        double p[3], q[4];
        rel_pose(m, b, jpar[b], p, q);
        memcpy(Lb.body[s].pos, p, 24); memcpy(Lb.body[s].quat, q, 32);
      }
      for (int j = 0; j < m->body_jntnum[b]; j++) {
        const std::string e = fill_joint(m->body_jntadr[b] + j, Lb.jnt[lj_off[s] + j]);
        if (!e.empty()) return e;
        n_act += Lb.jnt[lj_off[s] + j].act >= 0;
      }
    }
    for (int s2 = s; s2 < kLB; s2++) { Lb.body[s2].quat[0] = 1; }
    for (int j = 0; j < kLD; j++) if (!Lb.jnt[j].on) { Lb.jnt[j].armature = 1; Lb.jnt[j].dof = -1; Lb.jnt[j].qadr = -1; Lb.jnt[j].act = -1; Lb.jnt[j].axis[2] = 1; Lb.jnt[j].ctrl_lo = -1e30; Lb.jnt[j].ctrl_hi = 1e30; }
  }
  for (int k = 0; k < 3; k++) if (!L.tjnt[k].on) { L.tjnt[k].armature = 1; L.tjnt[k].dof = -1; L.tjnt[k].qadr = -1; L.tjnt[k].act = -1; L.tjnt[k].axis[2] = 1; L.tjnt[k].ctrl_lo = -1e30; L.tjnt[k].ctrl_hi = 1e30; }
  if (n_act != m->nu) return "an actuator on a joint outside the trunk and the limbs";
  for (size_t i = 1; i < trunk.size(); i++) {  // trunk bodies' poses relative to the previous trunk body
    double p[3], q[4];
    rel_pose(m, trunk[i], trunk[i - 1], p, q);
    memcpy(L.tbody[i].pos, p, 24); memcpy(L.tbody[i].quat, q, 32);
  }
  for (size_t i = trunk.size(); i < (size_t)kTB; i++) L.tbody[i].quat[0] = 1;
  // the trunk's rows: recorded by lane 2, its actuators' splines evaluated there too (the arms' lanes have slots to spare)
  for (int l = 0; l < kLimbs; l++) { for (int k = 0; k < 3; k++) L.limb[l].tact[k] = -1; }
  L.limb[2].owns_trunk_rows = 1;
  for (int k = 0; k < 3; k++) L.limb[2].tact[k] = L.tjnt[k].act;
  // ---- tendons: limits of fixed tendons over two joints of one limb
  for (int t = 0; t < m->ntendon; t++) {
    if (!m->tendon_limited[t]) continue;
    if (m->tendon_num[t] != 2) return "a limited tendon over other than two joints";
    int ln = -1, sl[2];
    for (int w = 0; w < 2; w++) {
      const int j = m->wrap_objid[m->tendon_adr[t] + w], b = m->jnt_bodyid[j];
      if (lane_of[b] < 0 || lane_of[b] == 4 || (ln >= 0 && lane_of[b] != ln)) return "a limited tendon that is not inside one limb";
      ln = lane_of[b];
      sl[w] = lj_off[slot_of[b]] + (j - m->body_jntadr[b]);
    }
    LTendonT<double>& T = L.limb[ln].tendon;
    if (T.on) return "two limited tendons in one limb";
    T.on = 1; T.slot[0] = sl[0]; T.slot[1] = sl[1];
    T.coef[0] = m->wrap_prm[m->tendon_adr[t]]; T.coef[1] = m->wrap_prm[m->tendon_adr[t] + 1];
    T.range[0] = m->tendon_range[2 * t]; T.range[1] = m->tendon_range[2 * t + 1]; T.margin = m->tendon_margin[t]; T.invw = m->tendon_invweight0[t];
    solref_kb(m, m->tendon_solref_lim + 2 * t, m->tendon_solimp_lim + 5 * t, &T.k, &T.b);
    digest_solimp(T.imp, m->tendon_solimp_lim + 5 * t);
  }
  // ---- the floor: one static plane
  int plane = -1;
  for (int g = 0; g < m->ngeom; g++) {
    const int b = m->geom_bodyid[g];
    if (in_tree[b] || !(m->geom_contype[g] || m->geom_conaffinity[g])) continue;
    if (m->geom_type[g] != MJPCX_GEOM_PLANE || b != 0 || plane >= 0) return "static collidable geoms other than one plane of the world body";
    plane = g;
  }
  if (plane < 0) return "no floor";
  {
    double Rm[9];
    q2m(Rm, m->geom_quat + 4 * plane);
    for (int k = 0; k < 3; k++) { L.plane_pos[k] = m->geom_pos[3 * plane + k]; L.plane_n[k] = Rm[3 * k + 2]; }
    // the contact frame of every floor contact (oracle make_frame on the normal)
    double* x = L.plane_n; double* y = L.plane_t1; double* z = L.plane_t2;
    if (x[1] < 0.5 && x[1] > -0.5) { y[0] = 0; y[1] = 1; y[2] = 0; } else { y[0] = 0; y[1] = 0; y[2] = 1; }
    const double dt = x[0] * y[0] + x[1] * y[1] + x[2] * y[2];
    for (int k = 0; k < 3; k++) y[k] -= dt * x[k];
    const double nn = sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
    for (int k = 0; k < 3; k++) y[k] /= nn;
    z[0] = x[1] * y[2] - x[2] * y[1]; z[1] = x[2] * y[0] - x[0] * y[2]; z[2] = x[0] * y[1] - x[1] * y[0];
  }
  // contact parameters of a geom pair (mj_contactParam; oracle contact_param)
  int npset = 0;
  auto params = [&](int g1, int g2, int& dim, double& mu, double& diag, int& ps) -> std::string {
    LPSetT<double> s;
    memset(&s, 0, sizeof s);
    const double margin = std::max(m->geom_margin[g1], m->geom_margin[g2]), gap = std::max(m->geom_gap[g1], m->geom_gap[g2]);
    s.margin = margin; s.includemargin = margin - gap;
    double fr, solref[2], solimp[5];
    const int p1 = m->geom_priority[g1], p2 = m->geom_priority[g2];
    if (p1 != p2) {
      const int g = p1 > p2 ? g1 : g2;
      dim = m->geom_condim[g]; fr = m->geom_friction[3 * g];
      memcpy(solref, m->geom_solref + 2 * g, sizeof solref); memcpy(solimp, m->geom_solimp + 5 * g, sizeof solimp);
    } else {
      dim = std::max(m->geom_condim[g1], m->geom_condim[g2]); fr = std::max(m->geom_friction[3 * g1], m->geom_friction[3 * g2]);
      const double s1 = m->geom_solmix[g1], s2 = m->geom_solmix[g2];
      const double mix = (s1 >= 1e-15 && s2 >= 1e-15) ? s1 / (s1 + s2) : (s1 < 1e-15 && s2 < 1e-15 ? 0.5 : (s1 < 1e-15 ? 0.0 : 1.0));
      for (int k = 0; k < 2; k++) solref[k] = mix * m->geom_solref[2 * g1 + k] + (1 - mix) * m->geom_solref[2 * g2 + k];
      for (int k = 0; k < 5; k++) solimp[k] = mix * m->geom_solimp[5 * g1 + k] + (1 - mix) * m->geom_solimp[5 * g2 + k];
    }
    mu = std::max(fr, 1e-5);
    solref_kb(m, solref, solimp, &s.k, &s.b);
    digest_solimp(s.imp, solimp);
    diag = m->body_invweight0[2 * m->geom_bodyid[g1]] + m->body_invweight0[2 * m->geom_bodyid[g2]];
    ps = -1;
    for (int i = 0; i < npset; i++) if (memcmp(&L.pset[i], &s, sizeof s) == 0) ps = i;
    if (ps < 0) { if (npset == kMaxPSet) return "more distinct contact-parameter sets than staged"; L.pset[npset] = s; ps = npset++; }
    return "";
  };
  // ---- moving geoms: spheres and capsules of the tree's bodies, in the frame of the jointed body they ride on
  std::vector<int> gslot(m->ngeom, -1);
  int ng = 0, ntg = 0;
  int lgn[kLimbs] = {0, 0, 0, 0};
  for (int g = 0; g < m->ngeom; g++) {
    const int b = m->geom_bodyid[g];
    if (!in_tree[b] || !(m->geom_contype[g] || m->geom_conaffinity[g])) continue;
    if (m->geom_type[g] != MJPCX_GEOM_SPHERE && m->geom_type[g] != MJPCX_GEOM_CAPSULE) return "a collidable geom of the robot other than a sphere or a capsule";
    if (ng == kNG) return "more moving geoms than the shared pose table holds";
    const int h = host[b], ln = lane_of[h];
    LGeomT<double> o;
    memset(&o, 0, sizeof o);
    double p[3], q[4], Rm[9], v[3], qg[4], Rg[9];
    rel_pose(m, b, h, p, q);
    q2m(Rm, q);
    mv(v, Rm, m->geom_pos + 3 * g);
    qmul(qg, q, m->geom_quat + 4 * g);
    q2m(Rg, qg);
    for (int k = 0; k < 3; k++) { o.pos[k] = p[k] + v[k]; o.axis[k] = Rg[3 * k + 2]; }
    o.radius = m->geom_size[3 * g]; o.half = m->geom_type[g] == MJPCX_GEOM_CAPSULE ? m->geom_size[3 * g + 1] : 0.0;
    o.on = 1; o.body = slot_of[h]; o.gslot = ng;
    if ((m->geom_contype[plane] & m->geom_conaffinity[g]) || (m->geom_contype[g] & m->geom_conaffinity[plane])) {
      const std::string e = params(plane, g, o.pdim, o.pmu, o.pdiag, o.pset);
      if (!e.empty()) return e;
      if (o.pdim != 1 && !(o.pdim == 3 && m->cone == 0)) return "floor contacts other than frictionless or pyramidal condim 3";
      if (!(L.pset[o.pset].margin < 0.009)) return "a contact margin of 9 mm or more";
    }
    L.grad[ng] = o.radius; L.ghalf[ng] = o.half; L.glane[ng] = (unsigned char)ln; L.gbody[ng] = (unsigned char)slot_of[h];
    gslot[g] = ng++;
    if (ln == 4) {  // a trunk geom: dealt round robin over the lanes, the arms' first
      const int order[kLimbs] = {2, 3, 0, 1};
      const int tl = order[ntg % kLimbs], ti = ntg / kLimbs;
      if (ti >= kTGL) return "more trunk geoms than dealt slots";
      L.limb[tl].tgeom[ti] = o;
      ntg++;
    } else {
      if (lgn[ln] == kLG) return "more geoms on a limb than staged";
      L.limb[ln].geom[lgn[ln]++] = o;
    }
  }
  L.ngeom = ng;
  // ---- moving-geom pairs behind MuJoCo's filters (pair_cull.h), dealt over the lanes
  {
    std::vector<char> moving(nb, 0);
    for (int b = 1; b < nb; b++) moving[b] = moving[m->body_parentid[b]] || m->body_dofnum[b] > 0;
    std::vector<MovingPair> mp;
    moving_pairs(m, moving, false, mp);
    if ((int)mp.size() > kMaxPair) return "more moving-geom pairs than staged";
    std::vector<LPairT<double>> all;
    for (const MovingPair& q : mp) {
      if (q.kind != kPairThin) return "a moving-geom pair with a box or a cylinder";
      LPairT<double> P;
      memset(&P, 0, sizeof P);
      int dim, ps; double mu;
      const std::string e = params(q.g1, q.g2, dim, mu, P.diag, ps);
      if (!e.empty()) return e;
      if (dim != 1) return "a contact between two moving geoms that is not frictionless (condim 1)";
      auto rb = [&](int g) { return m->geom_type[g] == MJPCX_GEOM_CAPSULE ? m->geom_size[3 * g] + m->geom_size[3 * g + 1] : m->geom_size[3 * g]; };
      P.reach = rb(q.g1) + rb(q.g2) + L.pset[ps].margin;
      P.ga = (unsigned char)gslot[q.g1]; P.gb = (unsigned char)gslot[q.g2]; P.pset = (unsigned char)ps;
      all.push_back(P);
    }
    // dealt evenly, in the model's order within a lane (the order contacts are created in)
    const int n = (int)all.size();
    L.npair = n;
    for (int l = 0; l < kLimbs; l++) {
      const int lo = (int)((long long)n * l / kLimbs), hi = (int)((long long)n * (l + 1) / kLimbs);
      L.limb[l].pair0 = lo; L.limb[l].npair = hi - lo;
      if (hi - lo > kPairsPerLane) return "more pairs per lane than staged";
    }
    for (int i = 0; i < n; i++) L.pair[i] = all[i];
  }
  // ---- the tracking residual: 16 markers (site, mocap body), dealt to the lane that moves the site
  if (task->num_residual_int < 34) return "residual_int too short for the tracking residual";
  {
    const int* ri = task->residual_int;
    L.key_start = ri[0]; L.key_last = ri[1];
    int ls[kLimbs] = {0, 0, 0, 0}, tsn = 0;
    for (int bi = 0; bi < 16; bi++) {
      const int s = ri[2 + bi], b = m->site_bodyid[s];
      if (!in_tree[b]) return "a tracking site outside the robot";
      const int h = host[b];
      LSiteT<double> o;
      memset(&o, 0, sizeof o);
      double p[3], q[4], Rm[9], v[3];
      rel_pose(m, b, h, p, q);
      q2m(Rm, q);
      mv(v, Rm, m->site_pos + 3 * s);
      for (int k = 0; k < 3; k++) o.pos[k] = p[k] + v[k];
      o.on = 1; o.marker = bi; o.mocap = ri[18 + bi];
      int ln = lane_of[h];
      if (ln == 4) { o.body = 3 + slot_of[h]; ln = 2 + (tsn++ & 1); } else o.body = slot_of[h];
      if (ls[ln] == kLS) return "more tracking sites on a lane than staged";
      L.limb[ln].site[ls[ln]++] = o;
    }
  }
  // ---- cost terms: every norm must be a plain sum over its entries (then a lane can add up its own entries' share)
  if (task->num_term > kMaxTerm || task->num_residual > kMaxResid) return "more cost terms / residual entries than staged";
  L.nterm = task->num_term; L.nr = task->num_residual;
  if (L.nr != (m->nv - 6) + m->nu + 3 + 48 + 48) return "residual size is not the tracking residual's";
  for (int t = 0, off = 0; t < task->num_term; t++) {
    const int nt = task->norm[t];
    if (!(nt == 0 || nt == 3 || nt == 5 || nt == 6 || nt == 7 || nt == 8)) return "a cost term whose norm is not a sum over its entries (L2, L22)";
    L.term_norm[t] = nt;
    for (int i = 0; i < task->dim_norm_residual[t]; i++) L.term_of[off + i] = (unsigned char)t;
    off += task->dim_norm_residual[t];
  }
  {  // the residual's layout fixes the term of every entry group: one term each for the joint velocities, the controls and the average, one
     // position and one velocity term per marker (three entries each)
    const int nj = m->nv - 6, c0 = nj + m->nu;
    auto same = [&](int lo, int n) { for (int i = 1; i < n; i++) if (L.term_of[lo + i] != L.term_of[lo]) return false; return true; };
    if (!same(0, nj) || !same(nj, m->nu) || !same(c0, 3)) return "the joint-velocity, control or average entries span several cost terms";
    L.t_qvel = L.term_of[0]; L.t_ctrl = L.term_of[nj]; L.t_avg = L.term_of[c0];
    for (int l = 0; l < kLimbs; l++)
      for (int i = 0; i < kLS; i++) {
        LSiteT<double>& St = L.limb[l].site[i];
        if (!St.on) continue;
        if (!same(c0 + 3 + 3 * St.marker, 3) || !same(c0 + 51 + 3 * St.marker, 3)) return "a marker's three entries span several cost terms";
        St.tpos = L.term_of[c0 + 3 + 3 * St.marker]; St.tvel = L.term_of[c0 + 51 + 3 * St.marker];
      }
  }
  if (task->num_trace > kMaxTrace) return "more traces than staged";
  L.ntrace = task->num_trace;
  for (int q = 0; q < task->num_trace; q++) {
    const int id = task->trace_site[q];
    const int b = id >= 0 ? m->site_bodyid[id] : -1 - id;
    if (!in_tree[b]) return "a trace outside the robot";
    const int h = host[b];
    double p[3], qq[4], Rm[9], v[3] = {0, 0, 0};
    rel_pose(m, b, h, p, qq);
    if (id >= 0) { q2m(Rm, qq); mv(v, Rm, m->site_pos + 3 * id); }
    for (int k = 0; k < 3; k++) L.trace[q].pos[k] = p[k] + v[k];
    L.trace[q].lane = lane_of[h]; L.trace[q].body = slot_of[h];
  }
  for (int k = 0; k < 3; k++) L.gravity[k] = m->gravity[k];
  L.timestep = m->timestep; L.tolerance = m->solver_tolerance; L.meaninertia = m->meaninertia; L.iterations = m->solver_iterations;
  L.nv = m->nv; L.nq = m->nq; L.nu = m->nu; L.nmocap = m->nmocap;
  return "";
}

// the image in the kernel's precision
template <typename R>
inline void limb_cast(const LimbModelD& s, LimbModelT<R>& d) {
  static_assert(sizeof(LimbModelT<float>) % 4 == 0, "staged in 4-byte words");
  memset(&d, 0, sizeof d);
  auto cb = [](const LBodyT<double>& a, LBodyT<R>& b) { for (int k = 0; k < 3; k++) { b.pos[k] = (R)a.pos[k]; b.ipos[k] = (R)a.ipos[k]; } for (int k = 0; k < 4; k++) b.quat[k] = (R)a.quat[k]; for (int k = 0; k < 6; k++) b.inertia[k] = (R)a.inertia[k]; b.mass = (R)a.mass; };
  auto cj = [](const LJointT<double>& a, LJointT<R>& b) {
    for (int k = 0; k < 3; k++) { b.pos[k] = (R)a.pos[k]; b.axis[k] = (R)a.axis[k]; }
    b.qpos0 = (R)a.qpos0; b.qspring = (R)a.qspring; b.stiffness = (R)a.stiffness; b.damping = (R)a.damping; b.armature = (R)a.armature;
    b.range[0] = (R)a.range[0]; b.range[1] = (R)a.range[1]; b.margin = (R)a.margin; b.invw = (R)a.invw; b.lim_k = (R)a.lim_k; b.lim_b = (R)a.lim_b;
    for (int k = 0; k < 5; k++) b.lim_imp[k] = (R)a.lim_imp[k];
    b.gear_gain = (R)a.gear_gain; b.ctrl_lo = (R)a.ctrl_lo; b.ctrl_hi = (R)a.ctrl_hi;
    b.on = a.on; b.limited = a.limited; b.dof = a.dof; b.qadr = a.qadr; b.act = a.act;
  };
  auto cg = [](const LGeomT<double>& a, LGeomT<R>& b) {
    for (int k = 0; k < 3; k++) { b.pos[k] = (R)a.pos[k]; b.axis[k] = (R)a.axis[k]; }
    b.radius = (R)a.radius; b.half = (R)a.half; b.pdiag = (R)a.pdiag; b.pmu = (R)a.pmu; b.on = a.on; b.body = a.body; b.gslot = a.gslot; b.pset = a.pset; b.pdim = a.pdim;
  };
  for (int l = 0; l < kLimbs; l++) {
    const LimbT<double>& a = s.limb[l]; LimbT<R>& b = d.limb[l];
    for (int i = 0; i < kLB; i++) cb(a.body[i], b.body[i]);
    for (int i = 0; i < kLD; i++) cj(a.jnt[i], b.jnt[i]);
    for (int i = 0; i < kLG; i++) cg(a.geom[i], b.geom[i]);
    for (int i = 0; i < kTGL; i++) cg(a.tgeom[i], b.tgeom[i]);
    for (int i = 0; i < kLS; i++) { for (int k = 0; k < 3; k++) b.site[i].pos[k] = (R)a.site[i].pos[k]; b.site[i].on = a.site[i].on; b.site[i].body = a.site[i].body; b.site[i].marker = a.site[i].marker; b.site[i].mocap = a.site[i].mocap; b.site[i].tpos = a.site[i].tpos; b.site[i].tvel = a.site[i].tvel; }
    const LTendonT<double>& t = a.tendon; LTendonT<R>& u = b.tendon;
    u.coef[0] = (R)t.coef[0]; u.coef[1] = (R)t.coef[1]; u.range[0] = (R)t.range[0]; u.range[1] = (R)t.range[1]; u.margin = (R)t.margin; u.invw = (R)t.invw; u.k = (R)t.k; u.b = (R)t.b;
    for (int k = 0; k < 5; k++) u.imp[k] = (R)t.imp[k];
    u.on = t.on; u.slot[0] = t.slot[0]; u.slot[1] = t.slot[1];
    b.attach = a.attach; b.nanc = a.nanc; b.npair = a.npair; b.pair0 = a.pair0; b.owns_trunk_rows = a.owns_trunk_rows;
    for (int k = 0; k < 3; k++) b.tact[k] = a.tact[k];
  }
  for (int i = 0; i < kTB; i++) cb(s.tbody[i], d.tbody[i]);
  for (int i = 0; i < 3; i++) cj(s.tjnt[i], d.tjnt[i]);
  for (int k = 0; k < 6; k++) { d.tdamp[k] = (R)s.tdamp[k]; d.tarm[k] = (R)s.tarm[k]; }
  for (int k = 0; k < 3; k++) { d.plane_pos[k] = (R)s.plane_pos[k]; d.plane_n[k] = (R)s.plane_n[k]; d.plane_t1[k] = (R)s.plane_t1[k]; d.plane_t2[k] = (R)s.plane_t2[k]; d.gravity[k] = (R)s.gravity[k]; }
  d.timestep = (R)s.timestep; d.tolerance = (R)s.tolerance; d.meaninertia = (R)s.meaninertia;
  for (int g = 0; g < kNG; g++) { d.grad[g] = (R)s.grad[g]; d.ghalf[g] = (R)s.ghalf[g]; d.glane[g] = s.glane[g]; d.gbody[g] = s.gbody[g]; }
  for (int i = 0; i < kMaxPSet; i++) { d.pset[i].margin = (R)s.pset[i].margin; d.pset[i].includemargin = (R)s.pset[i].includemargin; d.pset[i].k = (R)s.pset[i].k; d.pset[i].b = (R)s.pset[i].b; for (int k = 0; k < 5; k++) d.pset[i].imp[k] = (R)s.pset[i].imp[k]; }
  for (int i = 0; i < kMaxPair; i++) { d.pair[i].diag = (R)s.pair[i].diag; d.pair[i].reach = (R)s.pair[i].reach; d.pair[i].ga = s.pair[i].ga; d.pair[i].gb = s.pair[i].gb; d.pair[i].pset = s.pair[i].pset; }
  for (int q = 0; q < kMaxTrace; q++) { for (int k = 0; k < 3; k++) d.trace[q].pos[k] = (R)s.trace[q].pos[k]; d.trace[q].lane = s.trace[q].lane; d.trace[q].body = s.trace[q].body; }
  memcpy(d.term_of, s.term_of, sizeof d.term_of); memcpy(d.term_norm, s.term_norm, sizeof d.term_norm);
  d.iterations = s.iterations; d.nv = s.nv; d.nq = s.nq; d.nu = s.nu; d.nr = s.nr; d.nterm = s.nterm; d.ntrace = s.ntrace; d.nmocap = s.nmocap; d.ngeom = s.ngeom; d.npair = s.npair;
  d.key_start = s.key_start; d.key_last = s.key_last; d.t_qvel = s.t_qvel; d.t_ctrl = s.t_ctrl; d.t_avg = s.t_avg;
  for (int i = 0; i < kTB; i++) d.nattach[i] = s.nattach[i];
}

} }  // namespace mjpcx::limb
