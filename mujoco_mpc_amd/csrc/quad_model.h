// quad_model.h -- model constants of the QUAD kernel family (quad_step.h, quad_kernel.h): four lanes per candidate, one lane
// per LEG of a legged floating-base robot (the Unitree A1 of BASELINE configs[2]: trunk with a free joint + 4 chains of 3 hinge
// links). Sixteen candidates share a wavefront; every per-leg quantity (kinematics of the three links, the leg's 3 x 3 block
// of M and of the Newton Hessian, its contacts, friction-loss and limit rows, its three actuators) lives in the lane's own
// registers, everything of the trunk is replicated in the four lanes, and the only cross-lane traffic is DPP quad
// permutes (sums over the four legs). The inertia matrix of such a tree is an ARROWHEAD: four independent 3 x 3 leg blocks
// coupled only through the 6 x 6 trunk block -- a contact on a leg touches the trunk dofs and that leg's dofs, so the
// Newton Hessian M + J' D J keeps that shape (quad_step.h: qd_arrow_factor).
//
// This header is plain C++ (no HIP): the host builds the struct once per context (mjpcx_create) and the CPU emulator of the
// kernel (tests/quademu, test infrastructure) builds the same one.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include <stdint.h>

#include "../../include/mjpcx.h"
#include "pair_cull.h"
#include "quad_abi.h"

namespace mjpcx {

constexpr int kQLegs = 4, kQLinks = 3;
constexpr int kQLegGeom = 8;     // collidable geoms per leg
constexpr int kQTrunkGeom = 8;   // collidable geoms on the trunk (dealt over the four lanes: lane l tests geoms l, l + 4)
constexpr int kQStatic = 4;      // collidable static geoms (world / mocap bodies)
constexpr int kQPairGeom = 8;    // geoms per leg in moving-geom pairs (sphere | capsule; cylinder: against the sphere | capsule geoms of other legs); kQTrunkPairGeom on the trunk (sphere | capsule)
constexpr int kQTrunkPairGeom = 2;
constexpr int kQMaxFric = 8;     // distinct friction sets (mu, tangential, torsional, rolling) over the contact pairs
constexpr int kQMaxKey = 4, kQMaxTrace = 2, kQMaxTerm = 16, kQMaxRay = 4;
constexpr int kQMaxCon = 24;     // contacts per lane and step; more -> the candidate is handed to the wavefront-per-candidate kernel

// contact parameters of a (static geom, moving geom) pair: mj_contactParam with solref / solimp pre-digested
// (oracle/contact.inc contact_param, solref_kb, impedance's clipping)
struct QuadPair {
  int collide, dim, fid, pad;   // fid: friction set (QuadModel::fric)
  double margin, includemargin, mu, fric1, fric3, fric4;  // mu = friction[0] / sqrt(impratio); tangential, torsional, rolling
  double k, b;                                            // reference acceleration: aref = -b vel - k imp pos
  double imp[5];                                          // dmin dmax width mid power, clipped
  double diag;                                            // mj_diagApprox: translational body_invweight0 of the two bodies
};

// the part of a (static geom, moving geom) pair's record that instantiating a contact reads, de-duplicated over the pairs (the A1's 160
// pairs share a dozen): staged in LDS with the model, so that the collision stage -- which tests every capsule end near the floor against
// the pair's margin -- never waits for global memory (measured: that stage was 11 of 79 M cycles per wavefront and rollout)
struct QuadSPair {
  double margin, includemargin, k, b, imp[5], diag;
  int dim, fid;
};
constexpr int kQMaxSPair = 20;

struct QuadGeom {
  int type, link, model_id, static_mask;  // link: -1 trunk, 0..2 link of the lane's leg; static_mask: bit s = collides with static geom s
  int spair, pgi;                 // byte s: index of the pair (static geom s, this geom) in QuadModel::spair; pgi: the geom's place among its leg's pair geoms (QuadLeg::pg_slot) or -1
  double pos[3], rot[9], size[3]; // pose in the body frame (rotation matrix of geom_quat)
  double bound;                   // radius of its bounding sphere
};

struct QuadStatic {
  int type, mocap, model_id, ray;  // mocap: mocap id of its body or -1 (world); ray: group 0 (Ground() casts against it)
  double pos[3], rot[9], size[3];  // pose in the body frame (world body: world pose)
  double bound;                    // radius of its bounding sphere (box)
};

struct QuadLeg {
  double body_pos[kQLinks][3], body_quat[kQLinks][4], body_ipos[kQLinks][3], body_iquat[kQLinks][4];
  double body_mass[kQLinks], body_inertia[kQLinks][3];
  double jnt_pos[kQLinks][3], jnt_axis[kQLinks][3];
  double qpos0[kQLinks], qpos_spring[kQLinks], stiffness[kQLinks], armature[kQLinks], damping[kQLinks];
  double range[kQLinks][2], margin[kQLinks];
  double self_box[kQLinks][2];  // joint values inside: the leg's own pairs (pg_active[own leg]) are proven apart and need no test (empty box: no proof)
  double guard[kQLinks][2];  // joint values outside [guard[j][0], guard[j][1]] leave the box the bake-time proofs of the dropped pairs cover (pair_cull.h)
  double lim_k[kQLinks], lim_b[kQLinks], lim_imp[kQLinks][5], lim_diag[kQLinks];
  double floss[kQLinks], floss_R[kQLinks], floss_D[kQLinks], floss_b[kQLinks];
  double act_gear[kQLinks], act_gain[kQLinks], act_bias[kQLinks][3], ctrlrange[kQLinks][2], forcerange[kQLinks][2];
  double key_q[kQMaxKey][kQLinks];  // keyframe joint values of this leg (Posture residual)
  int limited[kQLinks], act_biastype[kQLinks], ctrllimited[kQLinks], forcelimited[kQLinks];
  double reach;                           // no point of the leg's geoms is farther than this from the origin of its first link (its geoms' bounds along the chain)
  int ngeom, foot_slot, foot_index, npg;  // foot_slot: the leg's geom the residual reads; foot_index: its place in foot_geom_id_ (FL HL FR HR)
  int pg_slot[kQPairGeom];                // the leg's geoms that can touch another leg or the trunk (self-collision test)
  double pg_reach[kQPairGeom];              // radius of the pair geom's bounding sphere about its centre (capsule: radius + half length)
  double pg_half[kQPairGeom], pg_rad[kQPairGeom];  // the leg-level cull's end spheres: half length along the axis, radius about each end (cylinder: 0, its bounding radius)
  unsigned long long pg_first[kQLegs + 1];  // per other leg (kQLegs: the trunk), bit 8 i + j: the own pair geom i is geom1 of the pair with the other's j
  unsigned long long pg_active[kQLegs + 1]; // bit 8 i + j: (own pair geom i, the other's j) is a pair MuJoCo's filters leave and pair_cull.h did not prove apart
  QuadGeom geom[kQLegGeom];
};

struct QuadModel {
  double timestep, gravity[3], tolerance, meaninertia, total_mass;
  int iterations, gravity_on, nstatic, ntrunk_geom, ntrace, nterm, nr, nray;
  // trunk
  double trunk_ipos[3], trunk_iquat[4], trunk_mass, trunk_inertia[3];
  double trunk_reach;               // no point of the trunk's geoms is farther than this from the trunk's origin
  double head_pos[3];               // the site the Position residual reads (trunk frame)
  double trace_pos[kQMaxTrace][3];  // traced points (trunk frame)
  QuadGeom trunk_geom[kQTrunkGeom];
  QuadStatic stat[kQStatic];
  QuadLeg leg[kQLegs];
  // cost terms
  int term_dim[kQMaxTerm], term_norm[kQMaxTerm], term_off[kQMaxTerm];
  // model ids the host needs when it decodes results
  int trunk_body, first_leg_body, goal_mocap, npair;
  // self-collision: the moving-geom pairs MuJoCo's filters leave are ALL pairs (leg geom, geom of another leg) and (trunk geom, leg geom)
  // over these sets (checked by quad_build); the kernel only tests them -- a pair within pair_margin hands the candidate on
  int ntpg, tpg_slot[kQTrunkPairGeom];
  double tpg_reach[kQTrunkPairGeom];  // bounding-sphere radii of the trunk's pair geoms (as QuadLeg::pg_reach)
  double tpg_box[2][3];               // box of the trunk's pair geoms' bounding spheres in the trunk frame (lower, upper corner)
  double pair_margin;
  // friction sets of the contact pairs: regularised mu, then the tangential / torsional / rolling coefficient, ZERO for rows the pair's
  // condim does not have (the cone formulas then reduce to the lower condim's)
  int nfric, nspair;
  double fric[kQMaxFric][6];   // mu, tangential, torsional, rolling, 1 / mu^2, 1 / (mu^2 (1 + mu^2))
  QuadSPair spair[kQMaxSPair]; // distinct contact-parameter sets of the (static geom, moving geom) pairs
};

// everything the kernel needs that is too rarely read to deserve LDS: the pair table [static][trunk geoms | leg geoms]
struct QuadTables {
  QuadPair trunk[kQStatic][kQTrunkGeom];
  QuadPair leg[kQLegs][kQStatic][kQLegGeom];
  // moving-geom pairs (self-collision): [own leg][own pair geom][other leg, kQLegs = the trunk][other pair geom]; `pad` = 1 if the own geom
  // is geom1 of the pair in MuJoCo's order (lower geom type first, then lower index): the contact normal points from geom1 to geom2
  QuadPair mm[kQLegs][kQPairGeom][kQLegs + 1][kQPairGeom];
  int npair;
};


namespace quad_detail {
inline void quat2mat(double* m, const double* q) {
  const double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  const double q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3], q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2 * (q12 - q03); m[2] = 2 * (q13 + q02); m[3] = 2 * (q12 + q03);
  m[5] = 2 * (q23 - q01); m[6] = 2 * (q13 - q02); m[7] = 2 * (q23 + q01);
}
inline double clip(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
inline void digest_solimp(double* out, const double* solimp) {
  out[0] = clip(solimp[0], 0.0001, 0.9999); out[1] = clip(solimp[1], 0.0001, 0.9999); out[2] = solimp[2];
  out[3] = clip(solimp[3], 0.0001, 0.9999); out[4] = solimp[4] < 1 ? 1 : solimp[4];
}
inline void solref_kb(const mjpcx_model* m, const double* solref, const double* solimp, double* k, double* b) {
  const double dmax = clip(solimp[1], 0.0001, 0.9999);
  if (solref[0] > 0) {
    double tc = solref[0];
    if (!(m->disableflags & MJPCX_DSBL_REFSAFE) && tc < 2 * m->timestep) tc = 2 * m->timestep;
    *k = 1.0 / (dmax * dmax * tc * tc * solref[1] * solref[1]);
    *b = 2.0 / (dmax * tc);
  } else {
    *k = -solref[0] / (dmax * dmax);
    *b = -solref[1] / dmax;
  }
}
// impedance at zero violation (friction-loss rows: pos = margin = 0)
inline double impedance0(const double* solimp) {
  double d[5];
  digest_solimp(d, solimp);
  if (d[0] == d[1] || d[2] <= 1e-15) return 0.5 * (d[0] + d[1]);
  return d[0];
}
}  // namespace quad_detail

// Builds the quad kernel's view of a model + task. Returns "" on success, otherwise why the model is outside the class the
// kernel covers (the caller then keeps the wavefront-per-candidate kernels: not an error).
inline std::string quad_build(const mjpcx_model* m, const mjpcx_task* task, QuadModel* qm, QuadTables* qt) {
  using namespace quad_detail;
  std::memset(qm, 0, sizeof *qm);
  std::memset(qt, 0, sizeof *qt);
  for (int l = 0; l < kQLegs; l++) for (int g = 0; g < kQLegGeom; g++) qm->leg[l].geom[g].pgi = -1;  // (not a pair geom until the pair list says so)
  for (int g = 0; g < kQTrunkGeom; g++) qm->trunk_geom[g].pgi = -1;
  if (m->integrator != MJPCX_INT_EULER) return "integrator is not Euler";
  if (m->disableflags != 0) return "non-default disable flags";
  if (m->ntendon != 0 || m->na != 0) return "tendons / activations";
  if (m->nv != 6 + kQLegs * kQLinks || m->nq != 7 + kQLegs * kQLinks || m->nu != kQLegs * kQLinks) return "not a 6 + 4 x 3 dof model";
  // the trunk: the only body with a free joint, child of the world; the legs: 12 bodies after it, leg-major, one hinge each
  int trunk = -1;
  for (int b = 1; b < m->nbody; b++)
    if (m->body_jntnum[b] == 1 && m->jnt_type[m->body_jntadr[b]] == MJPCX_JNT_FREE) { trunk = b; break; }
  if (trunk < 0 || m->body_parentid[trunk] != 0 || m->body_dofadr[trunk] != 0 || m->jnt_qposadr[m->body_jntadr[trunk]] != 0) return "no floating base at dof 0";
  if (trunk + kQLegs * kQLinks > m->nbody - 1) return "fewer than 12 bodies after the trunk";
  for (int b = 1; b < m->nbody; b++) {
    const bool leg_body = b > trunk && b <= trunk + kQLegs * kQLinks;
    if (b == trunk || leg_body) continue;
    if (m->body_dofnum[b] != 0 || m->body_parentid[b] != 0) return "a moving body outside the trunk + legs";
  }
  for (int l = 0; l < kQLegs; l++)
    for (int j = 0; j < kQLinks; j++) {
      const int b = trunk + 1 + kQLinks * l + j;
      if (m->body_parentid[b] != (j == 0 ? trunk : b - 1)) return "leg bodies are not chains of the trunk";
      if (m->body_jntnum[b] != 1 || m->body_dofnum[b] != 1) return "a leg link without exactly one joint";
      const int jn = m->body_jntadr[b];
      if (m->jnt_type[jn] != MJPCX_JNT_HINGE) return "a leg joint that is not a hinge";
      if (m->jnt_dofadr[jn] != 6 + kQLinks * l + j || m->jnt_qposadr[jn] != 7 + kQLinks * l + j) return "leg dofs are not leg-major";
      if (m->actuator_trnid[kQLinks * l + j] != jn) return "actuators are not one per leg joint, in joint order";
      if (m->body_mocapid[b] >= 0) return "mocap leg body";
    }
  for (int i = 0; i < 6; i++)
    if (m->dof_frictionloss[i] > 0 || m->dof_damping[i] > 0 || m->dof_armature[i] != 0) return "friction loss / damping / armature on the floating base";
  if (m->jnt_stiffness[m->body_jntadr[trunk]] != 0) return "spring on the floating base";
  int max_condim = 1;
  for (int g = 0; g < m->ngeom; g++) if (m->geom_contype[g] || m->geom_conaffinity[g]) max_condim = std::max(max_condim, (int)m->geom_condim[g]);
  if (m->cone != 1 && max_condim > 1) return "pyramidal friction cones";
  if (m->nkey > kQMaxKey) return "more keyframes than the quad kernel stages";

  qm->timestep = m->timestep; qm->tolerance = m->solver_tolerance; qm->meaninertia = m->meaninertia; qm->iterations = m->solver_iterations;
  for (int k = 0; k < 3; k++) qm->gravity[k] = m->gravity[k];
  qm->gravity_on = 1;
  qm->trunk_body = trunk; qm->first_leg_body = trunk + 1;
  qm->total_mass = m->body_subtreemass ? m->body_subtreemass[trunk] : 0;
  if (!(qm->total_mass > 0)) {
    qm->total_mass = 0;
    for (int b = trunk; b <= trunk + kQLegs * kQLinks; b++) qm->total_mass += m->body_mass[b];
  }
  for (int k = 0; k < 3; k++) { qm->trunk_ipos[k] = m->body_ipos[3 * trunk + k]; qm->trunk_inertia[k] = m->body_inertia[3 * trunk + k]; }
  for (int k = 0; k < 4; k++) qm->trunk_iquat[k] = m->body_iquat[4 * trunk + k];
  qm->trunk_mass = m->body_mass[trunk];

  for (int l = 0; l < kQLegs; l++) {
    QuadLeg& L = qm->leg[l];
    for (int j = 0; j < kQLinks; j++) {
      const int b = trunk + 1 + kQLinks * l + j, jn = m->body_jntadr[b], dof = 6 + kQLinks * l + j, qa = 7 + kQLinks * l + j, u = kQLinks * l + j;
      for (int k = 0; k < 3; k++) {
        L.body_pos[j][k] = m->body_pos[3 * b + k]; L.body_ipos[j][k] = m->body_ipos[3 * b + k]; L.body_inertia[j][k] = m->body_inertia[3 * b + k];
        L.jnt_pos[j][k] = m->jnt_pos[3 * jn + k]; L.jnt_axis[j][k] = m->jnt_axis[3 * jn + k];
      }
      for (int k = 0; k < 4; k++) { L.body_quat[j][k] = m->body_quat[4 * b + k]; L.body_iquat[j][k] = m->body_iquat[4 * b + k]; }
      L.body_mass[j] = m->body_mass[b];
      L.qpos0[j] = m->qpos0[qa]; L.qpos_spring[j] = m->qpos_spring[qa]; L.stiffness[j] = m->jnt_stiffness[jn];
      L.armature[j] = m->dof_armature[dof]; L.damping[j] = m->dof_damping[dof];
      L.range[j][0] = m->jnt_range[2 * jn]; L.range[j][1] = m->jnt_range[2 * jn + 1]; L.margin[j] = m->jnt_margin[jn];
      L.limited[j] = m->jnt_limited[jn];
      if (L.limited[j] && !(L.range[j][1] - L.range[j][0] > 2 * L.margin[j])) return "a joint whose two limits can be active at once";
      solref_kb(m, m->jnt_solref + 2 * jn, m->jnt_solimp + 5 * jn, &L.lim_k[j], &L.lim_b[j]);
      digest_solimp(L.lim_imp[j], m->jnt_solimp + 5 * jn);
      L.lim_diag[j] = m->dof_invweight0[dof];
      L.floss[j] = m->dof_frictionloss[dof];
      if (L.floss[j] > 0) {
        const double imp = impedance0(m->dof_solimp + 5 * dof);
        double R = (1 - imp) / imp * m->dof_invweight0[dof];
        if (R < 1e-15) R = 1e-15;
        double k_;
        solref_kb(m, m->dof_solref + 2 * dof, m->dof_solimp + 5 * dof, &k_, &L.floss_b[j]);
        L.floss_R[j] = R; L.floss_D[j] = 1.0 / R;
      }
      L.act_gear[j] = m->actuator_gear[u]; L.act_gain[j] = m->actuator_gainprm[3 * u];
      for (int k = 0; k < 3; k++) L.act_bias[j][k] = m->actuator_biasprm[3 * u + k];
      L.act_biastype[j] = m->actuator_biastype[u]; L.ctrllimited[j] = m->actuator_ctrllimited[u]; L.forcelimited[j] = m->actuator_forcelimited[u];
      for (int k = 0; k < 2; k++) { L.ctrlrange[j][k] = m->actuator_ctrlrange[2 * u + k]; L.forcerange[j][k] = m->actuator_forcerange[2 * u + k]; }
      for (int key = 0; key < m->nkey; key++) L.key_q[key][j] = m->key_qpos[(size_t)key * m->nq + qa];
    }
    L.foot_slot = -1; L.foot_index = -1;
  }

  // ---- geoms: static (world / mocap bodies) and moving (trunk, legs), collidable ones only
  std::vector<int> slot_leg(m->ngeom, -2), slot_idx(m->ngeom, -1);  // moving geoms: leg (-1 trunk), slot
  std::vector<int> static_slot(m->ngeom, -1);
  for (int g = 0; g < m->ngeom; g++) {
    const int b = m->geom_bodyid[g];
    const bool collidable = m->geom_contype[g] || m->geom_conaffinity[g];
    const bool ray = m->geom_group[g] == 0 && (m->geom_type[g] == MJPCX_GEOM_PLANE || m->geom_type[g] == MJPCX_GEOM_SPHERE || m->geom_type[g] == MJPCX_GEOM_BOX);
    double rot[9];
    quat2mat(rot, m->geom_quat + 4 * g);
    if (b < trunk || b > trunk + kQLegs * kQLinks) {  // static
      if (!collidable && !ray) continue;
      if (qm->nstatic == kQStatic) return "more static geoms than the quad kernel stages";
      QuadStatic& s = qm->stat[qm->nstatic];
      s.type = m->geom_type[g]; s.mocap = m->body_mocapid[b]; s.model_id = g; s.ray = ray;
      if (b != 0 && s.mocap < 0) return "static geom on a non-mocap body";
      for (int k = 0; k < 3; k++) { s.pos[k] = m->geom_pos[3 * g + k]; s.size[k] = m->geom_size[3 * g + k]; }
      std::memcpy(s.rot, rot, sizeof rot);
      s.bound = std::sqrt(s.size[0] * s.size[0] + s.size[1] * s.size[1] + s.size[2] * s.size[2]);
      static_slot[g] = collidable ? qm->nstatic : -1;
      if (!collidable) s.type = -1 - s.type;  // ray-only geom: the collision pass skips it
      qm->nstatic++;
      if (ray) qm->nray++;
      continue;
    }
    if (ray) return "a moving geom in group 0 (Ground() would hit it)";
    if (!collidable) continue;
    QuadGeom* dst;
    if (b == trunk) {
      if (qm->ntrunk_geom == kQTrunkGeom) return "more trunk geoms than the quad kernel stages";
      slot_leg[g] = -1; slot_idx[g] = qm->ntrunk_geom;
      dst = &qm->trunk_geom[qm->ntrunk_geom++];
      dst->link = -1;
    } else {
      const int l = (b - trunk - 1) / kQLinks, j = (b - trunk - 1) % kQLinks;
      QuadLeg& L = qm->leg[l];
      if (L.ngeom == kQLegGeom) return "more geoms on a leg than the quad kernel stages";
      slot_leg[g] = l; slot_idx[g] = L.ngeom;
      dst = &L.geom[L.ngeom++];
      dst->link = j;
    }
    dst->type = m->geom_type[g]; dst->model_id = g;
    for (int k = 0; k < 3; k++) { dst->pos[k] = m->geom_pos[3 * g + k]; dst->size[k] = m->geom_size[3 * g + k]; }
    std::memcpy(dst->rot, rot, sizeof rot);
    const double* sz = dst->size;
    dst->bound = dst->type == MJPCX_GEOM_CAPSULE ? sz[0] + sz[1] : dst->type == MJPCX_GEOM_CYLINDER ? std::sqrt(sz[0] * sz[0] + sz[1] * sz[1])
               : dst->type == MJPCX_GEOM_BOX ? std::sqrt(sz[0] * sz[0] + sz[1] * sz[1] + sz[2] * sz[2]) : sz[0];
    {  // how far from the origin of the leg's first link (of the trunk) a point of this geom can be: the links' offsets down to its body + its own
      double far = std::sqrt(dst->pos[0] * dst->pos[0] + dst->pos[1] * dst->pos[1] + dst->pos[2] * dst->pos[2]) + dst->bound;
      if (b == trunk) qm->trunk_reach = std::max(qm->trunk_reach, far);
      else {
        QuadLeg& L = qm->leg[(b - trunk - 1) / kQLinks];
        for (int j = 1; j <= dst->link; j++)  // (a hinge off its body's origin moves that origin by up to twice the offset)
          far += std::sqrt(L.body_pos[j][0] * L.body_pos[j][0] + L.body_pos[j][1] * L.body_pos[j][1] + L.body_pos[j][2] * L.body_pos[j][2]) +
                 2 * std::sqrt(L.jnt_pos[j][0] * L.jnt_pos[j][0] + L.jnt_pos[j][1] * L.jnt_pos[j][1] + L.jnt_pos[j][2] * L.jnt_pos[j][2]);
        L.reach = std::max(L.reach, far);
      }
    }
  }
  // pair parameters (oracle/contact.inc contact_param; o_collision's type table)
  const double impratio = m->impratio > 1e-15 ? m->impratio : 1.0;
  // contact parameters of a geom pair (oracle/contact.inc contact_param + the solref / solimp digestion), "" or an error
  auto pair_params = [&](int g1, int g2, QuadPair& p) -> std::string {
      const double margin = std::max(m->geom_margin[g1], m->geom_margin[g2]), gap = std::max(m->geom_gap[g1], m->geom_gap[g2]);
      p.margin = margin; p.includemargin = margin - gap;
      double fr[3], solref[2], solimp[5];
      const int p1 = m->geom_priority[g1], p2 = m->geom_priority[g2];
      if (p1 != p2) {
        const int g = p1 > p2 ? g1 : g2;
        p.dim = m->geom_condim[g];
        for (int k = 0; k < 3; k++) fr[k] = m->geom_friction[3 * g + k];
        std::memcpy(solref, m->geom_solref + 2 * g, sizeof solref);
        std::memcpy(solimp, m->geom_solimp + 5 * g, sizeof solimp);
      } else {
        p.dim = std::max(m->geom_condim[g1], m->geom_condim[g2]);
        for (int k = 0; k < 3; k++) fr[k] = std::max(m->geom_friction[3 * g1 + k], m->geom_friction[3 * g2 + k]);
        const double s1 = m->geom_solmix[g1], s2 = m->geom_solmix[g2];
        const double mix = (s1 >= 1e-15 && s2 >= 1e-15) ? s1 / (s1 + s2) : (s1 < 1e-15 && s2 < 1e-15 ? 0.5 : (s1 < 1e-15 ? 0.0 : 1.0));
        for (int k = 0; k < 2; k++) solref[k] = mix * m->geom_solref[2 * g1 + k] + (1 - mix) * m->geom_solref[2 * g2 + k];
        for (int k = 0; k < 5; k++) solimp[k] = mix * m->geom_solimp[5 * g1 + k] + (1 - mix) * m->geom_solimp[5 * g2 + k];
      }
      if (p.dim != 1 && p.dim != 3 && p.dim != 4 && p.dim != 6) return std::string("unsupported condim");
      p.fric1 = std::max(fr[0], 1e-5); p.fric3 = std::max(fr[1], 1e-5); p.fric4 = std::max(fr[2], 1e-5);
      p.mu = p.fric1 / std::sqrt(impratio);
      solref_kb(m, solref, solimp, &p.k, &p.b);
      digest_solimp(p.imp, solimp);
      p.diag = m->body_invweight0[2 * m->geom_bodyid[g1]] + m->body_invweight0[2 * m->geom_bodyid[g2]];
      if (!(p.margin < 0.009)) return std::string("a contact margin of 9 mm or more (the collision pass rejects geoms 1 cm clear of a static geom)");
      const double fs[6] = {p.mu, p.dim >= 3 ? p.fric1 : 0.0, p.dim >= 4 ? p.fric3 : 0.0, p.dim >= 6 ? p.fric4 : 0.0, 1.0 / (p.mu * p.mu),
                            1.0 / (p.mu * p.mu * (1 + p.mu * p.mu))};
      p.fid = -1;
      for (int f = 0; f < qm->nfric; f++) if (std::memcmp(qm->fric[f], fs, sizeof fs) == 0) p.fid = f;
      if (p.fid < 0) {
        if (qm->nfric == kQMaxFric) return std::string("more distinct friction sets than the quad kernel stages");
        std::memcpy(qm->fric[qm->nfric], fs, sizeof fs);
        p.fid = qm->nfric++;
      }
      return std::string();
  };
  for (int s = 0; s < qm->nstatic; s++) {
    const int g1 = qm->stat[s].model_id;
    if (static_slot[g1] < 0) continue;
    const int t1 = m->geom_type[g1];
    for (int g2 = 0; g2 < m->ngeom; g2++) {
      if (slot_leg[g2] == -2) continue;
      QuadPair& p = slot_leg[g2] < 0 ? qt->trunk[s][slot_idx[g2]] : qt->leg[slot_leg[g2]][s][slot_idx[g2]];
      const int t2 = m->geom_type[g2];
      bool ok = (m->geom_contype[g1] & m->geom_conaffinity[g2]) || (m->geom_contype[g2] & m->geom_conaffinity[g1]);
      if (t1 == MJPCX_GEOM_PLANE) ok &= t2 == MJPCX_GEOM_SPHERE || t2 == MJPCX_GEOM_CAPSULE || t2 == MJPCX_GEOM_BOX || t2 == MJPCX_GEOM_CYLINDER;
      else if (t1 == MJPCX_GEOM_SPHERE || t1 == MJPCX_GEOM_BOX) ok &= t2 == MJPCX_GEOM_SPHERE;
      else ok = false;
      p.collide = ok;
      if (!ok) continue;
      QuadGeom& qg = slot_leg[g2] < 0 ? qm->trunk_geom[slot_idx[g2]] : qm->leg[slot_leg[g2]].geom[slot_idx[g2]];
      qg.static_mask |= 1 << s;
      { const std::string err = pair_params(g1, g2, p); if (!err.empty()) return err; }
      QuadSPair sp;
      std::memset(&sp, 0, sizeof sp);
      sp.margin = p.margin; sp.includemargin = p.includemargin; sp.k = p.k; sp.b = p.b; sp.diag = p.diag; sp.dim = p.dim; sp.fid = p.fid;
      std::memcpy(sp.imp, p.imp, sizeof sp.imp);
      int si = -1;
      for (int f = 0; f < qm->nspair; f++) if (std::memcmp(&qm->spair[f], &sp, sizeof sp) == 0) si = f;
      if (si < 0) {
        if (qm->nspair == kQMaxSPair) return std::string("more distinct contact-parameter sets over the (static geom, moving geom) pairs than the quad kernel stages");
        qm->spair[qm->nspair] = sp;
        si = qm->nspair++;
      }
      qg.spair |= si << (8 * s);
    }
  }
  // moving-geom pairs (pair_cull.h: MuJoCo's filters, each pair's class, the proofs). The kernel walks (own pair geom, pair geom of another
  // leg | of the trunk) by the bits of pg_active: sphere | capsule pairs and (sphere | capsule, cylinder) pairs between two legs.
  {
    std::vector<char> in_pair(m->ngeom, 0), moving(m->nbody, 0);
    for (int b = 1; b < m->nbody; b++) moving[b] = moving[m->body_parentid[b]] || m->body_dofnum[b] > 0;
    double pmargin = 0;
    std::vector<MovingPair> mp, plist;
    if (m->body_weldid) moving_pairs(m, moving, true, mp);
    bool proofs = false, self_unproven = false;
    std::vector<double> pad_lo(m->njnt, kPairCullPad), pad_hi(m->njnt, kPairCullPad);
    std::vector<double> self_lo(m->njnt, kPairCullPad), self_hi(m->njnt, kPairCullPad);  // pads of the proofs of the WALKED pairs inside one leg
    for (const MovingPair& q : mp) {
      if (q.kind == kPairOther) continue;                                 // (no narrow phase anywhere: left out and reported by WaveHost::build)
      // two solids have no narrow phase: the wavefront-per-candidate kernels and the oracle WATCH every such pair (warning bit 128 should a
      // thin geom containing one come within the margin of the other) -- this layout cannot, so only a plain proof lets it drop one; with a
      // tight-pad proof or none the model is declined (pair_cull.h's contract) and keeps the kernels that watch
      if (q.kind == kPairSolids && !(q.apart && q.tight_jnt < 0))
        return "two solids (box | cylinder) of moving bodies that are not proven apart with the wide pad: only the wavefront-per-candidate kernels watch such a pair";
      // what the layout can walk: sphere | capsule pairs between two legs or a leg and the trunk; a leg's sphere | capsule against a cylinder of
      // another leg, or of its own leg on a link above it
      const char* why = nullptr;
      if (slot_leg[q.g1] == -2 || slot_leg[q.g2] == -2) why = "a moving-geom pair with a geom outside the trunk and the legs";
      else if (slot_leg[q.g1] == slot_leg[q.g2] && (q.kind != kPairThinSolid || m->geom_bodyid[q.g1] <= m->geom_bodyid[q.g2]))
        why = "a self-collision pair inside one leg other than (sphere | capsule on a lower link, cylinder on a higher one)";
      else if (q.kind == kPairThinSolid && (m->geom_type[q.g2] != MJPCX_GEOM_CYLINDER || slot_leg[q.g1] < 0 || slot_leg[q.g2] < 0))
        why = "a (sphere | capsule, box | cylinder) pair other than between legs' geoms with a cylinder";
      else if (q.kind == kPairSolids) why = "two solids";
      // dropped on the strength of its proof (the step function checks the joint ranges the proofs cover: kFlagRange) -- unless the proof
      // needed the tight pad and the pair can be walked instead
      if (q.apart && (q.tight_jnt < 0 || why)) {
        proofs = true;
        if (q.tight_jnt >= 0) (q.tight_side == 0 ? pad_lo : pad_hi)[q.tight_jnt] = kPairCullPadTight;
        continue;
      }
      if (why) return why;
      if (slot_leg[q.g1] == slot_leg[q.g2]) {  // walked, yet proven apart inside some joint box: there the step function skips the test
        if (!q.apart) self_unproven = true;
        else if (q.tight_jnt >= 0) (q.tight_side == 0 ? self_lo : self_hi)[q.tight_jnt] = kPairCullPadTight;
      }
      in_pair[q.g1] = in_pair[q.g2] = 1;
      plist.push_back(q);
      pmargin = std::max(pmargin, std::max(m->geom_margin[q.g1], m->geom_margin[q.g2]));
    }
    for (int g = 0; g < m->ngeom; g++) {
      if (!in_pair[g]) continue;
      if (slot_leg[g] < 0) { if (qm->ntpg == kQTrunkPairGeom) return "more trunk geoms in self-collision pairs than staged"; qm->tpg_slot[qm->ntpg++] = slot_idx[g]; }
      else { QuadLeg& L = qm->leg[slot_leg[g]]; if (L.npg == kQPairGeom) return "more leg geoms in self-collision pairs than staged"; L.geom[slot_idx[g]].pgi = L.npg; L.pg_slot[L.npg++] = slot_idx[g]; }
    }
    qt->npair = (int)plist.size();
    qm->pair_margin = pmargin;
    // the joint box the proofs of the dropped pairs cover (qpos units); without proofs, or for a joint without a range, everything
    for (int l = 0; l < kQLegs; l++)
      for (int j = 0; j < kQLinks; j++) {
        const int jid = m->body_jntadr[trunk + 1 + kQLinks * l + j];
        const bool on = proofs && m->jnt_limited[jid];
        qm->leg[l].guard[j][0] = on ? m->jnt_range[2 * jid] - pad_lo[jid] : -1e30;
        qm->leg[l].guard[j][1] = on ? m->jnt_range[2 * jid + 1] + pad_hi[jid] : 1e30;
        const bool box = !self_unproven && m->jnt_limited[jid];
        qm->leg[l].self_box[j][0] = box ? m->jnt_range[2 * jid] - self_lo[jid] : 1e30;
        qm->leg[l].self_box[j][1] = box ? m->jnt_range[2 * jid + 1] + self_hi[jid] : -1e30;
      }
    auto pg_index = [&](int g) { const int l = slot_leg[g]; const int* sl = l < 0 ? qm->tpg_slot : qm->leg[l].pg_slot; const int n = l < 0 ? qm->ntpg : qm->leg[l].npg;
                                 for (int i = 0; i < n; i++) if (sl[i] == slot_idx[g]) return i; return -1; };
    for (const MovingPair& pr : plist) {
      const int g1 = pr.g1, g2 = pr.g2;  // MuJoCo's order: lower geom type first, then lower index
      for (int side = 0; side < 2; side++) {
        const int own = side == 0 ? g1 : g2, other = side == 0 ? g2 : g1;
        if (slot_leg[own] < 0) continue;  // (the trunk has no lane of its own: the leg's lane handles a trunk-leg pair)
        if (side == 1 && slot_leg[own] == slot_leg[other]) continue;  // (a pair inside one leg is walked once, from its geom1)
        const int o = slot_leg[other] < 0 ? kQLegs : slot_leg[other];
        QuadPair& p = qt->mm[slot_leg[own]][pg_index(own)][o][pg_index(other)];
        const std::string err = pair_params(g1, g2, p);
        if (!err.empty()) return err;
        p.collide = 1; p.pad = own == g1;
        const unsigned long long bit = 1ull << (8 * pg_index(own) + pg_index(other));
        qm->leg[slot_leg[own]].pg_active[o] |= bit;
        if (own == g1) qm->leg[slot_leg[own]].pg_first[o] |= bit;
      }
    }
  }
  {
    auto reach = [](const QuadGeom& g) { return g.type == MJPCX_GEOM_CAPSULE ? g.size[0] + g.size[1] : (g.type == MJPCX_GEOM_CYLINDER ? std::sqrt(g.size[0] * g.size[0] + g.size[1] * g.size[1]) : g.size[0]); };
    for (int l = 0; l < kQLegs; l++) for (int i = 0; i < qm->leg[l].npg; i++) {
      const QuadGeom& g = qm->leg[l].geom[qm->leg[l].pg_slot[i]];
      qm->leg[l].pg_reach[i] = reach(g);
      qm->leg[l].pg_half[i] = g.type == MJPCX_GEOM_CAPSULE ? g.size[1] : 0.0;
      qm->leg[l].pg_rad[i] = g.type == MJPCX_GEOM_CAPSULE ? g.size[0] : reach(g);
    }
    for (int j = 0; j < qm->ntpg; j++) qm->tpg_reach[j] = reach(qm->trunk_geom[qm->tpg_slot[j]]);
    for (int k = 0; k < 3; k++) { qm->tpg_box[0][k] = 1e30; qm->tpg_box[1][k] = -1e30; }
    for (int j = 0; j < qm->ntpg; j++) {
      const QuadGeom& g = qm->trunk_geom[qm->tpg_slot[j]];
      const double half = g.type == MJPCX_GEOM_CAPSULE ? g.size[1] : 0.0, rad = g.type == MJPCX_GEOM_CAPSULE ? g.size[0] : qm->tpg_reach[j];
      for (int k = 0; k < 3; k++) {  // (the geom's axis in the trunk frame: the third column of its rotation)
        const double ext = half * std::fabs(g.rot[3 * k + 2]) + rad;
        qm->tpg_box[0][k] = std::min(qm->tpg_box[0][k], g.pos[k] - ext);
        qm->tpg_box[1][k] = std::max(qm->tpg_box[1][k], g.pos[k] + ext);
      }
    }
  }
  qm->npair = qt->npair;

  // ---- task: QuadrupedFlat residual (wave_residual.h / oracle/quadruped.inc), traces, cost terms
  if (task->residual_id != MJPCX_RESIDUAL_QUADRUPED_FLAT) return "residual is not QuadrupedFlat";
  if (task->num_residual_int < 17 || task->num_residual_real < 30) return "frozen residual state too short";
  const int32_t* ri = task->residual_int;
  if (ri[1] != trunk) return "torso body is not the floating base";
  if (ri[2] < 0 || ri[2] >= m->nsite || m->site_bodyid[ri[2]] != trunk) return "head site is not on the trunk";
  for (int k = 0; k < 3; k++) qm->head_pos[k] = m->site_pos[3 * ri[2] + k];
  qm->goal_mocap = ri[3];
  for (int f = 0; f < 4; f++) {
    const int g = ri[4 + f];
    if (g < 0 || g >= m->ngeom || slot_leg[g] < 0) return "a foot geom that is not a collidable leg geom";
    QuadLeg& L = qm->leg[slot_leg[g]];
    if (L.foot_slot >= 0) return "two foot geoms on one leg";
    L.foot_slot = slot_idx[g]; L.foot_index = f;
  }
  for (int l = 0; l < kQLegs; l++) if (qm->leg[l].foot_slot < 0) return "a leg without a foot geom";
  if (ri[15] < 0 || ri[15] >= m->nkey || ri[16] < 0 || ri[16] >= m->nkey) return "home / crouch keyframes missing";
  if (task->num_trace > kQMaxTrace) return "more traces than the quad kernel records";
  qm->ntrace = task->num_trace;
  for (int t = 0; t < task->num_trace; t++) {
    const int ts = task->trace_site[t];
    if (ts >= 0) {
      if (m->site_bodyid[ts] != trunk) return "a traced site off the trunk";
      for (int k = 0; k < 3; k++) qm->trace_pos[t][k] = m->site_pos[3 * ts + k];
    } else if (-1 - ts != trunk) return "a traced body other than the trunk";
  }
  if (task->num_term > kQMaxTerm) return "more cost terms than the quad kernel stages";
  qm->nterm = task->num_term; qm->nr = task->num_residual;
  if (qm->nr != 18 + 2 * m->nu) return "unexpected residual size";
  for (int k = 0, off = 0; k < task->num_term; k++) {
    qm->term_dim[k] = task->dim_norm_residual[k]; qm->term_norm[k] = task->norm[k]; qm->term_off[k] = off;
    off += task->dim_norm_residual[k];
  }
  // the kernel's residual code assumes the reference's term partition: Upright 3 | Height 1 | Position 3 | Gait 4 | Balance 2 |
  // Effort nu | Posture nu | Yaw 2 | Angmom 3
  const int want[9] = {3, 1, 3, 4, 2, m->nu, m->nu, 2, 3};
  if (task->num_term != 9) return "unexpected cost-term partition";
  for (int k = 0; k < 9; k++) if (task->dim_norm_residual[k] != want[k]) return "unexpected cost-term partition";
  return "";
}

}  // namespace mjpcx
