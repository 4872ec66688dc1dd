// wave_model.h -- device-side model/task views of the wavefront-per-candidate kernel family (free joints,
// contacts, friction loss: the Quadruped class of models) and the host code that bakes them.
//
// Unlike the lane-per-candidate family (LaneModel by value in the kernarg segment), these models are too large
// for kernel arguments: every array lives in ONE device allocation and the kernel receives a struct of pointers
// into it. All reads of it are wave-uniform or lane-indexed gathers of a few hundred bytes that stay in the
// scalar / vector L1 after the first step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mjpcx.h"
#include "pair_cull.h"

namespace mjpcx {

constexpr int kWaveMaxBody = 64, kWaveMaxDof = 32, kWaveMaxGeom = 64, kWaveMaxLevel = 16;
constexpr int kWaveMaxCon = 16;   // contacts the row-table kernels stage per step: a candidate with more is FLAGGED (failure), not truncated --
constexpr int kWaveMaxEfc = 64;   // constraint rows likewise. (The oracle carries a MuJoCo-sized arena; the Jacobian-free kernels, which serve every
                                  // shipped contact model since round 3, have no such cap: the row-table kernels remain for RK4 models and A/B runs.)

template <typename T>
struct WaveModelT {
  int nq, nv, nu, nbody, njnt, nsite, nmocap, ngeom, nkey;
  int cone, disableflags, solver_iterations, any_damping, integrator;
  double timestep, gravity[3], solver_tolerance, meaninertia, impratio;
  const int *body_parentid, *body_rootid, *body_jntnum, *body_jntadr, *body_dofnum, *body_dofadr, *body_mocapid;
  const T *body_pos, *body_quat, *body_ipos, *body_iquat, *body_mass, *body_inertia, *body_invweight0, *body_subtreemass;
  const int *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *jnt_limited;
  const T *jnt_pos, *jnt_axis, *jnt_stiffness, *jnt_range, *jnt_margin, *jnt_solref, *jnt_solimp;
  const int *dof_bodyid, *dof_jntid, *dof_parentid;
  const T *dof_armature, *dof_damping, *dof_frictionloss, *dof_invweight0, *dof_solref, *dof_solimp;
  const T *qpos0, *qpos_spring;
  const int* site_bodyid;
  const T *site_pos, *site_quat;
  const int *actuator_trnid, *actuator_biastype, *actuator_ctrllimited, *actuator_forcelimited;
  const T *actuator_gear, *actuator_gainprm, *actuator_biasprm, *actuator_ctrlrange, *actuator_forcerange;
  const int *geom_type, *geom_bodyid, *geom_contype, *geom_conaffinity, *geom_condim, *geom_priority, *geom_group;
  const T *geom_size, *geom_pos, *geom_quat, *geom_friction, *geom_solref, *geom_solimp, *geom_margin, *geom_gap, *geom_solmix;
  const T* key_qpos;
  const T* key_mpos;                       // nkey x nmocap x 3 (Humanoid tracking residual); stays in HBM / L2
  int ntendon;                                  // fixed tendons (limits)
  const int *tendon_adr, *tendon_num, *tendon_limited, *wrap_objid;
  const T *wrap_prm, *tendon_range, *tendon_margin, *tendon_solref_lim, *tendon_solimp_lim, *tendon_invweight0;
  const unsigned* tendon_dofmask;               // baked: dofs a tendon's Jacobian touches
  // ---- baked helpers (host-computed once)
  const unsigned long long* body_subtree_mask;  // bit j: body j is in the subtree rooted at body i (incl. i)
  const unsigned* body_dofmask;                 // bit k: dof k is on the chain from body i to the root
  const int* level_body;                        // bodies 1..nbody-1 sorted by depth
  int nlevel;
  int level_start[kWaveMaxLevel + 1];           // level l = level_body[level_start[l] .. level_start[l+1])
  const int* static_geom;                       // collidable geoms on bodies without dofs (world, mocap), model order
  const int* dynamic_geom;                      // collidable geoms on moving bodies, model order
  int nstatic_geom, ndynamic_geom;
  int nbody_model;                              // nbody of the model (nbody above may exclude inert trailing bodies)
  const int *pair_g1, *pair_g2;                 // moving-geom pairs that pass MuJoCo's body filters (oracle: bake_pairs)
  int npair;
  int full;                                     // 1: rows beyond joint limits can occur (Newton path of the oracle)
  const int* ray_geom;                          // geoms of group 0 a downward ray can hit (plane / sphere / box), model order
  int nray_geom;
  const unsigned char* base;                    // the single device allocation all pointers above point into
  int bytes;                                    // its size (a multiple of 16): the kernel stages it into LDS
};
using WaveModel = WaveModelT<double>;  // the parity path, the iLQG kernels and the lane-per-candidate experiment

// every pointer member of WaveModel, for rebasing the struct onto a copy of the allocation (LDS staging)
#define MJPCX_WAVE_MODEL_POINTERS(X)                                                                                         \
  X(body_parentid) X(body_rootid) X(body_jntnum) X(body_jntadr) X(body_dofnum) X(body_dofadr) X(body_mocapid) X(body_pos)  \
  X(body_quat) X(body_ipos) X(body_iquat) X(body_mass) X(body_inertia) X(body_invweight0) X(body_subtreemass) X(jnt_type)  \
  X(jnt_qposadr) X(jnt_dofadr) X(jnt_bodyid) X(jnt_limited) X(jnt_pos) X(jnt_axis) X(jnt_stiffness) X(jnt_range)          \
  X(jnt_margin) X(jnt_solref) X(jnt_solimp) X(dof_bodyid) X(dof_jntid) X(dof_parentid) X(dof_armature) X(dof_damping)     \
  X(dof_frictionloss) X(dof_invweight0) X(dof_solref) X(dof_solimp) X(qpos0) X(qpos_spring) X(site_bodyid) X(site_pos)    \
  X(site_quat) X(actuator_trnid) X(actuator_biastype) X(actuator_ctrllimited) X(actuator_forcelimited) X(actuator_gear)   \
  X(actuator_gainprm) X(actuator_biasprm) X(actuator_ctrlrange) X(actuator_forcerange) X(geom_type) X(geom_bodyid)        \
  X(geom_contype) X(geom_conaffinity) X(geom_condim) X(geom_priority) X(geom_group) X(geom_size) X(geom_pos) X(geom_quat) \
  X(geom_friction) X(geom_solref) X(geom_solimp) X(geom_margin) X(geom_gap) X(geom_solmix) X(key_qpos) X(key_mpos)        \
  X(tendon_adr) X(tendon_num) X(tendon_limited) X(wrap_objid) X(wrap_prm) X(tendon_range) X(tendon_margin)                \
  X(tendon_solref_lim) X(tendon_solimp_lim) X(tendon_invweight0) X(tendon_dofmask) X(pair_g1) X(pair_g2)                  \
  X(body_subtree_mask) X(body_dofmask) X(level_body) X(static_geom) X(dynamic_geom) X(ray_geom)

// Per-plan task values: one small blob re-staged with every rollout (Planner::SetState + the frozen ResidualFn copy)
template <typename T>
struct WaveTaskT {
  int residual_id, nr, nterm, ntrace, nparam, nri, nrr;
  const int *dim_norm_residual, *norm, *trace_site;   // static, in the model allocation
  const int *term_off, *res_term;                     // baked: first residual entry of a term; term of a residual entry
  // blob (doubles): state[nq+nv] time mocap[7 nmocap] weight[nterm] norm_p[nterm] norm_q[nterm] parameters[nparam]
  //                 risk residual_real[nrr] ; then residual_int[nri] as int32
  const T* blob;
  long long* stamps;  // optional (tuning): s_memtime at the phase boundaries of step `stamp_step` of candidate 0
  int stamp_step;
  int off_time, off_mocap, off_weight, off_normp, off_normq, off_param, off_risk, off_rreal, off_rint;
};
using WaveTask = WaveTaskT<double>;

// ---------------------------------------------------------------- host side
struct WaveHost {
  WaveModel m{};
  WaveTask t{};
  void* dev = nullptr;  // the model allocation
  // fp32 twin (precision 32 contexts): same struct layout, every real array converted to float in its own allocation
  WaveModelT<float> m32{};
  WaveTaskT<float> t32{};
  void* dev32 = nullptr;
  size_t blob_bytes32 = 0;
  size_t blob_doubles = 0, blob_bytes = 0;
  bool tree_ok = false;  // the Jacobian-free constraint path (wave_tree.h) covers this model
  // host copies of the baked helpers (lds_model.h builds the LDS image of a registered model from them)
  std::vector<unsigned long long> h_subtree_mask;
  std::vector<unsigned> h_dofmask;
  std::vector<int> h_level_body, h_static_geom, h_dynamic_geom, h_ray_geom, h_term_off, h_res_term;
  void* dev_image = nullptr; void* dev_image32 = nullptr;  // LDS images of a registered model (fp64 / fp32)
  std::string warning;                                     // non-fatal findings of build() (reported through mjpcx_create_error after MJPCX_OK)
  int pairs_apart = 0;                                     // moving-geom pairs dropped because they are proven never to touch (pair_cull.h)
  int registered = -1;                                     // index into the registered configurations, -1: generic kernels
  std::vector<double> state, mocap, weight, norm_p, norm_q, parameters, residual_real;
  std::vector<int32_t> residual_int, norm_types;
  double time = 0, risk = 0;

  void release() {
    if (dev) (void)hipFree(dev); dev = nullptr; if (dev32) (void)hipFree(dev32); dev32 = nullptr;
    if (dev_image) (void)hipFree(dev_image); dev_image = nullptr; if (dev_image32) (void)hipFree(dev_image32); dev_image32 = nullptr;
  }

  // Returns "" or an error message. Requires the device to be current.
  std::string build(const mjpcx_model* src, const mjpcx_task* task, bool want32 = false) {
    if (src->nbody > kWaveMaxBody || src->nv > kWaveMaxDof || src->ngeom > 4 * kWaveMaxGeom) return "model exceeds the wave kernel capacity";
    if (src->nbody > 64) return "more than 64 bodies";
    static_assert(sizeof(WaveModelT<float>) == sizeof(WaveModel) && sizeof(WaveTaskT<float>) == sizeof(WaveTask), "twin layouts");
    std::vector<unsigned char> host, host32;
    auto put_in = [](std::vector<unsigned char>& h, const void* p, size_t bytes) -> size_t {
      size_t off = (h.size() + 15) & ~(size_t)15;
      h.resize(off + (bytes ? bytes : 16));
      if (bytes) std::memcpy(h.data() + off, p, bytes);
      return off;
    };
    struct Fix { size_t field_off; size_t data_off; size_t data_off32; };
    std::vector<Fix> fixes;
    // integer / mask arrays: identical in both images
    auto reg = [&](const void* field_addr, const void* p, size_t bytes) {
      Fix f{(size_t)((const char*)field_addr - (const char*)&m), put_in(host, p, bytes), 0};
      if (want32) f.data_off32 = put_in(host32, p, bytes);
      fixes.push_back(f);
    };
    // real arrays: doubles in the fp64 image, floats in the fp32 image
    std::vector<float> narrow;
    auto regd = [&](const void* field_addr, const double* p, size_t n) {
      Fix f{(size_t)((const char*)field_addr - (const char*)&m), put_in(host, p, sizeof(double) * n), 0};
      if (want32) {
        narrow.resize(n);
        for (size_t i = 0; i < n; i++) narrow[i] = (float)p[i];
        f.data_off32 = put_in(host32, narrow.data(), sizeof(float) * n);
      }
      fixes.push_back(f);
    };
    std::memset(&m, 0, sizeof m);
    m.nq = src->nq; m.nv = src->nv; m.nu = src->nu; m.nbody = src->nbody; m.njnt = src->njnt; m.nsite = src->nsite;
    m.nmocap = src->nmocap; m.ngeom = src->ngeom; m.nkey = src->nkey; m.cone = src->cone; m.disableflags = src->disableflags; m.integrator = src->integrator;
    m.solver_iterations = src->solver_iterations; m.timestep = src->timestep;
    for (int k = 0; k < 3; k++) m.gravity[k] = src->gravity[k];
    m.solver_tolerance = src->solver_tolerance; m.meaninertia = src->meaninertia; m.impratio = src->impratio;
    const int nb = src->nbody, nj = src->njnt, nv = src->nv, nu = src->nu, ns = src->nsite, ng = src->ngeom;
#define I(name, n) reg(&m.name, src->name, sizeof(int32_t) * (size_t)(n))
#define D(name, n) regd(&m.name, src->name, (size_t)(n))
    I(body_parentid, nb); I(body_rootid, nb); I(body_jntnum, nb); I(body_jntadr, nb); I(body_dofnum, nb); I(body_dofadr, nb); I(body_mocapid, nb);
    D(body_pos, 3 * nb); D(body_quat, 4 * nb); D(body_ipos, 3 * nb); D(body_iquat, 4 * nb); D(body_mass, nb); D(body_inertia, 3 * nb);
    D(body_invweight0, 2 * nb); D(body_subtreemass, nb);
    I(jnt_type, nj); I(jnt_qposadr, nj); I(jnt_dofadr, nj); I(jnt_bodyid, nj); I(jnt_limited, nj);
    D(jnt_pos, 3 * nj); D(jnt_axis, 3 * nj); D(jnt_stiffness, nj); D(jnt_range, 2 * nj); D(jnt_margin, nj); D(jnt_solref, 2 * nj); D(jnt_solimp, 5 * nj);
    I(dof_bodyid, nv); I(dof_jntid, nv); I(dof_parentid, nv);
    D(dof_armature, nv); D(dof_damping, nv); D(dof_frictionloss, nv); D(dof_invweight0, nv); D(dof_solref, 2 * nv); D(dof_solimp, 5 * nv);
    D(qpos0, src->nq); D(qpos_spring, src->nq);
    I(site_bodyid, ns); D(site_pos, 3 * ns); D(site_quat, 4 * ns);
    I(actuator_trnid, nu); I(actuator_biastype, nu); I(actuator_ctrllimited, nu); I(actuator_forcelimited, nu);
    D(actuator_gear, nu); D(actuator_gainprm, 3 * nu); D(actuator_biasprm, 3 * nu); D(actuator_ctrlrange, 2 * nu); D(actuator_forcerange, 2 * nu);
    I(geom_type, ng); I(geom_bodyid, ng); I(geom_contype, ng); I(geom_conaffinity, ng); I(geom_condim, ng); I(geom_priority, ng); I(geom_group, ng);
    D(geom_size, 3 * ng); D(geom_pos, 3 * ng); D(geom_quat, 4 * ng); D(geom_friction, 3 * ng); D(geom_solref, 2 * ng); D(geom_solimp, 5 * ng);
    D(geom_margin, ng); D(geom_gap, ng); D(geom_solmix, ng);
    D(key_qpos, (size_t)src->nkey * src->nq);
    if (src->key_mpos) D(key_mpos, (size_t)src->nkey * src->nmocap * 3); else regd(&m.key_mpos, nullptr, 0);
    const int nt = src->ntendon, nw = src->nwrap;
    m.ntendon = nt;
    I(tendon_adr, nt); I(tendon_num, nt); I(tendon_limited, nt); I(wrap_objid, nw);
    D(wrap_prm, nw); D(tendon_range, 2 * nt); D(tendon_margin, nt); D(tendon_solref_lim, 2 * nt); D(tendon_solimp_lim, 5 * nt); D(tendon_invweight0, nt);
#undef I
#undef D
    for (int i = 0; i < nv; i++) m.any_damping |= src->dof_damping[i] > 0;
    // Inert trailing bodies: the Humanoid tracking task appends 16 mocap marker bodies (no dofs, no geoms, visual sites) that
    // its residual never reads (it interpolates key_mpos itself). They are dropped from the device's body / site ranges:
    // 65 reals of LDS each. Only for that residual -- other tasks may read any body or site.
    int nb_live = nb, ns_live = ns;
    if (task->residual_id == MJPCX_RESIDUAL_HUMANOID_TRACK) {
      auto inert = [&](int b) {
        if (src->body_mocapid[b] < 0 || src->body_parentid[b] != 0 || src->body_dofnum[b] != 0) return false;
        for (int g = 0; g < ng; g++) if (src->geom_bodyid[g] == b) return false;
        for (int c2 = 0; c2 < nb; c2++) if (c2 != b && src->body_parentid[c2] == b) return false;
        return true;
      };
      while (nb_live > 1 && inert(nb_live - 1)) nb_live--;
      while (ns_live > 0 && src->site_bodyid[ns_live - 1] >= nb_live) ns_live--;
      for (int k = 0; k < task->num_trace; k++) {
        const int ts = task->trace_site[k];
        if ((ts >= 0 && ts >= ns_live) || (ts < 0 && -1 - ts >= nb_live)) { nb_live = nb; ns_live = ns; break; }
      }
      for (int i = 0; i < ns_live; i++) if (src->site_bodyid[i] >= nb_live) { nb_live = nb; ns_live = ns; break; }
    }
    // baked helpers
    std::vector<unsigned long long> sub(nb, 0);
    std::vector<unsigned> dofmask(nb, 0);
    std::vector<int> depth(nb, 0);
    for (int i = 0; i < nb; i++) {
      if (i < nb_live) for (int b = i; ; b = src->body_parentid[b]) { sub[b] |= 1ull << i; if (b == 0) break; }
      if (i > 0) depth[i] = depth[src->body_parentid[i]] + 1;
      unsigned mk = i > 0 ? dofmask[src->body_parentid[i]] : 0u;
      for (int k = 0; k < src->body_dofnum[i]; k++) mk |= 1u << (src->body_dofadr[i] + k);
      dofmask[i] = mk;
    }
    int maxdepth = 0;
    for (int i = 1; i < nb; i++) maxdepth = depth[i] > maxdepth ? depth[i] : maxdepth;
    if (maxdepth > kWaveMaxLevel) return "kinematic tree deeper than the wave kernel supports";
    std::vector<int> level_body;
    m.nlevel = maxdepth;
    for (int l = 1; l <= maxdepth; l++) {
      m.level_start[l - 1] = (int)level_body.size();
      for (int i = 1; i < nb_live; i++) if (depth[i] == l) level_body.push_back(i);
    }
    m.level_start[maxdepth] = (int)level_body.size();
    std::vector<int> sg, dg;
    bool any_floss = false;
    for (int i = 0; i < nv; i++) any_floss |= src->dof_frictionloss[i] > 0 && !(src->disableflags & MJPCX_DSBL_FRICTIONLOSS);
    for (int g = 0; g < ng; g++) {
      if (!(src->geom_contype[g] || src->geom_conaffinity[g])) continue;
      (dofmask[src->geom_bodyid[g]] == 0 ? sg : dg).push_back(g);
    }
    if ((int)dg.size() > 64) return "more than 64 collidable geoms on moving bodies";
    m.nstatic_geom = (int)sg.size(); m.ndynamic_geom = (int)dg.size();
    m.full = any_floss || (!sg.empty() && !dg.empty() && !(src->disableflags & MJPCX_DSBL_CONTACT));
    std::vector<unsigned> tmask(nt > 0 ? nt : 1, 0u);
    for (int t = 0; t < nt; t++) {
      m.full |= src->tendon_limited[t] != 0;
      for (int w = src->tendon_adr[t]; w < src->tendon_adr[t] + src->tendon_num[t]; w++) tmask[t] |= 1u << src->jnt_dofadr[src->wrap_objid[w]];
    }
    reg(&m.tendon_dofmask, tmask.data(), sizeof(unsigned) * (size_t)nt);
    // moving-geom pairs after MuJoCo's body filters (pair_cull.h: the list, each pair's class, the proofs); canonical order = lower geom
    // type first. The kernels collide sphere | capsule pairs and (sphere | capsule) x (box | cylinder) pairs (solid_pairs.h).
    std::vector<int> pg1, pg2;
    int skipped_pairs = 0, skipped_a = -1, skipped_b = -1;
    pairs_apart = 0;
    {
      const bool contacts_on = !(src->disableflags & (MJPCX_DSBL_CONTACT | MJPCX_DSBL_CONSTRAINT));
      std::vector<char> moving(nb);
      for (int b = 0; b < nb; b++) moving[b] = dofmask[b] != 0;
      std::vector<MovingPair> mp;
      moving_pairs(src, moving, contacts_on, mp);
      for (const MovingPair& q : mp) {
        if (q.kind == kPairThin || q.kind == kPairThinSolid) {  // (collided whether proven apart or not: these kernels have the narrow phase)
          pg1.push_back(q.g1);
          pg2.push_back(q.g2);
          continue;
        }
        // two solids: no narrow phase anywhere. The pair stays in the list and is WATCHED (a thin geom that contains one of them within the
        // margin of the other raises warning bit 128: the rollout fails, as the oracle's does) -- proven apart or not: a proof holds for
        // joints inside their ranges + pad, and a caller may start a rollout anywhere. One that cannot be proven apart is REPORTED too
        // (mjpcx_create_error() after MJPCX_OK; MJPCX_STRICT_PAIRS refuses the model).
        if (q.kind == kPairSolids) {
          // (already within reach at the reference pose -- boxes resting on each other, say: every rollout from there would fail at its
          // first step; one clear refusal instead)
          if (contacts_on && solids_touch_at_qpos0(src, q.g1, q.g2)) {
            return "geoms " + std::to_string(q.g1) + " and " + std::to_string(q.g2) + " (box | cylinder both, on two moving bodies) are within their contact margin at "
                       "qpos0, and a pair of two such solids has no narrow phase here: every rollout from there would fail (warning bit 128)";
          }
          pg1.push_back(q.g1);
          pg2.push_back(q.g2);
          if (q.apart && q.tight_jnt < 0) { pairs_apart++; continue; }
        }
        if (contacts_on && skipped_pairs++ == 0) { skipped_a = q.g1; skipped_b = q.g2; }
      }
    }
    warning.clear();
    if (skipped_pairs > 0)
      warning = std::to_string(skipped_pairs) + " collidable geom pair(s) between two moving bodies have no narrow phase here (two box | cylinder geoms that could not be "
                "proven apart over the joint ranges, or a geom type other than sphere | capsule | cylinder | box; first: geoms " + std::to_string(skipped_a) + ", " +
                std::to_string(skipped_b) + ") and are NOT collided: a rollout in which two such solids come within reach FAILS (contacts with static geoms are unaffected)";
    m.npair = (int)pg1.size();
    reg(&m.pair_g1, pg1.data(), sizeof(int) * pg1.size());
    reg(&m.pair_g2, pg2.data(), sizeof(int) * pg2.size());
    for (int j = 0; j < nj; j++) m.full |= src->jnt_type[j] == MJPCX_JNT_FREE || src->jnt_type[j] == MJPCX_JNT_BALL;
    {
      // wave_tree.h: one moving kinematic tree (all cdof about the same point), no limited tendons, elliptic cones or
      // frictionless contacts only
      int root = -1;
      bool one_tree = true, limited_tendon = false;
      int max_condim = 1;
      for (int b = 1; b < nb; b++)
        if (dofmask[b] != 0) { if (root < 0) root = src->body_rootid[b]; else one_tree &= src->body_rootid[b] == root; }
      for (int t = 0; t < nt; t++) limited_tendon |= src->tendon_limited[t] != 0;
      for (int g = 0; g < ng; g++)
        if (src->geom_contype[g] || src->geom_conaffinity[g]) max_condim = src->geom_condim[g] > max_condim ? src->geom_condim[g] : max_condim;
      // (nb_live <= nv: the per-body Newton work vectors reuse cdof_dot's storage, wave_carve_tree)
      // (pyramidal cones and limits of fixed tendons are covered since round 3; spatial tendons are refused at create)
      (void)limited_tendon; (void)max_condim;
      tree_ok = one_tree && nv <= 32 && nb_live <= nv && nt <= 64;
    }
    reg(&m.body_subtree_mask, sub.data(), sizeof(unsigned long long) * nb);
    reg(&m.body_dofmask, dofmask.data(), sizeof(unsigned) * nb);
    reg(&m.level_body, level_body.data(), sizeof(int) * level_body.size());
    reg(&m.static_geom, sg.data(), sizeof(int) * sg.size());
    reg(&m.dynamic_geom, dg.data(), sizeof(int) * dg.size());
    std::vector<int> rg;
    for (int g = 0; g < ng; g++)
      if (src->geom_group[g] == 0 && (src->geom_type[g] == MJPCX_GEOM_PLANE || src->geom_type[g] == MJPCX_GEOM_SPHERE || src->geom_type[g] == MJPCX_GEOM_BOX))
        rg.push_back(g);
    m.nray_geom = (int)rg.size();
    reg(&m.ray_geom, rg.data(), sizeof(int) * rg.size());
    // static task arrays
    t.residual_id = task->residual_id; t.nr = task->num_residual; t.nterm = task->num_term; t.ntrace = task->num_trace;
    t.nparam = task->num_parameter; t.nri = task->num_residual_int; t.nrr = task->num_residual_real;
    auto put2 = [&](const void* p, size_t bytes) { return std::pair<size_t, size_t>(put_in(host, p, bytes), want32 ? put_in(host32, p, bytes) : 0); };
    const auto o_dim = put2(task->dim_norm_residual, sizeof(int32_t) * task->num_term);
    const auto o_norm = put2(task->norm, sizeof(int32_t) * task->num_term);
    const auto o_trace = put2(task->trace_site, sizeof(int32_t) * task->num_trace);
    std::vector<int32_t> term_off(task->num_term > 0 ? task->num_term : 1, 0), res_term(task->num_residual > 0 ? task->num_residual : 1, 0);
    for (int k = 0, off = 0; k < task->num_term; k++) {
      term_off[k] = off;
      for (int i = 0; i < task->dim_norm_residual[k] && off + i < task->num_residual; i++) res_term[off + i] = k;
      off += task->dim_norm_residual[k];
    }
    h_subtree_mask = sub; h_dofmask = dofmask; h_level_body = level_body; h_static_geom = sg; h_dynamic_geom = dg; h_ray_geom = rg;
    h_term_off.assign(term_off.begin(), term_off.end()); h_res_term.assign(res_term.begin(), res_term.end());
    const auto o_toff = put2(term_off.data(), sizeof(int32_t) * term_off.size());
    const auto o_rterm = put2(res_term.data(), sizeof(int32_t) * res_term.size());
    host.resize((host.size() + 15) & ~(size_t)15);
    if (hipMalloc(&dev, host.size()) != hipSuccess) return "hipMalloc of the model failed";
    if (hipMemcpy(dev, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) return "model upload failed";
    for (const Fix& f : fixes) *(const void**)((char*)&m + f.field_off) = (const char*)dev + f.data_off;
    t.dim_norm_residual = (const int*)((const char*)dev + o_dim.first);
    t.norm = (const int*)((const char*)dev + o_norm.first);
    t.trace_site = (const int*)((const char*)dev + o_trace.first);
    t.term_off = (const int*)((const char*)dev + o_toff.first);
    t.res_term = (const int*)((const char*)dev + o_rterm.first);
    m.base = (const unsigned char*)dev;
    m.bytes = (int)host.size();
    m.nbody = nb_live; m.nsite = ns_live;  // device ranges (the arrays keep the model's sizes)
    m.nbody_model = nb;
    // blob layout
    int o = 0;
    auto seg = [&](int n) { int at = o; o += n; return at; };
    seg(src->nq + src->nv);
    t.off_time = seg(1); t.off_mocap = seg(7 * src->nmocap); t.off_weight = seg(task->num_term); t.off_normp = seg(task->num_term);
    t.off_normq = seg(task->num_term); t.off_param = seg(task->num_parameter); t.off_risk = seg(1); t.off_rreal = seg(task->num_residual_real);
    t.off_rint = o;
    blob_doubles = (size_t)o;
    blob_bytes = blob_doubles * 8 + sizeof(int32_t) * (size_t)task->num_residual_int;
    blob_bytes = (blob_bytes + 15) & ~(size_t)15;
    if (want32) {  // the fp32 twin: same scalars and baked integers, float arrays, its own allocation
      std::memcpy((void*)&m32, (const void*)&m, sizeof m);
      host32.resize((host32.size() + 15) & ~(size_t)15);
      if (hipMalloc(&dev32, host32.size()) != hipSuccess) return "hipMalloc of the fp32 model failed";
      if (hipMemcpy(dev32, host32.data(), host32.size(), hipMemcpyHostToDevice) != hipSuccess) return "fp32 model upload failed";
      for (const Fix& f : fixes) *(const void**)((char*)&m32 + f.field_off) = (const char*)dev32 + f.data_off32;
      m32.base = (const unsigned char*)dev32;
      m32.bytes = (int)host32.size();
      std::memcpy((void*)&t32, (const void*)&t, sizeof t);
      t32.dim_norm_residual = (const int*)((const char*)dev32 + o_dim.second);
      t32.norm = (const int*)((const char*)dev32 + o_norm.second);
      t32.trace_site = (const int*)((const char*)dev32 + o_trace.second);
      t32.term_off = (const int*)((const char*)dev32 + o_toff.second);
      t32.res_term = (const int*)((const char*)dev32 + o_rterm.second);
      blob_bytes32 = (blob_doubles * 4 + sizeof(int32_t) * (size_t)task->num_residual_int + 15) & ~(size_t)15;
    }
    // host mirrors of the per-plan values
    state.assign(src->nq + src->nv, 0.0);
    for (int i = 0; i < src->nq; i++) state[i] = src->qpos0[i];
    mocap.assign(7 * (size_t)src->nmocap, 0.0);
    for (int b = 0; b < nb; b++)
      if (src->body_mocapid[b] >= 0) {
        for (int k = 0; k < 3; k++) mocap[7 * src->body_mocapid[b] + k] = src->body_pos[3 * b + k];
        for (int k = 0; k < 4; k++) mocap[7 * src->body_mocapid[b] + 3 + k] = src->body_quat[4 * b + k];
      }
    weight.assign(task->weight, task->weight + task->num_term);
    norm_types.assign(task->norm, task->norm + task->num_term);
    norm_p.assign(task->num_term, 0.0); norm_q.assign(task->num_term, 0.0);
    parameters.assign(task->parameters, task->parameters + task->num_parameter);
    residual_real.assign(task->residual_real, task->residual_real + task->num_residual_real);
    residual_int.assign(task->residual_int, task->residual_int + task->num_residual_int);
    risk = task->risk;
    return "";
  }

  // the same values narrowed to float (blob_bytes32)
  void fill_blob32(void* dst) const {
    float* d = (float*)dst;
    auto cp = [&](int off, const std::vector<double>& v) { for (size_t i = 0; i < v.size(); i++) d[off + i] = (float)v[i]; };
    cp(0, state);
    d[t.off_time] = (float)time;
    cp(t.off_mocap, mocap); cp(t.off_weight, weight); cp(t.off_normp, norm_p); cp(t.off_normq, norm_q); cp(t.off_param, parameters);
    d[t.off_risk] = (float)risk;
    cp(t.off_rreal, residual_real);
    std::memcpy(d + t.off_rint, residual_int.data(), residual_int.size() * sizeof(int32_t));
  }

  // serialise the per-plan values into `dst` (blob_bytes)
  void fill_blob(void* dst) const {
    double* d = (double*)dst;
    std::memcpy(d, state.data(), state.size() * 8);
    d[t.off_time] = time;
    std::memcpy(d + t.off_mocap, mocap.data(), mocap.size() * 8);
    std::memcpy(d + t.off_weight, weight.data(), weight.size() * 8);
    std::memcpy(d + t.off_normp, norm_p.data(), norm_p.size() * 8);
    std::memcpy(d + t.off_normq, norm_q.data(), norm_q.size() * 8);
    std::memcpy(d + t.off_param, parameters.data(), parameters.size() * 8);
    d[t.off_risk] = risk;
    std::memcpy(d + t.off_rreal, residual_real.data(), residual_real.size() * 8);
    std::memcpy(d + t.off_rint, residual_int.data(), residual_int.size() * sizeof(int32_t));
  }
};

}  // namespace mjpcx
