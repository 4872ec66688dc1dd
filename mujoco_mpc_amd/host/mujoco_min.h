// mujoco_min.h -- the slice of MuJoCo's public `mjModel` / `mjData` structs that the MJPC host code on
// the rollout path touches, with MuJoCo's own field names, element types and array strides
// (actuator_gear x6, actuator_gainprm/biasprm x mjNGAIN/mjNBIAS, actuator_trnid x2, limited flags as
// mjtByte). MuJoCo itself is a FetchContent dependency of the reference that is not available in this
// build environment; inside a real MJPC checkout this header is replaced by
// <mujoco/mujoco.h> and everything above it compiles unchanged.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

typedef double mjtNum;
typedef unsigned char mjtByte;

enum { mjNREF = 2, mjNIMP = 5, mjNGAIN = 10, mjNBIAS = 10 };
enum mjtJoint { mjJNT_FREE = 0, mjJNT_BALL, mjJNT_SLIDE, mjJNT_HINGE };
enum mjtSensor { mjSENS_USER = 100, mjSENS_FRAMEPOS = 25, mjSENS_OTHER = 0 };
enum mjtObj { mjOBJ_BODY = 1, mjOBJ_XBODY = 2, mjOBJ_GEOM = 5, mjOBJ_SITE = 6, mjOBJ_KEY = 21 };
enum mjtCone { mjCONE_PYRAMIDAL = 0, mjCONE_ELLIPTIC = 1 };
enum mjtWrap { mjWRAP_NONE = 0, mjWRAP_JOINT = 1 };
enum mjtBias { mjBIAS_NONE = 0, mjBIAS_AFFINE = 1 };

struct mjOption {
  mjtNum timestep;
  mjtNum gravity[3];
  mjtNum tolerance;
  int integrator;
  int iterations;
  int disableflags;
  int cone;
  mjtNum impratio;
};
struct mjStatistic {
  mjtNum meaninertia;
};

struct mjModel {
  int nq, nv, nu, na, nbody, njnt, nsite, nmocap, nuserdata, nsensor, nuser_sensor, nnumeric, ntext, nkey, ngeom;
  int ntendon, nwrap, nexclude;
  mjOption opt;
  mjStatistic stat;
  int *body_parentid, *body_rootid, *body_weldid, *body_jntnum, *body_jntadr, *body_dofnum, *body_dofadr, *body_mocapid;
  mjtNum *body_pos, *body_quat, *body_ipos, *body_iquat, *body_mass, *body_inertia, *body_invweight0, *body_subtreemass;
  int *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid;
  mjtByte* jnt_limited;
  mjtNum *jnt_pos, *jnt_axis, *jnt_stiffness, *jnt_range, *jnt_margin, *jnt_solref, *jnt_solimp;
  int *dof_bodyid, *dof_jntid, *dof_parentid;
  mjtNum *dof_armature, *dof_damping, *dof_frictionloss, *dof_invweight0, *dof_solref, *dof_solimp;
  int *geom_type, *geom_bodyid, *geom_contype, *geom_conaffinity, *geom_condim, *geom_priority, *geom_group;
  mjtNum *geom_size, *geom_pos, *geom_quat, *geom_friction, *geom_solref, *geom_solimp, *geom_margin, *geom_gap, *geom_solmix;
  mjtNum *qpos0, *qpos_spring;
  int* site_bodyid;
  mjtNum *site_pos, *site_quat;
  int *actuator_trnid, *actuator_gaintype, *actuator_biastype;
  mjtByte *actuator_ctrllimited, *actuator_forcelimited;
  mjtNum *actuator_gear, *actuator_gainprm, *actuator_biasprm, *actuator_ctrlrange, *actuator_forcerange;
  int *sensor_type, *sensor_objtype, *sensor_objid, *sensor_dim, *sensor_adr;
  mjtNum* sensor_user;
  int *numeric_adr, *numeric_size;
  mjtNum* numeric_data;
  mjtNum *key_qpos, *key_qvel, *key_mpos, *key_ctrl;
  int *tendon_adr, *tendon_num, *wrap_type, *wrap_objid, *exclude_signature;
  mjtByte* tendon_limited;
  mjtNum *wrap_prm, *tendon_range, *tendon_margin, *tendon_solref_lim, *tendon_solimp_lim, *tendon_invweight0;
  int *name_bodyadr, *name_jntadr, *name_siteadr, *name_sensoradr, *name_numericadr, *name_keyadr, *name_geomadr, *name_textadr;
  int *text_adr, *text_size;  // custom text fields: zero-terminated strings in text_data
  char* text_data;
  char* names;
};

struct mjData {
  mjtNum time;
  mjtNum *qpos, *qvel, *act, *ctrl, *mocap_pos, *mocap_quat, *userdata, *sensordata;
  // kinematics a Task::Transition may read (filled by the simulation; NULL when the caller has none):
  mjtNum *xpos, *xquat, *xmat, *xipos, *site_xpos, *subtree_com, *subtree_linvel;
};

// mj_makeData / mj_deleteData for the fields above (the state a planner thread's mjData carries: Planner::data_)
inline mjData* mj_makeData(const mjModel* m) {
  mjData* d = new mjData();
  auto arr = [](int n) { mjtNum* p = new mjtNum[n > 0 ? n : 1]; std::memset(p, 0, sizeof(mjtNum) * (n > 0 ? n : 1)); return p; };
  d->time = 0;
  d->qpos = arr(m->nq); d->qvel = arr(m->nv); d->act = arr(m->na); d->ctrl = arr(m->nu);
  d->mocap_pos = arr(3 * m->nmocap); d->mocap_quat = arr(4 * m->nmocap); d->userdata = arr(m->nuserdata);
  int nsensordata = 0;
  for (int i = 0; i < m->nsensor; i++) nsensordata = m->sensor_adr[i] + m->sensor_dim[i] > nsensordata ? m->sensor_adr[i] + m->sensor_dim[i] : nsensordata;
  d->sensordata = arr(nsensordata);
  if (m->qpos0) std::memcpy(d->qpos, m->qpos0, sizeof(mjtNum) * m->nq);
  for (int b = 0; b < m->nbody; b++) {  // mj_resetData: mocap bodies start at their model pose
    const int id = m->body_mocapid ? m->body_mocapid[b] : -1;
    if (id < 0) continue;
    std::memcpy(d->mocap_pos + 3 * id, m->body_pos + 3 * b, sizeof(mjtNum) * 3);
    std::memcpy(d->mocap_quat + 4 * id, m->body_quat + 4 * b, sizeof(mjtNum) * 4);
  }
  d->xpos = d->xquat = d->xmat = d->xipos = d->site_xpos = d->subtree_com = d->subtree_linvel = nullptr;
  return d;
}
// mj_resetDataKeyframe for the fields above: time 0, qpos / qvel / ctrl / mocap_pos of the key
inline void mj_resetDataKeyframe(const mjModel* m, mjData* d, int key) {
  if (key < 0 || key >= m->nkey) return;
  d->time = 0;
  std::memcpy(d->qpos, m->key_qpos + (size_t)key * m->nq, sizeof(mjtNum) * m->nq);
  std::memcpy(d->qvel, m->key_qvel + (size_t)key * m->nv, sizeof(mjtNum) * m->nv);
  if (m->key_ctrl) std::memcpy(d->ctrl, m->key_ctrl + (size_t)key * m->nu, sizeof(mjtNum) * m->nu);
  if (m->nmocap && m->key_mpos) std::memcpy(d->mocap_pos, m->key_mpos + (size_t)key * 3 * m->nmocap, sizeof(mjtNum) * 3 * m->nmocap);
}
inline void mj_deleteData(mjData* d) {
  if (!d) return;
  delete[] d->qpos; delete[] d->qvel; delete[] d->act; delete[] d->ctrl; delete[] d->mocap_pos; delete[] d->mocap_quat;
  delete[] d->userdata; delete[] d->sensordata;
  delete d;
}

#define mjMAX(a, b) (((a) > (b)) ? (a) : (b))
#define mjMIN(a, b) (((a) < (b)) ? (a) : (b))
inline void mju_copy(mjtNum* dst, const mjtNum* src, int n) { if (n > 0) std::memcpy(dst, src, sizeof(mjtNum) * n); }
inline void mju_addTo(mjtNum* res, const mjtNum* vec, int n) { for (int i = 0; i < n; i++) res[i] += vec[i]; }
inline void mju_scl(mjtNum* res, const mjtNum* vec, mjtNum scl, int n) { for (int i = 0; i < n; i++) res[i] = vec[i] * scl; }
inline void mju_zero(mjtNum* dst, int n) { if (n > 0) std::memset(dst, 0, sizeof(mjtNum) * n); }
inline mjtNum mju_max(mjtNum a, mjtNum b) { return a > b ? a : b; }
inline mjtNum mju_min(mjtNum a, mjtNum b) { return a < b ? a : b; }
inline mjtNum mju_clip(mjtNum x, mjtNum lo, mjtNum hi) { return mju_max(lo, mju_min(hi, x)); }
inline mjtNum mju_abs(mjtNum x) { return std::fabs(x); }
