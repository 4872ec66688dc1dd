/* physics.c -- oracle restatement of the MuJoCo forward-dynamics pipeline.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h). PARITY UNPINNED against MuJoCo:
 * MuJoCo (github.com/google-deepmind/mujoco @ 088079eff0450e32b98ee743141780ed68307506,
 * pinned by /root/reference/CMakeLists.txt:58-61) is not vendored in the
 * reference tree and not installed here. This file restates the published
 * pipeline of `mj_step` / `mj_forward` -- the calls made by
 * mjpc/trajectory.cc:158 and :198 -- stage by stage:
 *
 *   mj_step   = checkPos, checkVel, mj_forward, checkAcc, Euler (implicit joint damping)
 *   mj_forward= kinematics, comPos, crb, factorM, makeConstraint(limits),
 *               comVel, passive, rne, actuation, acceleration, constraint solve (PGS),
 *               [sensorAcc -> task residual via mjcb_sensor, mjpc/app.cc:110-126]
 *
 * Feature subset (this file + contact.inc, humanoid.inc, quadruped.inc): free/ball/slide/hinge joints in an arbitrary tree, joint
 * springs/dampers/armature, gravity, joint-transmission actuators (fixed gain, none/affine bias, ctrl/force clamps), friction loss,
 * slide/hinge joint limits and limits of fixed tendons as soft constraints (solref/solimp), contacts with elliptic and pyramidal
 * friction cones (condim 1/3/4/6) between spheres / capsules / boxes / cylinders and planes, spheres and static spheres / boxes,
 * spheres / capsules of two moving bodies behind MuJoCo's body-pair filters, primal Newton solver with an exact line search (a PGS
 * solver of the dual for cross-checks), Euler (implicit joint damping) and RK4 integrators, xfrc_applied. NOT restated: pairs of
 * moving geoms that MuJoCo hands to its general convex collider (boxes, cylinders: reported by odata_new's caller and by
 * mjpcx_create), spatial tendons, equality constraints, implicit integrators (models that need them are rejected).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define OMINVAL 1e-15 /* mjMINVAL */
#define OMINIMP 0.0001 /* mjMINIMP: solimp d0, d_width, midpoint are clipped to [mjMINIMP, mjMAXIMP] */
#define OMAXIMP 0.9999 /* mjMAXIMP */
#define OLS_TOLERANCE 0.01 /* mjOption.ls_tolerance default (not carried by mjpcx_model) */
#define OMAXVAL 1e10  /* mjMAXVAL */
#define OMAXEFC 1024 /* MuJoCo grows its arena on demand; the oracle carries a MuJoCo-sized one (cap-induced failures are the device's to avoid) */
#define OMAXCON 256
#define OMINMU 1e-5 /* mjMINMU */

typedef struct OContact {
  int g1, g2, dim, efc;
  int nrow; /* rows built: dim for a frictionless / elliptic contact, 2 (dim - 1) pyramid edges otherwise */
  double dist, margin, includemargin, mu;
  double pos[3], frame[9], friction[5], solref[2], solimp[5];
} OContact;

struct OData {
  const mjpcx_model* m;
  int nq, nv, nu, nbody, njnt, nsite;
  double time;
  double *qpos, *qvel, *ctrl, *mocap_pos, *mocap_quat, *userdata;
  double* xfrc_applied; /* 6 nbody: Cartesian force, torque at each body's centre of mass (mjData.xfrc_applied) */
  double *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis;
  double *site_xpos, *site_xmat, *subtree_com, *subtree_mass;
  double *cinert, *crb, *cdof, *cdof_dot, *cvel, *cacc, *cfrc;
  double *M, *L; /* dense nv x nv */
  double *qfrc_passive, *qfrc_bias, *qfrc_actuator, *qfrc_smooth, *qacc_smooth;
  double *qfrc_constraint, *qacc, *actuator_force;
  int nefc;
  int efc_jnt[OMAXEFC];
  double *efc_J; /* OMAXEFC x nv */
  double efc_pos[OMAXEFC], efc_margin[OMAXEFC], efc_R[OMAXEFC], efc_aref[OMAXEFC];
  double efc_b[OMAXEFC], efc_force[OMAXEFC];
  double *scratch_minvjt, *scratch_qacc, *scratch_A, *scratch_AR; /* preallocated work arrays */
  double* rk_scratch; /* mj_RungeKutta: X0 (nq + nv), F (4 x 2 nv), dX (2 nv) */
  /* contacts / friction loss / Newton solver (contact.inc) */
  int full; /* 1: constraint rows beyond joint limits can occur -> Newton path */
  int ngeom, ncon, solver_iter;
  int* geom_static;
  int npair; int *pair_g1, *pair_g2; /* moving-geom pairs that pass the body filters (baked once) */
  double *geom_xpos, *geom_xmat;
  OContact con[OMAXCON];
  int efc_type[OMAXEFC], efc_id[OMAXEFC], efc_zone[OMAXEFC];
  double efc_D[OMAXEFC], efc_floss[OMAXEFC], efc_vel[OMAXEFC];
  double *scratch_jac, *nw_jar, *nw_jv, *nw_grad, *nw_search, *nw_Ma, *nw_H, *nw_L;
  double* qacc_warmstart; /* previous step's qacc (mj_advance); valid iff have_warm */
  int have_warm;
  int warning;
};

/* ------------------------------------------------------------------ helpers */
static double* dalloc(int n) { return (double*)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); }
static void mul_quat(double r[4], const double a[4], const double b[4]) {
  double t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
}
static void quat2mat(double m[9], const double q[4]) {
  double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  double q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3];
  double q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2 * (q12 - q03); m[2] = 2 * (q13 + q02);
  m[3] = 2 * (q12 + q03); m[5] = 2 * (q23 - q01);
  m[6] = 2 * (q13 - q02); m[7] = 2 * (q23 + q01);
}
static void rot_vec_quat(double r[3], const double v[3], const double q[4]) {
  double m[9];
  quat2mat(m, q);
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  double y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  double z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void mul_mat_vec3(double r[3], const double m[9], const double v[3]) {
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  double y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  double z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void normalize4(double q[4]) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < OMINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  double s = 1.0 / n;
  q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
}
static void axis_angle2quat(double q[4], const double axis[3], double angle) {
  if (angle == 0) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  double s = sin(0.5 * angle);
  q[0] = cos(0.5 * angle); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
static void cross3(double r[3], const double a[3], const double b[3]) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
/* spatial inertia (about the subtree-com reference point) of a body: 10 numbers
 * [Ixx Iyy Izz Ixy Ixz Iyz, m*dx m*dy m*dz, m] */
static void inert_com(double res[10], const double inert[3], const double mat[9],
                      const double dif[3], double mass) {
  double tmp[9];
  for (int c = 0; c < 3; c++) { /* tmp = diag(inert) * mat' */
    tmp[0 + c] = inert[0] * mat[3 * c + 0];
    tmp[3 + c] = inert[1] * mat[3 * c + 1];
    tmp[6 + c] = inert[2] * mat[3 * c + 2];
  }
  res[0] = mat[0] * tmp[0] + mat[1] * tmp[3] + mat[2] * tmp[6];
  res[1] = mat[3] * tmp[1] + mat[4] * tmp[4] + mat[5] * tmp[7];
  res[2] = mat[6] * tmp[2] + mat[7] * tmp[5] + mat[8] * tmp[8];
  res[3] = mat[0] * tmp[1] + mat[1] * tmp[4] + mat[2] * tmp[7];
  res[4] = mat[0] * tmp[2] + mat[1] * tmp[5] + mat[2] * tmp[8];
  res[5] = mat[3] * tmp[2] + mat[4] * tmp[5] + mat[5] * tmp[8];
  res[0] += mass * (dif[1] * dif[1] + dif[2] * dif[2]);
  res[1] += mass * (dif[0] * dif[0] + dif[2] * dif[2]);
  res[2] += mass * (dif[0] * dif[0] + dif[1] * dif[1]);
  res[3] -= mass * dif[0] * dif[1];
  res[4] -= mass * dif[0] * dif[2];
  res[5] -= mass * dif[1] * dif[2];
  res[6] = mass * dif[0]; res[7] = mass * dif[1]; res[8] = mass * dif[2];
  res[9] = mass;
}
/* res = I * v, v = [angular; linear] */
static void mul_inert_vec(double res[6], const double i[10], const double v[6]) {
  res[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  res[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  res[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  res[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  res[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  res[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
static void cross_motion(double res[6], const double vel[6], const double v[6]) {
  double a[3], b[3], c[3];
  cross3(a, vel, v);         /* w x v_ang */
  cross3(b, vel, v + 3);     /* w x v_lin */
  cross3(c, vel + 3, v);     /* u x v_ang */
  res[0] = a[0]; res[1] = a[1]; res[2] = a[2];
  res[3] = b[0] + c[0]; res[4] = b[1] + c[1]; res[5] = b[2] + c[2];
}
static void cross_force(double res[6], const double vel[6], const double f[6]) {
  double a[3], b[3], c[3];
  cross3(a, vel, f);         /* w x f_ang */
  cross3(b, vel + 3, f + 3); /* u x f_lin */
  cross3(c, vel, f + 3);     /* w x f_lin */
  res[0] = a[0] + b[0]; res[1] = a[1] + b[1]; res[2] = a[2] + b[2];
  res[3] = c[0]; res[4] = c[1]; res[5] = c[2];
}
static double dot6(const double a[6], const double b[6]) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
static int is_bad(double x) { return isnan(x) || x > OMAXVAL || x < -OMAXVAL; }

void oresidual(const mjpcx_task* task, const OData* d, double* r);

/* ------------------------------------------------------------------ lifetime */
static int body_is_static(const mjpcx_model* m, int b);
static void bake_pairs(OData* d);
OData* odata_new(const mjpcx_model* m) {
  /* reject features this restatement does not cover */
  for (int j = 0; j < m->njnt; j++)
    if (m->jnt_limited[j] && (m->jnt_type[j] == MJPCX_JNT_BALL || m->jnt_type[j] == MJPCX_JNT_FREE))
      return NULL;
  if (m->integrator == MJPCX_INT_IMPLICITFAST) {
    /* mj_implicit's matrix M - h dqfrc_smooth/dqvel is mj_Euler's M + h diag(damping) when no actuator force depends on velocity
     * (include/mjpcx.h): such a model steps through o_euler; any other is rejected */
    for (int i = 0; i < m->nu; i++)
      if (m->actuator_biasprm && m->actuator_biastype && m->actuator_biastype[i] == MJPCX_BIAS_AFFINE && m->actuator_biasprm[3 * i + 2] != 0) return NULL;
    if (m->disableflags & MJPCX_DSBL_EULERDAMP) return NULL; /* mj_implicit ignores the flag, o_euler honours it: not the same update */
  } else if (m->integrator != MJPCX_INT_EULER && m->integrator != MJPCX_INT_RK4) return NULL;
  if (m->na != 0) return NULL;

  OData* d = (OData*)calloc(1, sizeof(OData));
  d->m = m;
  int nq = d->nq = m->nq, nv = d->nv = m->nv, nu = d->nu = m->nu;
  int nb = d->nbody = m->nbody, nj = d->njnt = m->njnt, ns = d->nsite = m->nsite;
  d->qpos = dalloc(nq); d->qvel = dalloc(nv); d->ctrl = dalloc(nu);
  d->mocap_pos = dalloc(3 * m->nmocap); d->mocap_quat = dalloc(4 * m->nmocap);
  d->userdata = dalloc(m->nuserdata);
  d->xpos = dalloc(3 * nb); d->xquat = dalloc(4 * nb); d->xmat = dalloc(9 * nb);
  d->xipos = dalloc(3 * nb); d->ximat = dalloc(9 * nb);
  d->xanchor = dalloc(3 * nj); d->xaxis = dalloc(3 * nj);
  d->site_xpos = dalloc(3 * ns); d->site_xmat = dalloc(9 * ns);
  d->subtree_com = dalloc(3 * nb); d->subtree_mass = dalloc(nb);
  d->cinert = dalloc(10 * nb); d->crb = dalloc(10 * nb);
  d->cdof = dalloc(6 * nv); d->cdof_dot = dalloc(6 * nv);
  d->cvel = dalloc(6 * nb); d->cacc = dalloc(6 * nb); d->cfrc = dalloc(6 * nb);
  d->M = dalloc(nv * nv); d->L = dalloc(nv * nv);
  d->qfrc_passive = dalloc(nv); d->qfrc_bias = dalloc(nv); d->qfrc_actuator = dalloc(nv);
  d->qfrc_smooth = dalloc(nv); d->qacc_smooth = dalloc(nv); d->qfrc_constraint = dalloc(nv);
  d->qacc = dalloc(nv); d->actuator_force = dalloc(nu);
  d->efc_J = dalloc(OMAXEFC * nv);
  d->scratch_minvjt = dalloc(OMAXEFC * nv); d->scratch_qacc = dalloc(nv); d->scratch_A = dalloc(2 * nv * nv);
  d->scratch_AR = dalloc((2 * nj + 1) * (2 * nj + 1));
  d->rk_scratch = dalloc(nq + nv + 10 * nv);
  /* geoms and the Newton solver's work space */
  d->ngeom = m->ngeom;
  d->geom_xpos = dalloc(3 * m->ngeom); d->geom_xmat = dalloc(9 * m->ngeom);
  d->geom_static = (int*)calloc((size_t)(m->ngeom > 0 ? m->ngeom : 1), sizeof(int));
  d->scratch_jac = dalloc(12 * nv); d->nw_jar = dalloc(OMAXEFC); d->nw_jv = dalloc(OMAXEFC);
  d->nw_grad = dalloc(nv); d->nw_search = dalloc(nv); d->nw_Ma = dalloc(nv); d->nw_H = dalloc(nv * nv); d->nw_L = dalloc(nv * nv);
  d->qacc_warmstart = dalloc(nv); d->have_warm = 0;
  d->xfrc_applied = dalloc(6 * nb);
  d->full = 0;
  for (int i = 0; i < nv; i++) if (m->dof_frictionloss[i] > 0 && !(m->disableflags & MJPCX_DSBL_FRICTIONLOSS)) d->full = 1;
  int nstatic = 0, ndynamic = 0;
  for (int g = 0; g < m->ngeom; g++) {
    d->geom_static[g] = body_is_static(m, m->geom_bodyid[g]);
    if (m->geom_contype[g] || m->geom_conaffinity[g]) { if (d->geom_static[g]) nstatic++; else ndynamic++; }
  }
  if (nstatic && ndynamic && !(m->disableflags & MJPCX_DSBL_CONTACT)) d->full = 1;
  for (int t = 0; t < m->ntendon; t++) if (m->tendon_limited[t]) d->full = 1;
  bake_pairs(d);
  for (int j = 0; j < m->njnt; j++) if (m->jnt_type[j] == MJPCX_JNT_FREE || m->jnt_type[j] == MJPCX_JNT_BALL) d->full = 1;
  /* subtree masses are model constants */
  for (int i = 0; i < nb; i++) d->subtree_mass[i] = m->body_mass[i];
  for (int i = nb - 1; i > 0; i--) d->subtree_mass[m->body_parentid[i]] += d->subtree_mass[i];
  /* default state = qpos0, mocap at model pose */
  memcpy(d->qpos, m->qpos0, sizeof(double) * nq);
  for (int i = 0; i < nb; i++)
    if (m->body_mocapid[i] >= 0) {
      memcpy(d->mocap_pos + 3 * m->body_mocapid[i], m->body_pos + 3 * i, 3 * sizeof(double));
      memcpy(d->mocap_quat + 4 * m->body_mocapid[i], m->body_quat + 4 * i, 4 * sizeof(double));
    }
  return d;
}
void odata_free(OData* d) {
  if (!d) return;
  double** p[] = {&d->qpos, &d->qvel, &d->ctrl, &d->mocap_pos, &d->mocap_quat, &d->userdata,
                  &d->xpos, &d->xquat, &d->xmat, &d->xipos, &d->ximat, &d->xanchor, &d->xaxis,
                  &d->site_xpos, &d->site_xmat, &d->subtree_com, &d->subtree_mass, &d->cinert,
                  &d->crb, &d->cdof, &d->cdof_dot, &d->cvel, &d->cacc, &d->cfrc, &d->M, &d->L,
                  &d->qfrc_passive, &d->qfrc_bias, &d->qfrc_actuator, &d->qfrc_smooth,
                  &d->qacc_smooth, &d->qfrc_constraint, &d->qacc, &d->actuator_force, &d->efc_J,
                  &d->scratch_minvjt, &d->scratch_qacc, &d->scratch_A, &d->scratch_AR, &d->rk_scratch, &d->geom_xpos, &d->geom_xmat, &d->scratch_jac,
                  &d->nw_jar, &d->nw_jv, &d->nw_grad, &d->nw_search, &d->nw_Ma, &d->nw_H, &d->nw_L, &d->qacc_warmstart,
                  &d->xfrc_applied};
  for (size_t i = 0; i < sizeof(p) / sizeof(p[0]); i++) free(*p[i]);
  free(d->geom_static); free(d->pair_g1); free(d->pair_g2);
  free(d);
}
void odata_set_state(OData* d, const double* state, double time, const double* mocap,
                     const double* userdata) {
  const mjpcx_model* m = d->m;
  memcpy(d->qpos, state, sizeof(double) * m->nq);
  memcpy(d->qvel, state + m->nq, sizeof(double) * m->nv);
  d->time = time;
  for (int i = 0; i < m->nmocap && mocap; i++) { /* trajectory.cc:121-124 */
    memcpy(d->mocap_pos + 3 * i, mocap + 7 * i, 3 * sizeof(double));
    memcpy(d->mocap_quat + 4 * i, mocap + 7 * i + 3, 4 * sizeof(double));
  }
  if (userdata && m->nuserdata) memcpy(d->userdata, userdata, sizeof(double) * m->nuserdata);
  d->warning = 0;
  memset(d->xfrc_applied, 0, sizeof(double) * 6 * m->nbody); /* likewise: the reference inherits the pooled mjData's forces */
  d->have_warm = 0; /* a rollout starts without a warm start (the reference inherits whatever the pooled mjData held) */
}
void odata_set_ctrl(OData* d, const double* ctrl) { memcpy(d->ctrl, ctrl, sizeof(double) * d->nu); }
int odata_warning(const OData* d) { return d->warning; }

/* ------------------------------------------------------------------ position stage */
static void o_kinematics(OData* d) {
  const mjpcx_model* m = d->m;
  /* world */
  d->xpos[0] = d->xpos[1] = d->xpos[2] = 0;
  d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
  quat2mat(d->xmat, d->xquat);
  memcpy(d->xipos, d->xpos, 3 * sizeof(double));
  memcpy(d->ximat, d->xmat, 9 * sizeof(double));
  for (int i = 1; i < m->nbody; i++) {
    double xpos[3], xquat[4];
    int pid = m->body_parentid[i];
    int jn = m->body_jntnum[i], ja = m->body_jntadr[i];
    if (m->body_mocapid[i] >= 0) {
      memcpy(xpos, d->mocap_pos + 3 * m->body_mocapid[i], sizeof xpos);
      memcpy(xquat, d->mocap_quat + 4 * m->body_mocapid[i], sizeof xquat);
      normalize4(xquat);
    } else if (jn == 1 && m->jnt_type[ja] == MJPCX_JNT_FREE) {
      int qa = m->jnt_qposadr[ja];
      memcpy(xpos, d->qpos + qa, sizeof xpos);
      memcpy(xquat, d->qpos + qa + 3, sizeof xquat);
      normalize4(xquat);
      memcpy(d->xanchor + 3 * ja, xpos, sizeof xpos);
      d->xaxis[3 * ja] = 0; d->xaxis[3 * ja + 1] = 0; d->xaxis[3 * ja + 2] = 1;
    } else {
      mul_mat_vec3(xpos, d->xmat + 9 * pid, m->body_pos + 3 * i);
      for (int k = 0; k < 3; k++) xpos[k] += d->xpos[3 * pid + k];
      mul_quat(xquat, d->xquat + 4 * pid, m->body_quat + 4 * i);
      for (int j = ja; j < ja + jn; j++) {
        int qa = m->jnt_qposadr[j];
        double* anchor = d->xanchor + 3 * j;
        double* axis = d->xaxis + 3 * j;
        rot_vec_quat(anchor, m->jnt_pos + 3 * j, xquat);
        for (int k = 0; k < 3; k++) anchor[k] += xpos[k];
        rot_vec_quat(axis, m->jnt_axis + 3 * j, xquat);
        switch (m->jnt_type[j]) {
          case MJPCX_JNT_SLIDE: {
            double s = d->qpos[qa] - m->qpos0[qa];
            for (int k = 0; k < 3; k++) xpos[k] += axis[k] * s;
            break;
          }
          case MJPCX_JNT_BALL:
          case MJPCX_JNT_HINGE: {
            double qloc[4], vec[3];
            if (m->jnt_type[j] == MJPCX_JNT_BALL) {
              memcpy(qloc, d->qpos + qa, sizeof qloc);
              normalize4(qloc);
            } else {
              axis_angle2quat(qloc, m->jnt_axis + 3 * j, d->qpos[qa] - m->qpos0[qa]);
            }
            mul_quat(xquat, xquat, qloc);
            rot_vec_quat(vec, m->jnt_pos + 3 * j, xquat); /* off-centre rotation */
            for (int k = 0; k < 3; k++) xpos[k] = anchor[k] - vec[k];
            break;
          }
          default: break;
        }
      }
    }
    normalize4(xquat);
    memcpy(d->xpos + 3 * i, xpos, sizeof xpos);
    memcpy(d->xquat + 4 * i, xquat, sizeof xquat);
    quat2mat(d->xmat + 9 * i, xquat);
    /* inertial frame */
    double v[3], q[4];
    mul_mat_vec3(v, d->xmat + 9 * i, m->body_ipos + 3 * i);
    for (int k = 0; k < 3; k++) d->xipos[3 * i + k] = xpos[k] + v[k];
    mul_quat(q, xquat, m->body_iquat + 4 * i);
    quat2mat(d->ximat + 9 * i, q);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_bodyid[s];
    double v[3], q[4];
    mul_mat_vec3(v, d->xmat + 9 * b, m->site_pos + 3 * s);
    for (int k = 0; k < 3; k++) d->site_xpos[3 * s + k] = d->xpos[3 * b + k] + v[k];
    mul_quat(q, d->xquat + 4 * b, m->site_quat + 4 * s);
    quat2mat(d->site_xmat + 9 * s, q);
  }
}

static void o_compos(OData* d) {
  const mjpcx_model* m = d->m;
  int nb = m->nbody;
  for (int i = 0; i < nb; i++)
    for (int k = 0; k < 3; k++) d->subtree_com[3 * i + k] = m->body_mass[i] * d->xipos[3 * i + k];
  for (int i = nb - 1; i > 0; i--)
    for (int k = 0; k < 3; k++) d->subtree_com[3 * m->body_parentid[i] + k] += d->subtree_com[3 * i + k];
  for (int i = 0; i < nb; i++) {
    if (d->subtree_mass[i] < OMINVAL) memcpy(d->subtree_com + 3 * i, d->xipos + 3 * i, 3 * sizeof(double));
    else for (int k = 0; k < 3; k++) d->subtree_com[3 * i + k] /= d->subtree_mass[i];
  }
  memset(d->cinert, 0, 10 * sizeof(double));
  for (int i = 1; i < nb; i++) {
    double off[3];
    const double* com = d->subtree_com + 3 * m->body_rootid[i];
    for (int k = 0; k < 3; k++) off[k] = d->xipos[3 * i + k] - com[k];
    inert_com(d->cinert + 10 * i, m->body_inertia + 3 * i, d->ximat + 9 * i, off, m->body_mass[i]);
  }
  for (int j = 0; j < m->njnt; j++) {
    int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j];
    double off[3];
    const double* com = d->subtree_com + 3 * m->body_rootid[b];
    for (int k = 0; k < 3; k++) off[k] = com[k] - d->xanchor[3 * j + k];
    const double* xmat = d->xmat + 9 * b;
    switch (m->jnt_type[j]) {
      case MJPCX_JNT_FREE:
        for (int k = 0; k < 3; k++) {
          double* c = d->cdof + 6 * (da + k);
          memset(c, 0, 6 * sizeof(double));
          c[3 + k] = 1;
        }
        da += 3;
        /* fallthrough: rotational dofs about body axes */
      case MJPCX_JNT_BALL:
        for (int k = 0; k < 3; k++) {
          double* c = d->cdof + 6 * (da + k);
          double ax[3] = {xmat[k], xmat[3 + k], xmat[6 + k]};
          memcpy(c, ax, sizeof ax);
          cross3(c + 3, ax, off);
        }
        break;
      case MJPCX_JNT_SLIDE: {
        double* c = d->cdof + 6 * da;
        c[0] = c[1] = c[2] = 0;
        memcpy(c + 3, d->xaxis + 3 * j, 3 * sizeof(double));
        break;
      }
      case MJPCX_JNT_HINGE: {
        double* c = d->cdof + 6 * da;
        memcpy(c, d->xaxis + 3 * j, 3 * sizeof(double));
        cross3(c + 3, d->xaxis + 3 * j, off);
        break;
      }
    }
  }
}

/* composite rigid body -> dense joint-space inertia M */
static void o_crb(OData* d) {
  const mjpcx_model* m = d->m;
  int nv = m->nv;
  memcpy(d->crb, d->cinert, 10 * m->nbody * sizeof(double));
  for (int i = m->nbody - 1; i > 0; i--)
    if (m->body_parentid[i] > 0)
      for (int k = 0; k < 10; k++) d->crb[10 * m->body_parentid[i] + k] += d->crb[10 * i + k];
  memset(d->M, 0, sizeof(double) * nv * nv);
  for (int i = 0; i < nv; i++) {
    double buf[6];
    mul_inert_vec(buf, d->crb + 10 * m->dof_bodyid[i], d->cdof + 6 * i);
    d->M[i * nv + i] = m->dof_armature[i] + dot6(d->cdof + 6 * i, buf);
    for (int j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j]) {
      double v = dot6(d->cdof + 6 * j, buf);
      d->M[i * nv + j] = v;
      d->M[j * nv + i] = v;
    }
  }
}

/* dense Cholesky A = L L'; returns 0 if not PD */
static int chol_factor(double* L, const double* A, int n) {
  memcpy(L, A, sizeof(double) * n * n);
  for (int j = 0; j < n; j++) {
    double s = L[j * n + j];
    for (int k = 0; k < j; k++) s -= L[j * n + k] * L[j * n + k];
    if (!(s > OMINVAL)) return 0;
    s = sqrt(s);
    L[j * n + j] = s;
    for (int i = j + 1; i < n; i++) {
      double t = L[i * n + j];
      for (int k = 0; k < j; k++) t -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = t / s;
    }
  }
  return 1;
}
static void chol_solve(double* x, const double* L, const double* b, int n) {
  if (x != b) memcpy(x, b, sizeof(double) * n);
  for (int i = 0; i < n; i++) {
    for (int k = 0; k < i; k++) x[i] -= L[i * n + k] * x[k];
    x[i] /= L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    for (int k = i + 1; k < n; k++) x[i] -= L[k * n + i] * x[k];
    x[i] /= L[i * n + i];
  }
}

/* solimp -> impedance at violation `dist` (pos - margin) */
static double impedance(const double* solimp, double dist) {
  double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  if (dmin < OMINIMP) dmin = OMINIMP; if (dmin > OMAXIMP) dmin = OMAXIMP;
  if (dmax < OMINIMP) dmax = OMINIMP; if (dmax > OMAXIMP) dmax = OMAXIMP;
  if (power < 1) power = 1;
  if (mid < OMINIMP) mid = OMINIMP; if (mid > OMAXIMP) mid = OMAXIMP;
  if (dmin == dmax || width <= OMINVAL) return 0.5 * (dmin + dmax);
  double x = fabs(dist) / width;
  if (x >= 1) return dmax;
  if (x <= 0) return dmin;
  double y;
  if (power == 1) y = x;
  else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
  else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
  return dmin + y * (dmax - dmin);
}

/* joint-limit constraint rows (mj_instantiateLimit + mj_makeImpedance) */
static void o_make_constraint(OData* d) {
  const mjpcx_model* m = d->m;
  int nv = m->nv;
  d->nefc = 0;
  if (m->disableflags & (MJPCX_DSBL_CONSTRAINT | MJPCX_DSBL_LIMIT)) return;
  for (int j = 0; j < m->njnt; j++) {
    if (!m->jnt_limited[j]) continue;
    if (m->jnt_type[j] != MJPCX_JNT_SLIDE && m->jnt_type[j] != MJPCX_JNT_HINGE) continue;
    double value = d->qpos[m->jnt_qposadr[j]];
    double margin = m->jnt_margin[j];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (m->jnt_range[2 * j + (side + 1) / 2] - value);
      if (dist < margin && d->nefc < OMAXEFC) {
        int r = d->nefc++;
        memset(d->efc_J + r * nv, 0, sizeof(double) * nv);
        d->efc_J[r * nv + m->jnt_dofadr[j]] = -side;
        d->efc_pos[r] = dist;
        d->efc_margin[r] = margin;
        d->efc_jnt[r] = j;
      }
    }
  }
}

/* ------------------------------------------------------------------ velocity stage */
static void o_comvel(OData* d) {
  const mjpcx_model* m = d->m;
  memset(d->cvel, 0, 6 * sizeof(double));
  for (int i = 1; i < m->nbody; i++) {
    double cvel[6];
    memcpy(cvel, d->cvel + 6 * m->body_parentid[i], sizeof cvel);
    for (int j = m->body_jntadr[i]; j < m->body_jntadr[i] + m->body_jntnum[i]; j++) {
      int da = m->jnt_dofadr[j];
      switch (m->jnt_type[j]) {
        case MJPCX_JNT_FREE:
          memset(d->cdof_dot + 6 * da, 0, 18 * sizeof(double));
          for (int k = 0; k < 3; k++)
            for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6 * (da + k) + c] * d->qvel[da + k];
          da += 3;
          /* fallthrough */
        case MJPCX_JNT_BALL:
          for (int k = 0; k < 3; k++) cross_motion(d->cdof_dot + 6 * (da + k), cvel, d->cdof + 6 * (da + k));
          for (int k = 0; k < 3; k++)
            for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6 * (da + k) + c] * d->qvel[da + k];
          break;
        default:
          cross_motion(d->cdof_dot + 6 * da, cvel, d->cdof + 6 * da);
          for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6 * da + c] * d->qvel[da];
      }
    }
    memcpy(d->cvel + 6 * i, cvel, sizeof cvel);
  }
}

static void o_passive(OData* d) {
  const mjpcx_model* m = d->m;
  memset(d->qfrc_passive, 0, sizeof(double) * m->nv);
  if (m->disableflags & MJPCX_DSBL_PASSIVE) return;
  for (int j = 0; j < m->njnt; j++) {
    double k = m->jnt_stiffness[j];
    if (k == 0) continue;
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == MJPCX_JNT_SLIDE || m->jnt_type[j] == MJPCX_JNT_HINGE)
      d->qfrc_passive[da] -= k * (d->qpos[qa] - m->qpos_spring[qa]);
    /* free/ball springs are not used by the covered models */
  }
  for (int i = 0; i < m->nv; i++) d->qfrc_passive[i] -= m->dof_damping[i] * d->qvel[i];
}

/* recursive Newton-Euler with zero acceleration: bias forces */
static void o_rne(OData* d) {
  const mjpcx_model* m = d->m;
  int nb = m->nbody;
  memset(d->cacc, 0, 6 * sizeof(double));
  if (!(m->disableflags & MJPCX_DSBL_GRAVITY))
    for (int k = 0; k < 3; k++) d->cacc[3 + k] = -m->gravity[k];
  memset(d->cfrc, 0, 6 * sizeof(double));
  for (int i = 1; i < nb; i++) {
    double* cacc = d->cacc + 6 * i;
    memcpy(cacc, d->cacc + 6 * m->body_parentid[i], 6 * sizeof(double));
    for (int k = m->body_dofadr[i]; k >= 0 && k < m->body_dofadr[i] + m->body_dofnum[i]; k++)
      for (int c = 0; c < 6; c++) cacc[c] += d->cdof_dot[6 * k + c] * d->qvel[k];
    double t1[6], t2[6], t3[6];
    mul_inert_vec(t1, d->cinert + 10 * i, cacc);
    mul_inert_vec(t2, d->cinert + 10 * i, d->cvel + 6 * i);
    cross_force(t3, d->cvel + 6 * i, t2);
    for (int c = 0; c < 6; c++) d->cfrc[6 * i + c] = t1[c] + t3[c];
  }
  for (int i = nb - 1; i > 0; i--)
    if (m->body_parentid[i] > 0)
      for (int c = 0; c < 6; c++) d->cfrc[6 * m->body_parentid[i] + c] += d->cfrc[6 * i + c];
  for (int k = 0; k < m->nv; k++) d->qfrc_bias[k] = dot6(d->cdof + 6 * k, d->cfrc + 6 * m->dof_bodyid[k]);
}

static void o_actuation(OData* d) {
  const mjpcx_model* m = d->m;
  memset(d->qfrc_actuator, 0, sizeof(double) * m->nv);
  memset(d->actuator_force, 0, sizeof(double) * m->nu);
  if (m->disableflags & MJPCX_DSBL_ACTUATION) return;
  for (int i = 0; i < m->nu; i++)
    if (is_bad(d->ctrl[i])) { /* mjWARN_BADCTRL: all controls are zeroed */
      d->warning |= 8;
      memset(d->ctrl, 0, sizeof(double) * m->nu);
      break;
    }
  for (int i = 0; i < m->nu; i++) {
    double ctrl = d->ctrl[i];
    if (m->actuator_ctrllimited[i] && !(m->disableflags & MJPCX_DSBL_CLAMPCTRL)) {
      double lo = m->actuator_ctrlrange[2 * i], hi = m->actuator_ctrlrange[2 * i + 1];
      ctrl = ctrl < lo ? lo : (ctrl > hi ? hi : ctrl);
    }
    int j = m->actuator_trnid[i];
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    double gear = m->actuator_gear[i];
    double force = m->actuator_gainprm[3 * i] * ctrl;
    if (m->actuator_biastype[i] == MJPCX_BIAS_AFFINE)
      force += m->actuator_biasprm[3 * i] + m->actuator_biasprm[3 * i + 1] * gear * d->qpos[qa] +
               m->actuator_biasprm[3 * i + 2] * gear * d->qvel[da];
    if (m->actuator_forcelimited[i]) {
      double lo = m->actuator_forcerange[2 * i], hi = m->actuator_forcerange[2 * i + 1];
      force = force < lo ? lo : (force > hi ? hi : force);
    }
    d->actuator_force[i] = force;
    d->qfrc_actuator[da] += gear * force;
  }
}

/* ------------------------------------------------------------------ acceleration stage */
/* impedance, reference acceleration and regulariser of each row, then the dual
 * problem  min_f 1/2 f'(A+R)f + f'b, f >= 0  by projected Gauss-Seidel. */
static void o_constraint(OData* d) {
  const mjpcx_model* m = d->m;
  int nv = m->nv, ne = d->nefc;
  memset(d->qfrc_constraint, 0, sizeof(double) * nv);
  memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
  if (ne == 0) return;
  double* AR = d->scratch_AR; /* (2 njnt)^2: this path only has joint-limit rows, two per joint at most */
  double* MinvJt = d->scratch_minvjt;
  for (int r = 0; r < ne; r++) {
    int j = d->efc_jnt[r];
    const double* solref = m->jnt_solref + 2 * j;
    const double* solimp = m->jnt_solimp + 5 * j;
    double pos = d->efc_pos[r] - d->efc_margin[r];
    double imp = impedance(solimp, pos);
    double dmax = solimp[1];
    if (dmax < OMINIMP) dmax = OMINIMP; if (dmax > OMAXIMP) dmax = OMAXIMP;
    double k, b;
    if (solref[0] > 0) { /* (timeconst, dampratio) */
      double tc = solref[0];
      if (!(m->disableflags & MJPCX_DSBL_REFSAFE) && tc < 2 * m->timestep) tc = 2 * m->timestep;
      k = 1.0 / (dmax * dmax * tc * tc * solref[1] * solref[1]);
      b = 2.0 / (dmax * tc);
    } else { /* direct (-stiffness, -damping) */
      k = -solref[0] / (dmax * dmax);
      b = -solref[1] / dmax;
    }
    double vel = 0;
    for (int c = 0; c < nv; c++) vel += d->efc_J[r * nv + c] * d->qvel[c];
    d->efc_aref[r] = -b * vel - k * imp * pos;
    double R = (1 - imp) / imp * m->dof_invweight0[m->jnt_dofadr[j]];
    d->efc_R[r] = R < OMINVAL ? OMINVAL : R;
    chol_solve(MinvJt + r * nv, d->L, d->efc_J + r * nv, nv);
  }
  for (int r = 0; r < ne; r++) {
    for (int s = 0; s < ne; s++) {
      double a = 0;
      for (int c = 0; c < nv; c++) a += d->efc_J[r * nv + c] * MinvJt[s * nv + c];
      AR[r * ne + s] = a;
    }
    AR[r * ne + r] += d->efc_R[r];
    double jar = 0;
    for (int c = 0; c < nv; c++) jar += d->efc_J[r * nv + c] * d->qacc_smooth[c];
    d->efc_b[r] = jar - d->efc_aref[r];
    d->efc_force[r] = 0;
  }
  double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  for (int it = 0; it < m->solver_iterations; it++) {
    double improvement = 0;
    for (int r = 0; r < ne; r++) {
      double res = d->efc_b[r];
      for (int s = 0; s < ne; s++) res += AR[r * ne + s] * d->efc_force[s];
      double old = d->efc_force[r];
      double f = old - res / AR[r * ne + r];
      if (f < 0) f = 0;
      d->efc_force[r] = f;
      double delta = f - old;
      improvement -= 0.5 * delta * delta * AR[r * ne + r] + delta * res;
    }
    if (improvement * scale < m->solver_tolerance) break;
  }
  for (int r = 0; r < ne; r++)
    for (int c = 0; c < nv; c++) {
      d->qfrc_constraint[c] += d->efc_J[r * nv + c] * d->efc_force[r];
      d->qacc[c] += MinvJt[r * nv + c] * d->efc_force[r];
    }
}

#include "contact.inc"

/* mj_xfrcAccumulate: Cartesian force / torque applied at each body's centre of mass -> generalized forces (mj_applyFT).
 * cdof is [angular, linear] about the subtree centre of mass of the root body. */
static void o_xfrc_accumulate(const OData* d, double* qfrc) {
  const mjpcx_model* m = d->m;
  for (int b = 1; b < m->nbody; b++) {
    const double* f = d->xfrc_applied + 6 * b;
    if (f[0] == 0 && f[1] == 0 && f[2] == 0 && f[3] == 0 && f[4] == 0 && f[5] == 0) continue;
    int body = b;
    while (body > 0 && m->body_dofnum[body] == 0) body = m->body_parentid[body];
    if (body <= 0) continue;
    const double* com = d->subtree_com + 3 * m->body_rootid[b];
    double off[3], tq[3];
    for (int k = 0; k < 3; k++) off[k] = d->xipos[3 * b + k] - com[k];
    cross3(tq, off, f); /* torque of the force about the reference point */
    for (int k = 0; k < 3; k++) tq[k] += f[3 + k];
    for (int i = m->body_dofadr[body] + m->body_dofnum[body] - 1; i >= 0; i = m->dof_parentid[i]) {
      const double* c = d->cdof + 6 * i;
      qfrc[i] += c[0] * tq[0] + c[1] * tq[1] + c[2] * tq[2] + c[3] * f[0] + c[4] * f[1] + c[5] * f[2];
    }
  }
}

void o_forward(OData* d) {
  const mjpcx_model* m = d->m;
  int nv = m->nv;
  o_kinematics(d);
  o_compos(d);
  o_crb(d);
  if (!chol_factor(d->L, d->M, nv)) d->warning |= 16;
  if (d->full) { o_geom_kinematics(d); o_collision(d); }
  else { if (m->ngeom) o_geom_kinematics(d); o_make_constraint(d); }
  o_comvel(d);
  if (d->full) o_make_constraint_full(d); /* needs qvel-dependent terms only through J qvel: after collision */
  o_passive(d);
  o_rne(d);
  o_actuation(d);
  for (int i = 0; i < nv; i++)
    d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
  o_xfrc_accumulate(d, d->qfrc_smooth);
  chol_solve(d->qacc_smooth, d->L, d->qfrc_smooth, nv);
  if (d->full) o_constraint_newton(d); else o_constraint(d);
}

/* mj_integratePos: qpos advanced by h along the velocity `vel` (quaternions by the exponential map) */
static void integrate_pos(const mjpcx_model* m, double* qpos, const double* vel, double h) {
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    switch (m->jnt_type[j]) {
      case MJPCX_JNT_FREE:
        for (int k = 0; k < 3; k++) qpos[qa + k] += h * vel[da + k];
        qa += 3; da += 3;
        /* fallthrough */
      case MJPCX_JNT_BALL: {
        double ax[3] = {vel[da], vel[da + 1], vel[da + 2]};
        double n = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
        if (n < OMINVAL) { ax[0] = 1; ax[1] = ax[2] = 0; }
        else { ax[0] /= n; ax[1] /= n; ax[2] /= n; }
        double qrot[4];
        axis_angle2quat(qrot, ax, h * n);
        normalize4(qpos + qa);
        mul_quat(qpos + qa, qpos + qa, qrot);
        break;
      }
      default:
        qpos[qa] += h * vel[da];
    }
  }
}

/* mj_RungeKutta(m, d, 4) (MuJoCo engine_forward.c): the classical fourth-order scheme on (qpos, qvel). Stage 0 is the mj_forward
 * mj_step has just run (with the sensor stage); stages 1..3 re-run mj_forward without sensors at the intermediate states
 * X_i = X_0 + h sum_j A_ij F_j (positions through mj_integratePos), F_j = (qvel_j, qacc_j); the step is then taken from X_0 with
 * the weights B. No implicit joint damping (an Euler-only feature). The solver's warm start stays the previous step's qacc for all
 * four stages and is replaced, in mj_advance, by the last stage's qacc. */
static void o_rk4(OData* d) {
  const mjpcx_model* m = d->m;
  const int nq = m->nq, nv = m->nv;
  const double h = m->timestep, t0 = d->time;
  static const double A[3][3] = {{0.5, 0, 0}, {0, 0.5, 0}, {0, 0, 1.0}}, B[4] = {1.0 / 6, 1.0 / 3, 1.0 / 3, 1.0 / 6}, C[3] = {0.5, 0.5, 1.0};
  double* X0 = d->rk_scratch;             /* nq + nv */
  double* F = X0 + nq + nv;               /* 4 x (nv + nv): velocity, acceleration */
  double* dX = F + 8 * nv;                /* nv + nv */
  memcpy(X0, d->qpos, sizeof(double) * nq);
  memcpy(X0 + nq, d->qvel, sizeof(double) * nv);
  memcpy(F, d->qvel, sizeof(double) * nv);
  memcpy(F + nv, d->qacc, sizeof(double) * nv);
  for (int i = 1; i < 4; i++) {
    for (int k = 0; k < 2 * nv; k++) { double s = 0; for (int j = 0; j < i; j++) s += A[i - 1][j] * F[j * 2 * nv + k]; dX[k] = s; }
    memcpy(d->qpos, X0, sizeof(double) * nq);
    integrate_pos(m, d->qpos, dX, h);
    for (int k = 0; k < nv; k++) d->qvel[k] = X0[nq + k] + h * dX[nv + k];
    d->time = t0 + C[i - 1] * h;
    o_forward(d); /* mj_forwardSkip(m, d, mjSTAGE_NONE, 1): no sensor stage */
    memcpy(F + i * 2 * nv, d->qvel, sizeof(double) * nv);
    memcpy(F + i * 2 * nv + nv, d->qacc, sizeof(double) * nv);
  }
  for (int k = 0; k < 2 * nv; k++) { double s = 0; for (int j = 0; j < 4; j++) s += B[j] * F[j * 2 * nv + k]; dX[k] = s; }
  memcpy(d->qpos, X0, sizeof(double) * nq);
  memcpy(d->qvel, X0 + nq, sizeof(double) * nv);
  d->time = t0;
  /* mj_advance(m, d, act_dot, qacc = dX acceleration, qvel = dX velocity) */
  memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv);
  d->have_warm = 1;
  for (int k = 0; k < nv; k++) d->qvel[k] += h * dX[nv + k];
  integrate_pos(m, d->qpos, dX, h);
  d->time += h;
}

/* mj_Euler with implicit joint damping, then mj_advance */
static void o_euler(OData* d) {
  const mjpcx_model* m = d->m;
  int nv = m->nv;
  double h = m->timestep;
  double* qacc = d->scratch_qacc;
  int damped = 0;
  for (int i = 0; i < nv; i++) damped |= m->dof_damping[i] > 0;
  if (damped && !(m->disableflags & MJPCX_DSBL_EULERDAMP)) {
    double* A = d->scratch_A;
    memcpy(A, d->M, sizeof(double) * nv * nv);
    for (int i = 0; i < nv; i++) A[i * nv + i] += h * m->dof_damping[i];
    for (int i = 0; i < nv; i++) qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
    if (chol_factor(A + nv * nv, A, nv)) chol_solve(qacc, A + nv * nv, qacc, nv);
    else memcpy(qacc, d->qacc, sizeof(double) * nv);
  } else {
    memcpy(qacc, d->qacc, sizeof(double) * nv);
  }
  memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv); /* mj_advance: save for the next step's solver */
  d->have_warm = 1;
  for (int i = 0; i < nv; i++) d->qvel[i] += h * qacc[i];
  integrate_pos(m, d->qpos, d->qvel, h);
  d->time += h;
}

void o_forward_task(OData* d, const mjpcx_task* task, double* r) {
  o_forward(d);
  if (task) oresidual(task, d, r);
}
/* mj_step; `task` plays the role of the mjcb_sensor callback at mjSTAGE_ACC */
void o_step_task(OData* d, const mjpcx_task* task, double* r) {
  const mjpcx_model* m = d->m;
  for (int i = 0; i < m->nq; i++) if (is_bad(d->qpos[i])) { d->warning |= 1; break; }
  for (int i = 0; i < m->nv; i++) if (is_bad(d->qvel[i])) { d->warning |= 2; break; }
  o_forward_task(d, task, r);
  for (int i = 0; i < m->nv; i++) if (is_bad(d->qacc[i])) { d->warning |= 4; break; }
  if (m->integrator == MJPCX_INT_RK4) o_rk4(d); else o_euler(d);
}
void o_step(OData* d) { o_step_task(d, NULL, NULL); }
const double* odata_site_xpos(const OData* d) { return d->site_xpos; }
double* odata_xfrc_applied(OData* d) { return d->xfrc_applied; }
/* GetTraces (utilities.cc:268-286): framepos of a site (id >= 0) or of a body frame (id = -1 - body) */
const double* odata_trace_point(const OData* d, int id) { return id >= 0 ? d->site_xpos + 3 * id : d->xpos + 3 * (-1 - id); }

/* ------------------------------------------------------------------ residuals */
/* the ResidualFn::Residual overrides of the covered tasks */
/* linear velocity of the centre of mass of the subtree rooted at `body` (sensor subtreelinvel, mj_subtreeVel) */
static void o_subtree_linvel(const OData* d, int body, double out[3]) {
  const mjpcx_model* m = d->m;
  double mom[3] = {0, 0, 0}, mass = 0;
  for (int i = body; i < m->nbody; i++) {
    int p = i, inside = 0;
    while (p > 0) { if (p == body) { inside = 1; break; } p = m->body_parentid[p]; }
    if (!inside && i != body) continue;
    const double* cv = d->cvel + 6 * i; /* [angular, linear at subtree_com[root]] */
    const double* com = d->subtree_com + 3 * m->body_rootid[i];
    double off[3], lin[3];
    for (int k = 0; k < 3; k++) off[k] = d->xipos[3 * i + k] - com[k];
    cross3(lin, cv, off);
    for (int k = 0; k < 3; k++) mom[k] += m->body_mass[i] * (cv[3 + k] + lin[k]);
    mass += m->body_mass[i];
  }
  for (int k = 0; k < 3; k++) out[k] = mass > OMINVAL ? mom[k] / mass : 0;
}
#include "quadruped.inc"
#include "humanoid.inc"

void oresidual(const mjpcx_task* task, const OData* d, double* r) {
  const mjpcx_model* m = d->m;
  switch (task->residual_id) {
    case MJPCX_RESIDUAL_QUADRUPED_FLAT:
      quadruped_residual(task, d, r);
      break;
    case MJPCX_RESIDUAL_HUMANOID_TRACK:
      humanoid_track_residual(task, d, r);
      break;
    case MJPCX_RESIDUAL_PARTICLE: /* test/testdata/particle_residual.h:33-43 */
      for (int i = 0; i < m->nq; i++) r[i] = d->qpos[i];
      r[0] -= d->mocap_pos[0];
      r[1] -= d->mocap_pos[1];
      for (int i = 0; i < m->nv; i++) r[2 + i] = d->qvel[i];
      break;
    case MJPCX_RESIDUAL_PARTICLE_COPY: /* test/agent/rollout_test.cc:37-42 */
      for (int i = 0; i < m->nq; i++) r[i] = d->qpos[i];
      for (int i = 0; i < m->nv; i++) r[m->nq + i] = d->qvel[i];
      break;
    case MJPCX_RESIDUAL_CARTPOLE: /* tasks/cartpole/cartpole.cc:36-49 */
      r[0] = cos(d->qpos[1]) - 1;
      r[1] = d->qpos[0] - task->parameters[0];
      r[2] = d->qvel[1];
      r[3] = d->ctrl[0];
      break;
    default:
      for (int i = 0; i < task->num_residual; i++) r[i] = 0;
  }
}

/* ------------------------------------------------------------------ introspection */
static int put(double* out, int cap, const double* src, int n) {
  if (n > cap) n = cap;
  memcpy(out, src, sizeof(double) * n);
  return n;
}
int odata_get(const OData* d, const char* name, double* out, int cap) {
  const mjpcx_model* m = d->m;
  int nv = m->nv, nb = m->nbody;
#define F(s, p, n) if (!strcmp(name, s)) return put(out, cap, p, n)
  F("qpos", d->qpos, m->nq); F("qvel", d->qvel, nv); F("qacc", d->qacc, nv);
  F("qacc_smooth", d->qacc_smooth, nv); F("M", d->M, nv * nv);
  F("xpos", d->xpos, 3 * nb); F("xquat", d->xquat, 4 * nb); F("xmat", d->xmat, 9 * nb);
  F("xipos", d->xipos, 3 * nb); F("site_xpos", d->site_xpos, 3 * m->nsite);
  F("subtree_com", d->subtree_com, 3 * nb); F("qfrc_bias", d->qfrc_bias, nv);
  F("qfrc_passive", d->qfrc_passive, nv); F("qfrc_actuator", d->qfrc_actuator, nv);
  F("qfrc_constraint", d->qfrc_constraint, nv); F("actuator_force", d->actuator_force, m->nu);
  F("efc_force", d->efc_force, d->nefc); F("time", &d->time, 1);
  F("cvel", d->cvel, 6 * nb); F("cdof", d->cdof, 6 * nv);
  F("geom_xpos", d->geom_xpos, 3 * m->ngeom); F("geom_xmat", d->geom_xmat, 9 * m->ngeom);
  F("efc_J", d->efc_J, d->nefc * nv); F("efc_aref", d->efc_aref, d->nefc); F("efc_R", d->efc_R, d->nefc);
  F("efc_pos", d->efc_pos, d->nefc); F("efc_floss", d->efc_floss, d->nefc);
#undef F
  if (!strcmp(name, "efc_type") || !strcmp(name, "efc_id")) { /* row kind (contact.inc EFC_*), and its joint / dof / tendon / contact index */
    int n = 0;
    for (int r = 0; r < d->nefc && n < cap; r++) out[n++] = name[4] == 't' ? d->efc_type[r] : d->efc_id[r];
    return n;
  }
  if (!strcmp(name, "contact_friction")) { /* per contact: regularised mu, friction[5] */
    int n = 0;
    for (int i = 0; i < d->ncon && n + 6 <= cap; i++) { out[n++] = d->con[i].mu; for (int k = 0; k < 5; k++) out[n++] = d->con[i].friction[k]; }
    return n;
  }
  if (!strcmp(name, "ncon")) { double v = d->ncon; return put(out, cap, &v, 1); }
  if (!strcmp(name, "solver_iter")) { double v = d->solver_iter; return put(out, cap, &v, 1); }
  if (!strcmp(name, "warning")) { double v = d->warning; return put(out, cap, &v, 1); }
  if (!strcmp(name, "contact")) { /* per contact: dist, pos[3], normal[3], geom1, geom2, dim, efc address */
    int n = 0;
    for (int i = 0; i < d->ncon && n + 11 <= cap; i++) {
      const OContact* c = d->con + i;
      out[n++] = c->dist; for (int k = 0; k < 3; k++) out[n++] = c->pos[k]; for (int k = 0; k < 3; k++) out[n++] = c->frame[k];
      out[n++] = c->g1; out[n++] = c->g2; out[n++] = c->dim; out[n++] = c->efc;
    }
    return n;
  }
  if (!strcmp(name, "subtree_linvel")) { /* per body */
    int n = 0;
    for (int i = 0; i < nb && n + 3 <= cap; i++) { o_subtree_linvel(d, i, out + n); n += 3; }
    return n;
  }
  if (!strcmp(name, "pairs")) { /* baked moving-geom pairs: geom1, geom2 */
    int n = 0;
    for (int i = 0; i < d->npair && 2 * i + 1 < cap; i++) { out[2 * i] = d->pair_g1[i]; out[2 * i + 1] = d->pair_g2[i]; n = 2 * i + 2; }
    return n;
  }
  if (!strcmp(name, "nefc")) { double v = d->nefc; return put(out, cap, &v, 1); }
  if (!strcmp(name, "energy")) { /* [potential, kinetic] (mj_energyPos/Vel) */
    double e[2] = {0, 0};
    for (int i = 1; i < nb; i++)
      for (int k = 0; k < 3; k++) e[0] -= m->body_mass[i] * m->gravity[k] * d->xipos[3 * i + k];
    for (int j = 0; j < m->njnt; j++)
      if (m->jnt_stiffness[j] > 0 && (m->jnt_type[j] == MJPCX_JNT_SLIDE || m->jnt_type[j] == MJPCX_JNT_HINGE)) {
        double dq = d->qpos[m->jnt_qposadr[j]] - m->qpos_spring[m->jnt_qposadr[j]];
        e[0] += 0.5 * m->jnt_stiffness[j] * dq * dq;
      }
    for (int i = 0; i < nv; i++)
      for (int j = 0; j < nv; j++) e[1] += 0.5 * d->qvel[i] * d->M[i * nv + j] * d->qvel[j];
    return put(out, cap, e, 2);
  }
  return -1;
}
