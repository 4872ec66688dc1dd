// lds_model.h -- the model of a REGISTERED wavefront-per-candidate model staged into LDS with a compile-time layout.
//
// The generic kernels (rollout_wave.h) read the model through ~100 pointers held in the kernel-argument segment: every
// lane-indexed read is a global load (L2 hit ~200 cycles on a dependent chain), the pointers cost ~200 SGPRs (spilled
// and reloaded lane by lane) and the sizes are run-time values, so no LDS address folds into an instruction's offset
// field. For a registered model (dimensions known when the library is built, `generated/static_models.h`) the hot arrays
// are laid out in ONE image whose offsets are constexpr functions of the dimensions; the rollout kernel copies the image
// to the start of its workgroup's LDS once and every device function -- templated on the model type -- then reads
// `m.body_pos[3 * i + k]` as a ds_read with an immediate offset. Cold arrays (contact parameters read once per contact,
// the pair list, key_mpos) stay behind global pointers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <vector>

#include "wave_model.h"

namespace mjpcx {

// the workgroup's dynamic LDS: [model image | per-plan blob | one candidate arena per wavefront]
extern __shared__ __attribute__((aligned(16))) unsigned char mjpcx_lds[];

// X(type tag, name, count): I = int32, R = real (the working type), U = uint32, L = uint64
#define MJPCX_LDS_MODEL_FIELDS(X)                                                                                              \
  X(I, body_parentid, C::NB) X(I, body_rootid, C::NB) X(I, body_jntnum, C::NB) X(I, body_jntadr, C::NB) X(I, body_dofnum, C::NB) \
  X(I, body_dofadr, C::NB) X(I, body_mocapid, C::NB)                                                                            \
  X(R, body_pos, 3 * C::NB) X(R, body_quat, 4 * C::NB) X(R, body_ipos, 3 * C::NB) X(R, body_iquat, 4 * C::NB) X(R, body_mass, C::NB) \
  X(R, body_inertia, 3 * C::NB) X(R, body_invweight0, 2 * C::NB) X(R, body_subtreemass, C::NB)                                  \
  X(I, jnt_type, C::NJ) X(I, jnt_qposadr, C::NJ) X(I, jnt_dofadr, C::NJ) X(I, jnt_bodyid, C::NJ) X(I, jnt_limited, C::NJ)        \
  X(R, jnt_pos, 3 * C::NJ) X(R, jnt_axis, 3 * C::NJ) X(R, jnt_stiffness, C::NJ) X(R, jnt_range, 2 * C::NJ) X(R, jnt_margin, C::NJ) \
  X(R, jnt_solref, 2 * C::NJ) X(R, jnt_solimp, 5 * C::NJ)                                                                       \
  X(I, dof_bodyid, C::NV) X(I, dof_jntid, C::NV) X(I, dof_parentid, C::NV)                                                      \
  X(R, dof_armature, C::NV) X(R, dof_damping, C::NV) X(R, dof_frictionloss, C::NV) X(R, dof_invweight0, C::NV)                  \
  X(R, dof_solref, 2 * C::NV) X(R, dof_solimp, 5 * C::NV)                                                                       \
  X(R, qpos0, C::NQ) X(R, qpos_spring, C::NQ)                                                                                   \
  X(I, site_bodyid, C::NS) X(R, site_pos, 3 * C::NS)                                                                            \
  X(I, actuator_trnid, C::NU) X(I, actuator_biastype, C::NU) X(I, actuator_ctrllimited, C::NU) X(I, actuator_forcelimited, C::NU) \
  X(R, actuator_gear, C::NU) X(R, actuator_gainprm, 3 * C::NU) X(R, actuator_biasprm, 3 * C::NU) X(R, actuator_ctrlrange, 2 * C::NU) \
  X(R, actuator_forcerange, 2 * C::NU)                                                                                          \
  X(I, geom_type, C::NG) X(I, geom_bodyid, C::NG) X(I, geom_contype, C::NG) X(I, geom_conaffinity, C::NG) X(I, geom_condim, C::NG) \
  X(I, geom_priority, C::NG)                                                                                                    \
  X(R, geom_size, 3 * C::NG) X(R, geom_pos, 3 * C::NG) X(R, geom_quat, 4 * C::NG) X(R, geom_margin, C::NG)                       \
  X(R, key_qpos, C::NKEY * C::NQ)                                                                                               \
  X(L, body_subtree_mask, C::NB) X(U, body_dofmask, C::NB)                                                                       \
  X(I, level_body, C::NB) X(I, level_start, kWaveMaxLevel + 1) X(I, static_geom, C::NSG) X(I, dynamic_geom, C::NDG) X(I, ray_geom, C::NRAY) \
  X(I, task_dim_norm_residual, C::NTERM) X(I, task_norm, C::NTERM) X(I, task_trace_site, C::NTRACE) X(I, task_term_off, C::NTERM) \
  X(I, task_res_term, C::NR)

enum LdsFieldTag { kLdsI = 0, kLdsR = 1, kLdsU = 2, kLdsL = 3 };
enum LdsField {
#define X(tag, name, count) kLdsF_##name,
  MJPCX_LDS_MODEL_FIELDS(X)
#undef X
  kLdsFieldCount
};

template <class C, typename T>
struct LdsLayout {
  static constexpr unsigned elem_bytes(int tag) { return tag == kLdsR ? sizeof(T) : tag == kLdsL ? 8u : 4u; }
  static constexpr unsigned bytes(int f) {
    switch (f) {
#define X(tag, name, count) case kLdsF_##name: return elem_bytes(kLds##tag) * (unsigned)((count) > 0 ? (count) : 1);
      MJPCX_LDS_MODEL_FIELDS(X)
#undef X
      default: return 0;
    }
  }
  static constexpr unsigned offset(int f) {
    unsigned o = 0;
    for (int i = 0; i < f; i++) o = (o + bytes(i) + 15u) & ~15u;
    return o;
  }
  static constexpr unsigned kBytes = offset(kLdsFieldCount);
};

// read-only view of one array of the image (the image sits at the start of the workgroup's LDS)
template <typename E, unsigned OFF>
struct LdsArr {
  __device__ __forceinline__ const E* ptr() const { return reinterpret_cast<const E*>(mjpcx_lds + OFF); }
  __device__ __forceinline__ E operator[](int i) const { return ptr()[i]; }
  __device__ __forceinline__ const E* operator+(int k) const { return ptr() + k; }
  __device__ __forceinline__ operator const E*() const { return ptr(); }
};

template <typename T, typename E> struct LdsElem { typedef E type; };
template <int TAG, typename T> struct LdsTagType;
template <typename T> struct LdsTagType<kLdsI, T> { typedef int type; };
template <typename T> struct LdsTagType<kLdsR, T> { typedef T type; };
template <typename T> struct LdsTagType<kLdsU, T> { typedef unsigned type; };
template <typename T> struct LdsTagType<kLdsL, T> { typedef unsigned long long type; };

// The device-side model of a registered configuration C: compile-time sizes, hot arrays in LDS, the rest copied from the
// generic WaveModelT (scalars by value, cold arrays as global pointers).
template <class C, typename T>
struct LdsModelT {
  typedef LdsLayout<C, T> Layout;
  static constexpr int nq = C::NQ, nv = C::NV, nu = C::NU, nbody = C::NB, njnt = C::NJ, nsite = C::NS, ngeom = C::NG, nkey = C::NKEY;
  static constexpr int nmocap = C::NMOCAP, nbody_model = C::NBM, ntendon = C::NT;
  static constexpr int nstatic_geom = C::NSG, ndynamic_geom = C::NDG, nray_geom = C::NRAY;
#define X(tag, name, count) LdsArr<typename LdsTagType<kLds##tag, T>::type, Layout::offset(kLdsF_##name)> name;
  MJPCX_LDS_MODEL_FIELDS(X)
#undef X
  // run-time scalars
  int cone, disableflags, solver_iterations, any_damping, nlevel, npair, full;
  double timestep, gravity[3], solver_tolerance, meaninertia, impratio;
  // (level_start is an array of the image: as a member it would be indexed at run time, and an aggregate indexed at run time lives in
  // scratch -- the whole struct with it, every scalar and pointer below a scratch load)
  // cold arrays (global memory)
  const T *geom_friction, *geom_solref, *geom_solimp, *geom_gap, *geom_solmix, *key_mpos;
  const int *pair_g1, *pair_g2;
  // the (two or three) limited fixed tendons of a model stay behind global pointers: a handful of reads per step
  const int *tendon_adr, *tendon_num, *tendon_limited, *wrap_objid;
  const T *wrap_prm, *tendon_range, *tendon_margin, *tendon_solref_lim, *tendon_solimp_lim, *tendon_invweight0;
  const unsigned* tendon_dofmask;

  __device__ __forceinline__ explicit LdsModelT(const WaveModelT<T>& m)
      : cone(m.cone), disableflags(m.disableflags), solver_iterations(m.solver_iterations), any_damping(m.any_damping), nlevel(m.nlevel),
        npair(m.npair), full(m.full), timestep(m.timestep), solver_tolerance(m.solver_tolerance), meaninertia(m.meaninertia), impratio(m.impratio),
        geom_friction(m.geom_friction), geom_solref(m.geom_solref), geom_solimp(m.geom_solimp), geom_gap(m.geom_gap), geom_solmix(m.geom_solmix),
        key_mpos(m.key_mpos), pair_g1(m.pair_g1), pair_g2(m.pair_g2), tendon_adr(m.tendon_adr), tendon_num(m.tendon_num), tendon_limited(m.tendon_limited),
        wrap_objid(m.wrap_objid), wrap_prm(m.wrap_prm), tendon_range(m.tendon_range), tendon_margin(m.tendon_margin), tendon_solref_lim(m.tendon_solref_lim),
        tendon_solimp_lim(m.tendon_solimp_lim), tendon_invweight0(m.tendon_invweight0), tendon_dofmask(m.tendon_dofmask) {
    for (int k = 0; k < 3; k++) gravity[k] = m.gravity[k];
  }
};

// static part of the task (the residual's term partition) through the same image; the per-plan blob follows the image
template <class C, typename T>
struct LdsTaskT {
  typedef LdsLayout<C, T> Layout;
  static constexpr int nr = C::NR, nterm = C::NTERM, ntrace = C::NTRACE;
  int residual_id, nparam, nri, nrr;
  LdsArr<int, Layout::offset(kLdsF_task_dim_norm_residual)> dim_norm_residual;
  LdsArr<int, Layout::offset(kLdsF_task_norm)> norm;
  LdsArr<int, Layout::offset(kLdsF_task_trace_site)> trace_site;
  LdsArr<int, Layout::offset(kLdsF_task_term_off)> term_off;
  LdsArr<int, Layout::offset(kLdsF_task_res_term)> res_term;
  const T* blob;  // LDS copy of the per-plan blob
  long long* stamps;
  int stamp_step;
  int off_time, off_mocap, off_weight, off_normp, off_normq, off_param, off_risk, off_rreal, off_rint;
  __device__ __forceinline__ LdsTaskT(const WaveTaskT<T>& t, const T* lds_blob)
      : residual_id(t.residual_id), nparam(t.nparam), nri(t.nri), nrr(t.nrr), blob(lds_blob), stamps(t.stamps), stamp_step(t.stamp_step),
        off_time(t.off_time), off_mocap(t.off_mocap), off_weight(t.off_weight), off_normp(t.off_normp), off_normq(t.off_normq),
        off_param(t.off_param), off_risk(t.off_risk), off_rreal(t.off_rreal), off_rint(t.off_rint) {}
};

// ---------------------------------------------------------------- host side: the image of a registered model
template <class C, typename T>
inline std::vector<unsigned char> lds_model_image(const mjpcx_model* src, const mjpcx_task* task, const WaveHost& wh) {
  typedef LdsLayout<C, T> L;
  std::vector<unsigned char> img(L::kBytes, 0);
  auto put_i = [&](int f, const int* p, size_t n) { if (p && n) std::memcpy(img.data() + L::offset(f), p, n * 4); };
  auto put_r = [&](int f, const double* p, size_t n) {
    if (!p) return;
    T* d = reinterpret_cast<T*>(img.data() + L::offset(f));
    for (size_t i = 0; i < n; i++) d[i] = (T)p[i];
  };
  // fields copied straight from the mjpcx_model
#define COPY_I(name, n) put_i(kLdsF_##name, src->name, (size_t)(n))
#define COPY_R(name, n) put_r(kLdsF_##name, src->name, (size_t)(n))
  COPY_I(body_parentid, C::NB); COPY_I(body_rootid, C::NB); COPY_I(body_jntnum, C::NB); COPY_I(body_jntadr, C::NB); COPY_I(body_dofnum, C::NB);
  COPY_I(body_dofadr, C::NB); COPY_I(body_mocapid, C::NB);
  COPY_R(body_pos, 3 * C::NB); COPY_R(body_quat, 4 * C::NB); COPY_R(body_ipos, 3 * C::NB); COPY_R(body_iquat, 4 * C::NB); COPY_R(body_mass, C::NB);
  COPY_R(body_inertia, 3 * C::NB); COPY_R(body_invweight0, 2 * C::NB); COPY_R(body_subtreemass, C::NB);
  COPY_I(jnt_type, C::NJ); COPY_I(jnt_qposadr, C::NJ); COPY_I(jnt_dofadr, C::NJ); COPY_I(jnt_bodyid, C::NJ); COPY_I(jnt_limited, C::NJ);
  COPY_R(jnt_pos, 3 * C::NJ); COPY_R(jnt_axis, 3 * C::NJ); COPY_R(jnt_stiffness, C::NJ); COPY_R(jnt_range, 2 * C::NJ); COPY_R(jnt_margin, C::NJ);
  COPY_R(jnt_solref, 2 * C::NJ); COPY_R(jnt_solimp, 5 * C::NJ);
  COPY_I(dof_bodyid, C::NV); COPY_I(dof_jntid, C::NV); COPY_I(dof_parentid, C::NV);
  COPY_R(dof_armature, C::NV); COPY_R(dof_damping, C::NV); COPY_R(dof_frictionloss, C::NV); COPY_R(dof_invweight0, C::NV);
  COPY_R(dof_solref, 2 * C::NV); COPY_R(dof_solimp, 5 * C::NV);
  COPY_R(qpos0, C::NQ); COPY_R(qpos_spring, C::NQ);
  COPY_I(site_bodyid, C::NS); COPY_R(site_pos, 3 * C::NS);
  COPY_I(actuator_trnid, C::NU); COPY_I(actuator_biastype, C::NU); COPY_I(actuator_ctrllimited, C::NU); COPY_I(actuator_forcelimited, C::NU);
  COPY_R(actuator_gear, C::NU); COPY_R(actuator_gainprm, 3 * C::NU); COPY_R(actuator_biasprm, 3 * C::NU); COPY_R(actuator_ctrlrange, 2 * C::NU);
  COPY_R(actuator_forcerange, 2 * C::NU);
  COPY_I(geom_type, C::NG); COPY_I(geom_bodyid, C::NG); COPY_I(geom_contype, C::NG); COPY_I(geom_conaffinity, C::NG); COPY_I(geom_condim, C::NG);
  COPY_I(geom_priority, C::NG);
  COPY_R(geom_size, 3 * C::NG); COPY_R(geom_pos, 3 * C::NG); COPY_R(geom_quat, 4 * C::NG); COPY_R(geom_margin, C::NG);
  COPY_R(key_qpos, (size_t)C::NKEY * C::NQ);
#undef COPY_I
#undef COPY_R
  // baked helpers (WaveHost::build keeps host copies)
  std::memcpy(img.data() + L::offset(kLdsF_body_subtree_mask), wh.h_subtree_mask.data(), wh.h_subtree_mask.size() * 8);
  std::memcpy(img.data() + L::offset(kLdsF_body_dofmask), wh.h_dofmask.data(), wh.h_dofmask.size() * 4);
  put_i(kLdsF_level_body, wh.h_level_body.data(), wh.h_level_body.size());
  put_i(kLdsF_level_start, wh.m.level_start, (size_t)kWaveMaxLevel + 1);
  put_i(kLdsF_static_geom, wh.h_static_geom.data(), wh.h_static_geom.size());
  put_i(kLdsF_dynamic_geom, wh.h_dynamic_geom.data(), wh.h_dynamic_geom.size());
  put_i(kLdsF_ray_geom, wh.h_ray_geom.data(), wh.h_ray_geom.size());
  put_i(kLdsF_task_dim_norm_residual, task->dim_norm_residual, (size_t)C::NTERM);
  put_i(kLdsF_task_norm, task->norm, (size_t)C::NTERM);
  put_i(kLdsF_task_trace_site, task->trace_site, (size_t)C::NTRACE);
  put_i(kLdsF_task_term_off, wh.h_term_off.data(), wh.h_term_off.size());
  put_i(kLdsF_task_res_term, wh.h_res_term.data(), wh.h_res_term.size());
  return img;
}

// does the run-time model have exactly the registered dimensions?
template <class C>
inline bool lds_model_matches(const mjpcx_model* m, const mjpcx_task* t, const WaveHost& wh) {
  return m->nq == C::NQ && m->nv == C::NV && m->nu == C::NU && m->nbody == C::NBM && m->njnt == C::NJ && m->nsite >= C::NS && m->ngeom == C::NG &&
         (C::NKEY == 0 || m->nkey == C::NKEY) && m->nmocap == C::NMOCAP && wh.m.nbody == C::NB && wh.m.nsite == C::NS && (int)wh.h_static_geom.size() == C::NSG &&
         (int)wh.h_dynamic_geom.size() == C::NDG && (int)wh.h_ray_geom.size() == C::NRAY && t->num_residual == C::NR && t->num_term == C::NTERM &&
         t->num_trace == C::NTRACE && m->ntendon == C::NT &&
         m->integrator == MJPCX_INT_EULER;  // (RK4 runs in the generic kernels: wave_kernel.h)
}

}  // namespace mjpcx
