#include "context.h"

namespace mjpc::gpu {

FlatModel::FlatModel(const mjModel* m, double timestep, int integrator, bool differentiable) {
  mjpcx_model& f = flat_;
  f.nq = m->nq; f.nv = m->nv; f.nu = m->nu; f.na = m->na; f.nbody = m->nbody; f.njnt = m->njnt;
  f.nsite = m->nsite; f.nmocap = m->nmocap; f.nuserdata = m->nuserdata;
  f.timestep = timestep;
  for (int k = 0; k < 3; k++) f.gravity[k] = m->opt.gravity[k];
  f.integrator = integrator;
  f.disableflags = m->opt.disableflags;
  f.solver_iterations = m->opt.iterations;
  f.solver_tolerance = m->opt.tolerance;
  f.meaninertia = m->stat.meaninertia;
  // arrays that already have the ABI's element type and stride are passed through
  f.body_parentid = m->body_parentid; f.body_rootid = m->body_rootid; f.body_jntnum = m->body_jntnum;
  f.body_jntadr = m->body_jntadr; f.body_dofnum = m->body_dofnum; f.body_dofadr = m->body_dofadr;
  f.body_mocapid = m->body_mocapid;
  f.body_pos = m->body_pos; f.body_quat = m->body_quat; f.body_ipos = m->body_ipos; f.body_iquat = m->body_iquat;
  f.body_mass = m->body_mass; f.body_inertia = m->body_inertia;
  f.jnt_type = m->jnt_type; f.jnt_qposadr = m->jnt_qposadr; f.jnt_dofadr = m->jnt_dofadr; f.jnt_bodyid = m->jnt_bodyid;
  f.jnt_pos = m->jnt_pos; f.jnt_axis = m->jnt_axis; f.jnt_stiffness = m->jnt_stiffness; f.jnt_range = m->jnt_range;
  f.jnt_margin = m->jnt_margin; f.jnt_solref = m->jnt_solref; f.jnt_solimp = m->jnt_solimp;
  f.dof_bodyid = m->dof_bodyid; f.dof_jntid = m->dof_jntid; f.dof_parentid = m->dof_parentid;
  f.dof_armature = m->dof_armature; f.dof_damping = m->dof_damping; f.dof_frictionloss = m->dof_frictionloss;
  f.dof_invweight0 = m->dof_invweight0;
  f.qpos0 = m->qpos0; f.qpos_spring = m->qpos_spring;
  f.site_bodyid = m->site_bodyid; f.site_pos = m->site_pos; f.site_quat = m->site_quat;
  f.actuator_gaintype = m->actuator_gaintype; f.actuator_biastype = m->actuator_biastype;
  f.actuator_ctrlrange = m->actuator_ctrlrange; f.actuator_forcerange = m->actuator_forcerange;
  // MuJoCo strides / byte flags -> the ABI's compact int32 / [nu x 3] layout
  jnt_limited_.assign(m->jnt_limited, m->jnt_limited + m->njnt);
  trnid_.resize(m->nu); ctrllimited_.resize(m->nu); forcelimited_.resize(m->nu);
  gear_.resize(m->nu); gainprm_.resize(3 * (size_t)m->nu); biasprm_.resize(3 * (size_t)m->nu);
  for (int u = 0; u < m->nu; u++) {
    trnid_[u] = m->actuator_trnid[2 * u];
    ctrllimited_[u] = m->actuator_ctrllimited[u];
    forcelimited_[u] = m->actuator_forcelimited[u];
    gear_[u] = m->actuator_gear[6 * u];
    for (int k = 0; k < 3; k++) {
      gainprm_[3 * u + k] = m->actuator_gainprm[mjNGAIN * u + k];
      biasprm_[3 * u + k] = m->actuator_biasprm[mjNBIAS * u + k];
    }
  }
  f.jnt_limited = jnt_limited_.data();
  f.actuator_trnid = trnid_.data(); f.actuator_ctrllimited = ctrllimited_.data();
  f.actuator_forcelimited = forcelimited_.data(); f.actuator_gear = gear_.data();
  f.actuator_gainprm = gainprm_.data(); f.actuator_biasprm = biasprm_.data();
  // contacts and constraint options (geom_* keep MuJoCo's strides: size 3, friction 3, solref mjNREF, solimp mjNIMP)
  f.ngeom = m->ngeom; f.nkey = m->nkey; f.cone = m->opt.cone; f.impratio = m->opt.impratio;
  f.geom_type = m->geom_type; f.geom_bodyid = m->geom_bodyid; f.geom_contype = m->geom_contype;
  f.geom_conaffinity = m->geom_conaffinity; f.geom_condim = m->geom_condim; f.geom_priority = m->geom_priority;
  f.geom_group = m->geom_group; f.geom_size = m->geom_size; f.geom_pos = m->geom_pos; f.geom_quat = m->geom_quat;
  f.geom_friction = m->geom_friction; f.geom_solref = m->geom_solref; f.geom_solimp = m->geom_solimp;
  f.geom_margin = m->geom_margin; f.geom_gap = m->geom_gap; f.geom_solmix = m->geom_solmix;
  f.body_invweight0 = m->body_invweight0; f.body_subtreemass = m->body_subtreemass;
  f.dof_solref = m->dof_solref; f.dof_solimp = m->dof_solimp; f.key_qpos = m->key_qpos;
  // fixed tendons (mjWRAP_JOINT entries only), contact excludes, weld ids, mocap keyframes
  f.ntendon = m->ntendon; f.nwrap = m->nwrap; f.nexclude = m->nexclude;
  for (int w = 0; w < m->nwrap; w++)
    if (m->wrap_type[w] != mjWRAP_JOINT) throw Error(MJPCX_EUNSUPPORTED, "spatial tendons are not supported by the device path");
  tendon_limited_.assign(m->tendon_limited, m->tendon_limited + m->ntendon);
  if (tendon_limited_.empty()) tendon_limited_.resize(1);
  f.tendon_adr = m->tendon_adr; f.tendon_num = m->tendon_num; f.tendon_limited = tendon_limited_.data();
  f.wrap_objid = m->wrap_objid; f.wrap_prm = m->wrap_prm; f.tendon_range = m->tendon_range; f.tendon_margin = m->tendon_margin;
  f.tendon_solref_lim = m->tendon_solref_lim; f.tendon_solimp_lim = m->tendon_solimp_lim;
  f.tendon_invweight0 = m->tendon_invweight0;
  f.exclude_signature = m->exclude_signature; f.body_weldid = m->body_weldid; f.key_mpos = m->key_mpos;
  if (differentiable) {
    jnt_solimp_.assign(m->jnt_solimp, m->jnt_solimp + (size_t)mjNIMP * m->njnt);
    geom_solimp_.assign(m->geom_solimp, m->geom_solimp + (size_t)mjNIMP * m->ngeom);
    for (int i = 0; i < m->njnt; i++) jnt_solimp_[(size_t)mjNIMP * i] = 0.0;
    for (int i = 0; i < m->ngeom; i++) geom_solimp_[(size_t)mjNIMP * i] = 0.0;
    f.jnt_solimp = jnt_solimp_.data();
    if (m->ngeom) f.geom_solimp = geom_solimp_.data();
  }
}

FlatTask::FlatTask(const Task& t) {
  norm_.assign(t.norm.begin(), t.norm.end());
  flat_.residual_id = t.DeviceResidualId();
  flat_.num_residual = t.num_residual;
  flat_.num_term = t.num_term;
  flat_.num_trace = t.num_trace;
  flat_.num_parameter = (int)t.parameters.size();
  flat_.dim_norm_residual = t.dim_norm_residual.data();
  flat_.norm = norm_.data();
  flat_.num_norm_parameter = t.num_norm_parameter.data();
  flat_.weight = t.weight.data();
  flat_.norm_parameter = t.norm_parameter.data();
  parameters_ = t.NumericParameters();
  flat_.parameters = parameters_.data();
  flat_.trace_site = t.trace_site.data();
  flat_.risk = t.risk;
  t.ResidualState(&residual_int_, &residual_real_);
  flat_.num_residual_int = (int)residual_int_.size();
  flat_.num_residual_real = (int)residual_real_.size();
  flat_.residual_int = residual_int_.data();
  flat_.residual_real = residual_real_.data();
}

Context::Context(const mjModel* model, const Task& task, int device, int precision, bool differentiable) {
  // the planning copy runs at agent_timestep / agent_integrator when the model defines them
  double timestep = model->opt.timestep;
  int integrator = model->opt.integrator;
  for (int i = 0; i < model->nnumeric; i++) {
    const std::string name = model->names + model->name_numericadr[i];
    if (name == "agent_timestep") timestep = model->numeric_data[model->numeric_adr[i]];
    if (name == "agent_integrator") integrator = (int)model->numeric_data[model->numeric_adr[i]];
  }
  FlatModel fm(model, timestep, integrator, differentiable);
  FlatTask ft(task);
  const int rc = mjpcx_create(fm.get(), ft.get(), device, precision, &ctx_);
  if (rc != MJPCX_OK) throw Error(rc, std::string(mjpcx_error_string(rc)) + ": " + mjpcx_create_error());
}

Context::~Context() { mjpcx_destroy(ctx_); }

void Context::Check(int rc) const {
  if (rc != MJPCX_OK) throw Error(rc, std::string(mjpcx_error_string(rc)) + ": " + mjpcx_last_error(ctx_));
}

// the per-plan frozen task copy (Agent::PlanIteration: residual_fn_ = task->Residual(), agent.cc:319)
void Context::SyncTask(const Task& t) {
  const std::vector<double> parameters = t.NumericParameters();
  Check(mjpcx_set_task_params(ctx_, t.weight.data(), t.norm_parameter.data(), parameters.data(), t.risk));
  std::vector<int32_t> ri;
  std::vector<double> rr;
  t.ResidualState(&ri, &rr);
  if (!ri.empty() || !rr.empty()) Check(mjpcx_set_residual_state(ctx_, ri.empty() ? nullptr : ri.data(), rr.empty() ? nullptr : rr.data()));
}

void KinematicsBuffers::Allocate(const mjModel* m) {
  xpos.assign(3 * (size_t)m->nbody, 0.0); xquat.assign(4 * (size_t)m->nbody, 0.0); xmat.assign(9 * (size_t)m->nbody, 0.0);
  xipos.assign(3 * (size_t)m->nbody, 0.0); site_xpos.assign(3 * (size_t)m->nsite, 0.0);
  subtree_com.assign(3 * (size_t)m->nbody, 0.0); subtree_linvel.assign(3 * (size_t)m->nbody, 0.0);
}

void KinematicsBuffers::Attach(mjData* d) {
  d->xpos = xpos.data(); d->xquat = xquat.data(); d->xmat = xmat.data(); d->xipos = xipos.data(); d->site_xpos = site_xpos.data();
  d->subtree_com = subtree_com.data(); d->subtree_linvel = subtree_linvel.data();
}

bool Context::Kinematics(KinematicsBuffers* out) {
  const int rc = mjpcx_kinematics(ctx_, out->xpos.data(), out->xquat.data(), out->xmat.data(), out->xipos.data(),
                                  out->site_xpos.data(), out->subtree_com.data(), out->subtree_linvel.data());
  if (rc == MJPCX_EUNSUPPORTED) return false;
  Check(rc);
  return true;
}

void Context::FetchTrajectory(int index, Trajectory* tr) {
  mjpcx_traj_view v{};
  v.horizon = (int)tr->times.size();
  v.states = tr->states.data(); v.actions = tr->actions.data(); v.times = tr->times.data();
  v.residual = tr->residual.data(); v.costs = tr->costs.data(); v.trace = tr->trace.data();
  Check(mjpcx_fetch_trajectory(ctx_, index, &v));
  tr->horizon = v.horizon;
  tr->total_return = v.total_return;
  tr->failure = v.failure != 0;
}

}  // namespace mjpc::gpu
