#include "model_io.h"

#include <cstdio>
#include <stdexcept>

namespace mjpc {
namespace {
template <typename T>
void ReadRaw(FILE* f, T* dst, size_t n) {
  if (n && std::fread(dst, sizeof(T), n, f) != n) throw std::runtime_error("model blob: truncated");
}
}  // namespace

std::unique_ptr<ModelStorage> ModelStorage::Load(const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("model blob: cannot open " + path);
  auto s = std::unique_ptr<ModelStorage>(new ModelStorage());
  try {
    char magic[10];
    ReadRaw(f, magic, 10);
    if (std::string(magic, 10) != "MJPXBLOB1\n") throw std::runtime_error("model blob: bad magic");
    uint32_t count = 0;
    ReadRaw(f, &count, 1);
    for (uint32_t e = 0; e < count; e++) {
      uint32_t len = 0;
      ReadRaw(f, &len, 1);
      std::string name(len, '\0');
      ReadRaw(f, name.data(), len);
      uint8_t kind = 0;
      uint64_t n = 0;
      ReadRaw(f, &kind, 1);
      ReadRaw(f, &n, 1);
      if (kind == 0) { auto& v = s->ints_[name]; v.resize(n); ReadRaw(f, v.data(), n); }
      else if (kind == 1) { auto& v = s->reals_[name]; v.resize(n); ReadRaw(f, v.data(), n); }
      else if (kind == 2) { auto& v = s->bytes_[name]; v.resize(n); ReadRaw(f, v.data(), n); }
      else throw std::runtime_error("model blob: unknown entry kind");
    }
  } catch (...) {
    std::fclose(f);
    throw;
  }
  std::fclose(f);
  s->Bind();
  return s;
}

int* ModelStorage::I(const std::string& name, size_t n) {
  auto& v = ints_[name];
  if (v.size() < n) throw std::runtime_error("model blob: field " + name + " too short");
  if (v.empty()) v.resize(1);
  return v.data();
}
double* ModelStorage::R(const std::string& name, size_t n) {
  auto& v = reals_[name];
  if (v.size() < n) throw std::runtime_error("model blob: field " + name + " too short");
  if (v.empty()) v.resize(1);
  return v.data();
}
unsigned char* ModelStorage::B(const std::string& name, size_t n) {
  auto& v = bytes_[name];
  if (v.size() < n) throw std::runtime_error("model blob: field " + name + " too short");
  if (v.empty()) v.resize(1);
  return v.data();
}

void ModelStorage::Bind() {
  mjModel& m = model_;
  const int* sz = I("sizes", 14);
  m.nq = sz[0]; m.nv = sz[1]; m.nu = sz[2]; m.na = sz[3]; m.nbody = sz[4]; m.njnt = sz[5]; m.nsite = sz[6];
  m.nmocap = sz[7]; m.nuserdata = sz[8]; m.nsensor = sz[9]; m.nuser_sensor = sz[10]; m.nnumeric = sz[11];
  m.ntext = sz[12]; m.nkey = sz[13];
  const double* opt = R("opt", 6);
  m.opt.timestep = opt[0]; m.opt.gravity[0] = opt[1]; m.opt.gravity[1] = opt[2]; m.opt.gravity[2] = opt[3];
  m.opt.tolerance = opt[4]; m.stat.meaninertia = opt[5];
  const int* oi = I("opt_int", 5);
  m.opt.integrator = oi[0]; m.opt.iterations = oi[1]; m.opt.disableflags = oi[2]; m.opt.cone = oi[3]; m.ngeom = oi[4];
  m.opt.impratio = R("opt_impratio", 1)[0];
  const size_t nb = m.nbody, nj = m.njnt, nv = m.nv, nq = m.nq, ns = m.nsite, nu = m.nu;
#define BI(f, n) m.f = I(#f, n)
#define BR(f, n) m.f = R(#f, n)
#define BB(f, n) m.f = B(#f, n)
  BI(body_parentid, nb); BI(body_rootid, nb); BI(body_jntnum, nb); BI(body_jntadr, nb); BI(body_dofnum, nb);
  BI(body_dofadr, nb); BI(body_mocapid, nb);
  BR(body_pos, 3 * nb); BR(body_quat, 4 * nb); BR(body_ipos, 3 * nb); BR(body_iquat, 4 * nb); BR(body_mass, nb);
  BR(body_inertia, 3 * nb);
  BI(jnt_type, nj); BI(jnt_qposadr, nj); BI(jnt_dofadr, nj); BI(jnt_bodyid, nj); BB(jnt_limited, nj);
  BR(jnt_pos, 3 * nj); BR(jnt_axis, 3 * nj); BR(jnt_stiffness, nj); BR(jnt_range, 2 * nj); BR(jnt_margin, nj);
  BR(jnt_solref, mjNREF * nj); BR(jnt_solimp, mjNIMP * nj);
  BI(dof_bodyid, nv); BI(dof_jntid, nv); BI(dof_parentid, nv);
  BR(dof_armature, nv); BR(dof_damping, nv); BR(dof_frictionloss, nv); BR(dof_invweight0, nv);
  BR(qpos0, nq); BR(qpos_spring, nq);
  BI(site_bodyid, ns); BR(site_pos, 3 * ns); BR(site_quat, 4 * ns);
  BI(actuator_trnid, 2 * nu); BI(actuator_gaintype, nu); BI(actuator_biastype, nu);
  BB(actuator_ctrllimited, nu); BB(actuator_forcelimited, nu);
  BR(actuator_gear, 6 * nu); BR(actuator_gainprm, mjNGAIN * nu); BR(actuator_biasprm, mjNBIAS * nu);
  BR(actuator_ctrlrange, 2 * nu); BR(actuator_forcerange, 2 * nu);
  const size_t nsn = m.nsensor;
  BI(sensor_type, nsn); BI(sensor_objtype, nsn); BI(sensor_objid, nsn); BI(sensor_dim, nsn); BI(sensor_adr, nsn);
  BR(sensor_user, nsn * (size_t)m.nuser_sensor);
  BI(numeric_adr, m.nnumeric); BI(numeric_size, m.nnumeric); BR(numeric_data, 0);
  BR(key_qpos, (size_t)m.nkey * nq); BR(key_qvel, (size_t)m.nkey * nv);
  BI(name_bodyadr, nb); BI(name_jntadr, nj); BI(name_siteadr, ns); BI(name_sensoradr, nsn);
  BI(name_numericadr, m.nnumeric); BI(name_keyadr, m.nkey);
  const size_t ng = m.ngeom;
  BR(body_invweight0, 2 * nb); BR(body_subtreemass, nb); BR(dof_solref, mjNREF * nv); BR(dof_solimp, mjNIMP * nv);
  BI(geom_type, ng); BI(geom_bodyid, ng); BI(geom_contype, ng); BI(geom_conaffinity, ng); BI(geom_condim, ng);
  BI(geom_priority, ng); BI(geom_group, ng);
  BR(geom_size, 3 * ng); BR(geom_pos, 3 * ng); BR(geom_quat, 4 * ng); BR(geom_friction, 3 * ng); BR(geom_solref, mjNREF * ng);
  BR(geom_solimp, mjNIMP * ng); BR(geom_margin, ng); BR(geom_gap, ng); BR(geom_solmix, ng);
  BI(name_geomadr, ng);
  const int* ts = I("tendon_sizes", 3);
  m.ntendon = ts[0]; m.nwrap = ts[1]; m.nexclude = ts[2];
  const size_t nt = m.ntendon, nw = m.nwrap;
  BI(tendon_adr, nt); BI(tendon_num, nt); BI(wrap_objid, nw); BI(exclude_signature, m.nexclude); BI(body_weldid, nb);
  BR(wrap_prm, nw); BR(tendon_range, 2 * nt); BR(tendon_margin, nt); BR(tendon_solref_lim, mjNREF * nt);
  BR(tendon_solimp_lim, mjNIMP * nt); BR(tendon_invweight0, nt);
  {  // byte flags / wrap types as mjModel stores them (the blob carries int32 limited flags, joint wraps only)
    const int* lim = I("tendon_limited", nt);
    auto& lb = bytes_["tendon_limited"]; lb.assign(nt ? nt : 1, 0);
    for (size_t i = 0; i < nt; i++) lb[i] = (unsigned char)lim[i];
    m.tendon_limited = lb.data();
    auto& wt = ints_["wrap_type"]; wt.assign(nw ? nw : 1, mjWRAP_JOINT);
    m.wrap_type = wt.data();
  }
  BR(key_mpos, (size_t)m.nkey * 3 * m.nmocap); BR(key_ctrl, (size_t)m.nkey * nu);
  BI(text_adr, m.ntext); BI(text_size, m.ntext); BI(name_textadr, m.ntext);
  m.text_data = reinterpret_cast<char*>(B("text_data", 0));
  m.names = reinterpret_cast<char*>(B("names", 1));
#undef BI
#undef BR
#undef BB
}

int NameToId(const mjModel* m, int objtype, const std::string& name) {
  const int* adr = nullptr;
  int n = 0;
  if (objtype == mjOBJ_BODY) { adr = m->name_bodyadr; n = m->nbody; }
  else if (objtype == mjOBJ_XBODY) { adr = m->name_bodyadr; n = m->nbody; }
  else if (objtype == mjOBJ_SITE) { adr = m->name_siteadr; n = m->nsite; }
  else if (objtype == mjOBJ_GEOM) { adr = m->name_geomadr; n = m->ngeom; }
  else if (objtype == mjOBJ_KEY) { adr = m->name_keyadr; n = m->nkey; }
  for (int i = 0; i < n; i++)
    if (name == m->names + adr[i]) return i;
  return -1;
}

}  // namespace mjpc
