// CPU build of the device's thin-vs-solid narrow phase (mujoco_mpc_amd/csrc/solid_pairs.h) for tests/test_solid_pairs.py and of the
// bake-time cull (pair_cull.h) -- TEST INFRASTRUCTURE ONLY.
#include "../../mujoco_mpc_amd/csrc/solid_pairs.h"
#include "../../mujoco_mpc_amd/csrc/pair_cull.h"

extern "C" double sp_thin_vs_solid(int is_cylinder, const double* size, const double* p, const double* a, double h, double r, double* n, double* c) {
  return mjpcx::solid::thin_vs_solid<double>(is_cylinder ? mjpcx::solid::kSolidCylinder : mjpcx::solid::kSolidBox, size, p, a, h, r, n, c);
}
extern "C" float sp_thin_vs_solid_f32(int is_cylinder, const float* size, const float* p, const float* a, float h, float r, float* n, float* c) {
  return mjpcx::solid::thin_vs_solid<float>(is_cylinder ? mjpcx::solid::kSolidCylinder : mjpcx::solid::kSolidBox, size, p, a, h, r, n, c);
}
// 1: the pair can never come within `margin` with every hinge between the two bodies within its range widened by `pad`; *evals: distance evaluations spent
extern "C" int sp_pair_never_touches(const mjpcx_model* m, int g1, int g2, double margin, double pad, int* evals, double* closest) {
  return mjpcx::pair_never_touches(m, g1, g2, margin, pad, evals, closest) ? 1 : 0;
}
// the moving-geom pairs of a model with their class and proofs: out[6 n] = g1, g2, kind, apart, tight_jnt, tight_side; returns n (<= cap)
extern "C" int sp_moving_pairs(const mjpcx_model* m, int* out, int cap) {
  std::vector<char> moving(m->nbody, 0);
  for (int b = 1; b < m->nbody; b++) moving[b] = moving[m->body_parentid[b]] || m->body_dofnum[b] > 0;
  std::vector<mjpcx::MovingPair> mp;
  mjpcx::moving_pairs(m, moving, true, mp);
  int n = 0;
  for (const auto& q : mp) { if (n == cap) break; int* o = out + 6 * n++; o[0] = q.g1; o[1] = q.g2; o[2] = q.kind; o[3] = q.apart; o[4] = q.tight_jnt; o[5] = q.tight_side; }
  return n;
}
