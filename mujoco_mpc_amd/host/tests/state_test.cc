// mjpc/test/state/state_test.cc: Set / CopyTo round trip (argv[1] = Particle.mjpx).
#include "mjpc/states/state.h"

#include "check.h"
#include "model_io.h"
using namespace mjpc;

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  auto storage = ModelStorage::Load(argv[1]);
  const mjModel* m = storage->model();
  State s;
  s.Allocate(m);
  s.Reset();
  CHECK(s.state().size() == 4 && s.mocap().size() == 7 && s.time() == 0.0);
  double qpos[2] = {0.1, 0.2}, qvel[2] = {0.3, 0.4}, mpos[3] = {1, 2, 3}, mquat[4] = {1, 0, 0, 0};
  s.Set(m, qpos, qvel, nullptr, mpos, mquat, nullptr, 1.5);
  double st[4], mc[7], ud[1], t;
  s.CopyTo(st, mc, ud, &t);
  CHECK(st[0] == 0.1 && st[1] == 0.2 && st[2] == 0.3 && st[3] == 0.4 && t == 1.5);
  CHECK(mc[0] == 1 && mc[2] == 3 && mc[3] == 1 && mc[6] == 0);
  double q2[2], v2[2], mp2[3], mq2[4];
  mjData d{};
  d.qpos = q2; d.qvel = v2; d.mocap_pos = mp2; d.mocap_quat = mq2;
  s.CopyTo(m, &d);
  CHECK(d.qpos[1] == 0.2 && d.qvel[0] == 0.3 && d.mocap_pos[1] == 2 && d.time == 1.5);
  State s2;
  s2.Set(m, &d);
  CHECK(s2.state() == s.state() && s2.mocap() == s.mocap());
  TEST_MAIN_END();
}
