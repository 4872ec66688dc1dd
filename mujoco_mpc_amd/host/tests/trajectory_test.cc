// mjpc/test/agent/trajectory_test.cc: Reset zeroes every buffer; copy-assignment is deep.
#include "mjpc/trajectory.h"

#include "check.h"
using namespace mjpc;

int main() {
  Trajectory t;
  const int T = 10;
  t.Initialize(2, 1, 3, 1, T);
  t.Allocate(T);
  for (auto* v : {&t.states, &t.actions, &t.times, &t.residual, &t.costs, &t.trace})
    for (double& x : *v) x = 1.0;
  t.total_return = 5.0;
  t.failure = true;
  CHECK(t.states.size() == 20 && t.actions.size() == 10 && t.residual.size() == 30 && t.trace.size() == 30);
  Trajectory copy = t;
  t.Reset(T);
  for (auto* v : {&t.states, &t.actions, &t.times, &t.residual, &t.costs, &t.trace})
    for (double x : *v) CHECK(x == 0.0);
  CHECK(t.total_return == 0.0 && !t.failure);
  CHECK(copy.states[3] == 1.0 && copy.total_return == 5.0 && copy.failure);
  const double a0[1] = {0.25};
  t.Reset(T, a0);
  for (double x : t.actions) CHECK(x == 0.25);
  TEST_MAIN_END();
}
