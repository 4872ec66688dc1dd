// The host classes the reference shares between its physics thread and its planner thread: State (writer:
// Agent's SetState from the simulation; readers: Planner::SetState -> CopyTo) and ThreadPool (Schedule / WaitCount / ResetCount from the
// planning thread while workers run). Built with -fsanitize=thread by `make tsan`; run by tests/test_host_cpp.py. argv[1] = Particle.mjpx
#include <atomic>
#include <thread>
#include <vector>

#include "check.h"
#include "mjpc/states/state.h"
#include "mjpc/threadpool.h"
#include "model_io.h"
using namespace mjpc;

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  auto storage = ModelStorage::Load(argv[1]);
  const mjModel* m = storage->model();
  // ---- State: one writer, three readers; a reader must never see a torn snapshot (every field carries the writer's counter)
  State s;
  s.Allocate(m);
  s.Reset();
  std::atomic<bool> stop{false};
  std::atomic<int> torn{0}, reads{0};
  std::thread writer([&] {
    for (int i = 1; i <= 20000; i++) {
      const double v = i;
      const double qpos[2] = {v, v}, qvel[2] = {v, v}, mpos[3] = {v, v, v}, mquat[4] = {1, 0, 0, 0};
      s.Set(m, qpos, qvel, nullptr, mpos, mquat, nullptr, v);
      if (i % 5 == 0) s.SetTime(m, v);
    }
    stop = true;
  });
  std::vector<std::thread> readers;
  for (int r = 0; r < 3; r++)
    readers.emplace_back([&] {
      double st[4], mc[7], ud[1], t;
      while (!stop) {
        s.CopyTo(st, mc, ud, &t);
        if (!(st[0] == st[1] && st[1] == st[2] && st[2] == st[3] && mc[0] == st[0] && mc[2] == st[0] && t == st[0])) torn++;
        reads++;
      }
    });
  writer.join();
  for (auto& t : readers) t.join();
  CHECK(torn == 0 && reads > 0);
  // ---- ThreadPool: rounds of Schedule + WaitCount + ResetCount, tasks touching shared and per-worker data
  ThreadPool pool(4);
  std::vector<long> per_worker(pool.NumThreads(), 0);
  std::atomic<long> total{0};
  for (int round = 0; round < 200; round++) {
    const int count_before = (int)pool.GetCount();
    for (int k = 0; k < 16; k++)
      pool.Schedule([&, k] {
        per_worker[ThreadPool::WorkerId()] += k;   // (a worker's own slot: no other thread writes it)
        total += k;
      });
    pool.WaitCount(count_before + 16);
    pool.ResetCount();
  }
  long sum = 0;
  for (long v : per_worker) sum += v;
  CHECK(sum == total && total == 200L * 120);
  TEST_MAIN_END();
}
