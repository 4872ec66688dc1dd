// mjpc/test/agent/threadpool_test.cc
#include "mjpc/threadpool.h"

#include <atomic>

#include "check.h"
using namespace mjpc;

int main() {
  ThreadPool pool(2);
  CHECK(pool.NumThreads() == 2 && ThreadPool::WorkerId() == -1);
  std::atomic<int> sum{0};
  std::atomic<int> bad_ids{0};
  const int count_before = (int)pool.GetCount();
  for (int i = 0; i < 5; i++)
    pool.Schedule([&sum, &bad_ids, i]() {
      sum += i;
      const int id = ThreadPool::WorkerId();
      if (id < 0 || id > 1) bad_ids++;
    });
  pool.WaitCount(count_before + 5);
  CHECK(sum == 10 && bad_ids == 0 && pool.GetCount() == 5u);
  pool.ResetCount();
  CHECK(pool.GetCount() == 0u);
  TEST_MAIN_END();
}
