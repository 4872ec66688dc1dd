// mjpc::ThreadPool -- the public surface of the reference's pool (mjpc/threadpool.h:40-60: Schedule, NumThreads, WorkerId,
// GetCount / ResetCount / WaitCount) kept so that Planner::OptimizePolicy(int, ThreadPool&) compiles unchanged. The GPU
// planners schedule no rollouts on it; it serves host-side helpers and the interface tests. Written from scratch: a
// deque of jobs guarded by one mutex, an explicit shutdown flag (workers drain the deque, then leave), a completed-job
// counter with its own condition variable.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace mjpc {

class ThreadPool {
 public:
  explicit ThreadPool(int num_threads);
  ThreadPool(const ThreadPool&) = delete;
  ThreadPool& operator=(const ThreadPool&) = delete;
  ~ThreadPool();

  int NumThreads() const { return static_cast<int>(workers_.size()); }
  // index of the calling worker (0 .. NumThreads() - 1), -1 on any other thread
  static int WorkerId() { return tls_worker_; }
  void Schedule(std::function<void()> job);
  // number of jobs finished since construction / the last ResetCount
  std::uint64_t GetCount();
  void ResetCount();
  void WaitCount(int value);  // blocks until GetCount() >= value

 private:
  void Run(int index);
  static thread_local int tls_worker_;
  std::vector<std::thread> workers_;
  std::mutex lock_;                    // guards jobs_, shutdown_ and finished_
  std::condition_variable have_work_;  // signalled on Schedule and on shutdown
  std::condition_variable progress_;   // signalled whenever a job finishes
  std::deque<std::function<void()>> jobs_;
  bool shutdown_ = false;
  std::uint64_t finished_ = 0;
};

}  // namespace mjpc
