// mjpc::ThreadPool (mjpc/threadpool.h). The GPU planners accept it to keep the Planner interface
// (OptimizePolicy(int, ThreadPool&)) but schedule no rollouts on it.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <queue>
#include <thread>
#include <vector>

namespace mjpc {

class ThreadPool {
 public:
  explicit ThreadPool(int num_threads);
  ~ThreadPool();
  int NumThreads() const { return (int)threads_.size(); }
  static int WorkerId() { return worker_id_; }  // 0..NumThreads()-1 inside a worker, -1 elsewhere
  void Schedule(std::function<void()> task);
  std::uint64_t GetCount() { return ctr_; }
  void ResetCount() { ctr_ = 0; }
  void WaitCount(int value) {
    std::unique_lock<std::mutex> lock(m_);
    cv_ext_.wait(lock, [&]() { return (int)this->GetCount() >= value; });
  }

 private:
  void WorkerThread(int i);
  static thread_local int worker_id_;
  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable cv_in_, cv_ext_;
  std::queue<std::function<void()>> queue_;
  std::uint64_t ctr_;
};

}  // namespace mjpc
