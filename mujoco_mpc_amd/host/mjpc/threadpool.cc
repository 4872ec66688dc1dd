#include "threadpool.h"

#include <utility>

namespace mjpc {

thread_local int ThreadPool::tls_worker_ = -1;

ThreadPool::ThreadPool(int num_threads) {
  const int n = num_threads > 0 ? num_threads : 1;
  workers_.reserve(n);
  for (int index = 0; index < n; ++index) workers_.emplace_back([this, index]() { Run(index); });
}

ThreadPool::~ThreadPool() {
  {
    std::lock_guard<std::mutex> hold(lock_);
    shutdown_ = true;
  }
  have_work_.notify_all();
  for (std::thread& w : workers_) w.join();
}

void ThreadPool::Schedule(std::function<void()> job) {
  {
    std::lock_guard<std::mutex> hold(lock_);
    jobs_.push_back(std::move(job));
  }
  have_work_.notify_one();
}

std::uint64_t ThreadPool::GetCount() {
  std::lock_guard<std::mutex> hold(lock_);
  return finished_;
}

void ThreadPool::ResetCount() {
  std::lock_guard<std::mutex> hold(lock_);
  finished_ = 0;
}

void ThreadPool::WaitCount(int value) {
  std::unique_lock<std::mutex> hold(lock_);
  progress_.wait(hold, [this, value]() { return finished_ >= static_cast<std::uint64_t>(value > 0 ? value : 0); });
}

void ThreadPool::Run(int index) {
  tls_worker_ = index;
  std::unique_lock<std::mutex> hold(lock_);
  while (true) {
    have_work_.wait(hold, [this]() { return shutdown_ || !jobs_.empty(); });
    if (jobs_.empty()) return;  // shutdown with nothing left to drain
    std::function<void()> job = std::move(jobs_.front());
    jobs_.pop_front();
    hold.unlock();
    if (job) job();
    hold.lock();
    ++finished_;
    progress_.notify_all();
  }
}

}  // namespace mjpc
