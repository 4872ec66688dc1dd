#include "threadpool.h"

namespace mjpc {

thread_local int ThreadPool::worker_id_ = -1;

ThreadPool::ThreadPool(int num_threads) : ctr_(0) {
  for (int i = 0; i < num_threads; i++) threads_.emplace_back(&ThreadPool::WorkerThread, this, i);
}

ThreadPool::~ThreadPool() {
  {
    std::unique_lock<std::mutex> lock(m_);
    for (size_t i = 0; i < threads_.size(); i++) queue_.push(nullptr);  // one stop token per worker
    cv_in_.notify_all();
  }
  for (auto& t : threads_) t.join();
}

void ThreadPool::Schedule(std::function<void()> task) {
  std::unique_lock<std::mutex> lock(m_);
  queue_.push(std::move(task));
  cv_in_.notify_one();
}

void ThreadPool::WorkerThread(int i) {
  worker_id_ = i;
  for (;;) {
    std::function<void()> task;
    {
      std::unique_lock<std::mutex> lock(m_);
      cv_in_.wait(lock, [&]() { return !queue_.empty(); });
      task = std::move(queue_.front());
      queue_.pop();
      cv_in_.notify_one();
    }
    const bool stop = (task == nullptr);
    if (!stop) task();
    {
      std::unique_lock<std::mutex> lock(m_);
      ++ctr_;
      cv_ext_.notify_one();
    }
    if (stop) return;
  }
}

}  // namespace mjpc
