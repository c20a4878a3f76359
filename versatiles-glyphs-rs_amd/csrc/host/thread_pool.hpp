// thread_pool.hpp — persistent fork-join worker pool of the host stage.  The reference
// hands (font, block) tasks to rayon's global pool (src/font/manager.rs:117-121); here the
// same tasks are tessellated / packed / encoded by these workers around one GPU submission.
#pragma once
#include <atomic>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace vg {

class ThreadPool {
public:
	explicit ThreadPool(unsigned workers) : n_(workers ? workers : 1)
	{
		for (unsigned t = 1; t < n_; t++) // the caller is worker 0
			threads_.emplace_back([this, t] { loop(t); });
	}
	~ThreadPool()
	{
		{
			std::lock_guard<std::mutex> l(mu_);
			stop_ = true;
			gen_++;
		}
		cv_.notify_all();
		for (auto &th : threads_)
			th.join();
	}
	unsigned size() const { return n_; }

	// fn(item, worker) for item in [0, n), dynamically scheduled; returns when all are done;
	// rethrows the first exception (first error aborts, like try_for_each).
	void run(size_t n, const std::function<void(size_t, unsigned)> &fn)
	{
		if (n == 0)
			return;
		{
			std::lock_guard<std::mutex> l(mu_);
			fn_ = &fn;
			total_ = n;
			next_.store(0);
			failed_.store(false);
			error_.clear();
			pending_ = n_ - 1;
			gen_++;
		}
		cv_.notify_all();
		work(0);
		std::unique_lock<std::mutex> l(mu_);
		done_cv_.wait(l, [this] { return pending_ == 0; });
		fn_ = nullptr;
		if (failed_.load())
			throw std::runtime_error(error_);
	}

private:
	void work(unsigned id)
	{
		for (;;) {
			const size_t i = next_.fetch_add(1);
			if (i >= total_ || failed_.load())
				return;
			try {
				(*fn_)(i, id);
			} catch (const std::exception &e) {
				std::lock_guard<std::mutex> l(err_mu_);
				if (!failed_.exchange(true))
					error_ = e.what();
			}
		}
	}
	void loop(unsigned id)
	{
		uint64_t seen = 0;
		for (;;) {
			{
				std::unique_lock<std::mutex> l(mu_);
				cv_.wait(l, [&] { return gen_ != seen; });
				seen = gen_;
				if (stop_)
					return;
			}
			work(id);
			std::lock_guard<std::mutex> l(mu_);
			if (--pending_ == 0)
				done_cv_.notify_one();
		}
	}

	unsigned n_;
	std::vector<std::thread> threads_;
	std::mutex mu_, err_mu_;
	std::condition_variable cv_, done_cv_;
	const std::function<void(size_t, unsigned)> *fn_ = nullptr;
	size_t total_ = 0;
	std::atomic<size_t> next_{0};
	std::atomic<bool> failed_{false};
	std::string error_;
	unsigned pending_ = 0;
	uint64_t gen_ = 0;
	bool stop_ = false;
};

} // namespace vg
