// thread_pool.hpp — persistent fork-join worker pool of the host stage.  The reference
// hands (font, block) tasks to rayon's global pool (src/font/manager.rs:117-121); here the
// same tasks are tessellated / packed / encoded by these workers around one GPU submission.
//
// A font goes through three fork/joins (record, merge, encode) of ~0.1 ms each, so the hand-over matters as much
// as the work: workers poll the generation counter for a short while after finishing (the next fork usually
// follows within microseconds) before they sleep on the condition variable, and the caller polls the count of
// busy workers before it sleeps (a futex wake-up of 15 sleeping threads costs 20-50 us per fork).
#pragma once
#include <atomic>
#include <chrono>
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
			gen_.fetch_add(1, std::memory_order_release);
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
			pending_.store(n_ - 1, std::memory_order_relaxed);
			gen_.fetch_add(1, std::memory_order_release);
		}
		if (sleepers_.load(std::memory_order_acquire) != 0)
			cv_.notify_all();
		work(0);
		if (!poll([this] { return pending_.load(std::memory_order_acquire) == 0; })) {
			std::unique_lock<std::mutex> l(mu_);
			done_cv_.wait(l, [this] { return pending_.load(std::memory_order_acquire) == 0; });
		}
		fn_ = nullptr;
		if (failed_.load())
			throw std::runtime_error(error_);
	}

private:
	// busy-wait for `ready` for at most ~100 us; true when it came
	template <class Ready> static bool poll(Ready ready)
	{
		const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(100);
		for (;;) {
			for (int i = 0; i < 64; i++) {
				if (ready())
					return true;
#if defined(__x86_64__) || defined(__i386__)
				__builtin_ia32_pause();
#endif
			}
			if (std::chrono::steady_clock::now() >= until)
				return ready();
		}
	}
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
			if (!poll([&] { return gen_.load(std::memory_order_acquire) != seen; })) {
				std::unique_lock<std::mutex> l(mu_);
				sleepers_.fetch_add(1, std::memory_order_release);
				cv_.wait(l, [&] { return gen_.load(std::memory_order_acquire) != seen; });
				sleepers_.fetch_sub(1, std::memory_order_release);
			}
			{
				// (the fields of the fork are published under the mutex: take it once before reading them)
				std::lock_guard<std::mutex> l(mu_);
				seen = gen_.load(std::memory_order_acquire);
				if (stop_)
					return;
			}
			work(id);
			if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
				std::lock_guard<std::mutex> l(mu_); // (pairs with the caller's wait: no lost wake-up)
				done_cv_.notify_one();
			}
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
	std::atomic<unsigned> pending_{0}, sleepers_{0};
	std::atomic<uint64_t> gen_{0};
	bool stop_ = false;
};

} // namespace vg
