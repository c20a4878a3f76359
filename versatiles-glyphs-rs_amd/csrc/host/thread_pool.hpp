// thread_pool.hpp — persistent fork-join worker pool of the host stage.  The reference
// hands (font, block) tasks to rayon's global pool (src/font/manager.rs:117-121); here the
// same tasks are tessellated / packed / encoded by these workers around one GPU submission.
//
// A font goes through three or four fork/joins (record, merge, assemble) of 50-150 us each, so the hand-over matters
// as much as the work:
//  * a fork is ONE store: generation, item count and next item live in one 64-bit word, workers claim items with a
//    compare-exchange on it, so a worker that wakes up late — for a fork that is already over — can neither take an
//    item that does not exist (the count it checks against is the word's own) nor delay anybody;
//  * the join waits for the ITEMS, not for the workers: whoever is awake does the work, and the caller returns when
//    the last item is done (rounds 1-2 waited until every worker had checked in, i.e. for the slowest of 15-31
//    futex wake-ups, 50-90 us per fork when the workers had gone to sleep behind a device wait);
//  * every worker polls the word for ~20 us after a fork (the forks of one phase sequence — record, merge — follow each
//    other within microseconds: the merge of Noto Sans Regular takes 29 us with the workers still awake, 50-70 us when
//    all but a few have to be woken), a few (max_spinners) keep polling for ~100 us before they sleep on the condition
//    variable, the others sleep then; the caller polls the count of finished items before it sleeps.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <pthread.h>
#include <exception>
#include <functional>
#include <mutex>
#include <stdexcept>
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
			stop_.store(true, std::memory_order_release);
			state_.store(word(gen_of(state_.load(std::memory_order_relaxed)) + 1, 0, 0), std::memory_order_seq_cst);
		}
		cv_.notify_all();
		for (auto &th : threads_)
			th.join();
	}
	unsigned size() const { return n_; }

	// fn(item, worker) for item in [0, n), dynamically scheduled; returns when all are done;
	// rethrows the first exception (first error aborts, like try_for_each).
	// light: the whole fork is a few tens of microseconds of work — less than it takes a sleeping worker to wake up.  It is
	// left to the caller and the workers that are polling right now (if there are any); nobody is woken for it.
	void run(size_t n, const std::function<void(size_t, unsigned)> &fn, bool light = false)
	{
		if (n == 0)
			return;
		if (n > kMaxItems)
			throw std::runtime_error("ThreadPool::run: more than 2^24 - 1 items");
		// one fork at a time: a second run() — from an item of the running fork, or from another thread — would overwrite the
		// word, the function and the counts of the fork in flight
		struct InRun {
			std::atomic<bool> &f;
			explicit InRun(std::atomic<bool> &flag) : f(flag)
			{
				if (f.exchange(true, std::memory_order_acquire))
					throw std::logic_error("ThreadPool::run: called while a fork of this pool is running (not re-entrant)");
			}
			~InRun() { f.store(false, std::memory_order_release); }
		} in_run(in_run_);
		// (no item of the previous fork is left: its word says next == count, nobody can claim anything until the store below)
		fn_.store(&fn, std::memory_order_relaxed);
		done_.store(0, std::memory_order_relaxed);
		failed_.store(false, std::memory_order_relaxed);
		const uint32_t gen = (gen_of(state_.load(std::memory_order_relaxed)) + 1) & 0xFFFFu;
		// the fork: everything above is visible to whoever reads the word.  Sequentially consistent with the workers'
		// count of sleepers (store here, load there and the other way round): a worker on its way to sleep either sees
		// this fork or is seen here
		state_.store(word(gen, (uint32_t)n, 0), std::memory_order_seq_cst);
		if (sleepers_.load(std::memory_order_seq_cst) != 0) {
			const bool few = light && light_forks();
			// a light fork takes the workers that are polling right now; with none polling, two are woken — not all
			if (!(few && spinners_.load(std::memory_order_seq_cst) >= 2)) {
				{
					std::lock_guard<std::mutex> l(mu_); // (pairs with the sleepers' predicate check: no lost wake-up)
				}
				if (few) {
					cv_.notify_one();
					cv_.notify_one();
				} else {
					cv_.notify_all();
				}
			}
		}
		work(0, gen);
		const uint32_t total = (uint32_t)n;
		if (!poll([&] { return done_.load(std::memory_order_acquire) == total; })) {
			std::unique_lock<std::mutex> l(done_mu_);
			done_cv_.wait(l, [&] { return done_.load(std::memory_order_acquire) == total; });
		}
		if (failed_.load(std::memory_order_acquire)) {
			std::lock_guard<std::mutex> l(err_mu_);
			throw std::runtime_error(error_);
		}
	}

private:
	// busy-wait for `ready` for at most ~100 us (VG_POOL_SPIN_US: measurement switch); true when it came
	static long spin_us()
	{
		static const long us = [] {
			const char *e = std::getenv("VG_POOL_SPIN_US");
			return e ? std::max(0l, std::atol(e)) : 100l;
		}();
		return us;
	}
	static bool light_forks()
	{
		static const bool on = [] {
			const char *e = std::getenv("VG_POOL_LIGHT"); // (measurement switch)
			return e && e[0] == '1';
		}();
		return on;
	}
	static long short_us()
	{
		static const long us = [] {
			const char *e = std::getenv("VG_POOL_SPIN_ALL_US"); // (measurement switch)
			return e ? std::max(0l, std::atol(e)) : 20l;
		}();
		return us;
	}
	template <class Ready> static bool poll(Ready ready) { return poll_for(ready, spin_us()); }
	template <class Ready> static bool poll_for(Ready ready, long us)
	{
		const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(us);
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
	// items of fork `gen`, one compare-exchange each, until there is none left or the word has moved on
	void work(unsigned id, uint32_t gen)
	{
		for (;;) {
			uint64_t v = state_.load(std::memory_order_acquire);
			if (gen_of(v) != gen)
				return;
			const uint32_t i = (uint32_t)(v & kMaxItems), total = (uint32_t)((v >> 24) & kMaxItems);
			if (i >= total)
				return;
			// a run of items per claim (1/8 of a worker's share): thousands of tiny items — a block file each, most of them
			// empty — otherwise spend their time in failed compare-exchanges on this one word (5376 items on 32 threads:
			// 0.45 ms of "encoding" that was contention)
			const uint32_t take = std::min(total - i, std::max(1u, total / (n_ * 8u)));
			if (!state_.compare_exchange_weak(v, v + take, std::memory_order_acq_rel, std::memory_order_acquire))
				continue;
			if (!failed_.load(std::memory_order_relaxed)) {
				try {
					const auto &fn = *fn_.load(std::memory_order_relaxed);
					for (uint32_t k = i; k < i + take; k++)
						fn(k, id);
				} catch (const std::exception &e) {
					std::lock_guard<std::mutex> l(err_mu_);
					if (!failed_.exchange(true))
						error_ = e.what();
				} catch (...) { // (anything else would end the process from a pool thread: std::terminate)
					std::lock_guard<std::mutex> l(err_mu_);
					if (!failed_.exchange(true))
						error_ = "unknown exception in a pool item";
				}
			}
			if (done_.fetch_add(take, std::memory_order_acq_rel) + take == total) { // the last items: the caller may be asleep
				{
					std::lock_guard<std::mutex> l(done_mu_); // (pairs with the caller's wait: no lost wake-up)
				}
				done_cv_.notify_one();
			}
		}
	}
	// workers that may poll at the same time; the others go to sleep as soon as they run out of items.  Two or three hot
	// workers (and the caller) pick the next fork up at once — enough for the short phases — while the rest costs no CPU
	// time between forks: under a CPU quota every spinning thread is paid for (font_manager.cpp, worker_count)
	unsigned max_spinners() const
	{
		static const long env = [] {
			const char *e = std::getenv("VG_POOL_SPINNERS"); // (measurement switch)
			return e ? std::max(0l, std::atol(e)) : -1l;
		}();
		return env >= 0 ? (unsigned)env : std::max(2u, n_ / 12u);
	}
	void loop(unsigned id)
	{
		{
			char name[16];
			std::snprintf(name, sizeof name, "vg-pool-%u", id);
			pthread_setname_np(pthread_self(), name);
		}
		uint32_t seen = 0;
		for (;;) {
			auto moved = [&] { return gen_of(state_.load(std::memory_order_acquire)) != seen; };
			// every worker polls briefly (forks of one phase sequence follow each other within microseconds: record ->
			// merge), a few keep polling for the next sequence
			bool came = short_us() ? poll_for(moved, short_us()) : false;
			if (!came) {
				if (spinners_.fetch_add(1, std::memory_order_relaxed) < max_spinners())
					came = poll(moved);
				spinners_.fetch_sub(1, std::memory_order_relaxed);
			}
			if (!came) {
				std::unique_lock<std::mutex> l(mu_);
				sleepers_.fetch_add(1, std::memory_order_seq_cst);
				cv_.wait(l, [&] { return gen_of(state_.load(std::memory_order_seq_cst)) != seen; });
				sleepers_.fetch_sub(1, std::memory_order_seq_cst);
			}
			if (stop_.load(std::memory_order_acquire))
				return;
			seen = gen_of(state_.load(std::memory_order_acquire));
			work(id, seen);
		}
	}

	unsigned n_;
	std::vector<std::thread> threads_;
	std::mutex mu_, done_mu_, err_mu_;
	std::condition_variable cv_, done_cv_;
	static constexpr uint64_t kMaxItems = 0xFFFFFFull;
	static uint64_t word(uint32_t gen, uint32_t total, uint32_t next) { return ((uint64_t)gen << 48) | ((uint64_t)total << 24) | next; }
	static uint32_t gen_of(uint64_t v) { return (uint32_t)(v >> 48); }
	std::atomic<const std::function<void(size_t, unsigned)> *> fn_{nullptr}; // published by the store to state_
	std::atomic<uint64_t> state_{0}; // generation (16 bits) | item count (24) | next item (24)
	std::atomic<uint32_t> done_{0};
	std::atomic<bool> failed_{false}, stop_{false}, in_run_{false};
	std::string error_;
	std::atomic<unsigned> sleepers_{0}, spinners_{0};
};

} // namespace vg
