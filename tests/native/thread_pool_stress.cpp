// Stress test of the host pool (csrc/host/thread_pool.hpp): every item of every fork runs exactly once, also when workers
// wake up late for a fork that is over (round 3 found exactly that race with ThreadSanitizer: a late worker claimed an
// item of the NEXT fork against the previous fork's word), the first exception is rethrown, the pool survives it.
#ifndef ROUNDS
#define ROUNDS 4000
#endif
#include "thread_pool.hpp"
#include <cstdio>
#include <numeric>
int main() {
	vg::ThreadPool tp(8);
	std::vector<long> acc(8, 0);
	long want = 0;
	for (int r = 0; r < ROUNDS; r++) {
		const size_t n = 1 + (r * 7919) % 300;
		std::vector<int> hit(n, 0);
		tp.run(n, [&](size_t i, unsigned w) { hit[i]++; acc[w] += (long)i; }, /*light=*/r % 3 == 1); // (VG_POOL_LIGHT=1: those wake two workers at most)
		for (size_t i = 0; i < n; i++) { if (hit[i] != 1) { std::printf("item %zu run %d hit %d\n", i, r, hit[i]); return 1; } want += (long)i; }
		if (r % 100 == 0) std::this_thread::sleep_for(std::chrono::microseconds(300)); // let the workers fall asleep
	}
	long got = std::accumulate(acc.begin(), acc.end(), 0L);
	bool threw = false;
	try { tp.run(100, [&](size_t i, unsigned) { if (i == 37) throw std::runtime_error("boom"); }); } catch (const std::exception &e) { threw = std::string(e.what()) == "boom"; }
	tp.run(10, [&](size_t, unsigned) {});
	// an exception that is not a std::exception is reported, not std::terminate on a pool thread (round-3 advice)
	bool threw_other = false;
	try { tp.run(100, [&](size_t i, unsigned) { if (i == 5) throw 42; }); } catch (const std::exception &e) { threw_other = std::string(e.what()).find("unknown") != std::string::npos; }
	// run() from an item of a running fork is refused instead of overwriting the fork in flight
	bool nested = false;
	try { tp.run(4, [&](size_t, unsigned) { tp.run(2, [](size_t, unsigned) {}); }); } catch (const std::exception &e) { nested = std::string(e.what()).find("re-entrant") != std::string::npos; }
	tp.run(10, [&](size_t, unsigned) {});
	threw = threw && threw_other && nested;
	std::printf("%s sum %ld want %ld threw %d\n", got == want ? "OK" : "BAD", got, want, threw);
	return got == want && threw ? 0 : 1;
}
