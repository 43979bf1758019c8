// gbp_threads.hpp — the few host loops of the library that run on several threads (the file readers and the prior strengths of gbp_host.cpp,
// the per-position gather of gbp_upload): how many threads, and how they are started.  Results never depend on the count.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

namespace gbp {
namespace host {

// at most one thread per `min_items_per_thread` items, at most 32, at most the hardware's (GBP_HOST_THREADS=n overrides that)
inline unsigned host_threads(uint64_t work_items, uint64_t min_items_per_thread) {
  unsigned T = std::thread::hardware_concurrency();
  if (const char* e = std::getenv("GBP_HOST_THREADS")) T = (unsigned)std::max(1, std::atoi(e));
  T = std::max(1u, std::min(T, 32u));
  return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(T, work_items / std::max<uint64_t>(1, min_items_per_thread)));
}
// fn(t) for t in [0, T), fn(0) on the caller's thread.  fn must not throw (it allocates nothing: an exception inside a thread would end the
// process); a thread the system refuses to start has its share run on the caller's thread instead.
template <class F> void on_threads(unsigned T, F&& fn) {
  std::vector<std::thread> th;
  th.reserve(T);
  unsigned started = 1;
  try {
    for (; started < T; ++started) th.emplace_back(fn, started);
  } catch (...) {
  }
  fn(0u);
  for (unsigned t = started; t < T; ++t) fn(t);
  for (auto& x : th) x.join();
}


}  // namespace host
}  // namespace gbp
