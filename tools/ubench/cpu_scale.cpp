// cpu_scale.cpp — how many cores does this box really give a process?  N threads each spin through the same fixed amount of integer
// work; perfect scaling keeps the wall time flat until N exceeds the cores available.   g++ -O2 -pthread cpu_scale.cpp -o cpu_scale
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <thread>
#include <vector>
static uint64_t work(uint64_t seed) { uint64_t x = seed; for (uint64_t i = 0; i < 300000000ull; i++) x = x * 6364136223846793005ull + 1442695040888963407ull; return x; }
int main() {
  std::printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
  volatile uint64_t sink = 0;
  for (unsigned n : {1u, 8u, 32u, 64u, 128u, 256u}) {
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> ts;
    std::vector<uint64_t> r(n);
    for (unsigned i = 0; i < n; i++) ts.emplace_back([&, i] { r[i] = work(i + 1); });
    for (auto& t : ts) t.join();
    for (auto v : r) sink += v;
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("%3u threads: %.3f s  (cores' worth of work per second: %.1f)\n", n, s, (double)n / s);
  }
  return 0;
}
