// Golden-vector generator for tools/dataset/fgnn_dataset cache-by-random: what the standard-library calls of the
// reference's generator (utility/data-process/toolkit/cache/cache_by_random.cc:38-42: default-seeded std::mt19937,
// std::uniform_int_distribution<uint32_t>(0, n-i-1), swap with the tail) produce with this image's libstdc++.
// g++ -O2 -o /tmp/fy tests/golden/std_fisher_yates_probe.cc && /tmp/fy 1000 > tests/golden/cache_by_random_1000.txt
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <utility>
#include <vector>

int main(int argc, char **argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 1000;
  std::vector<uint32_t> r(n);
  for (uint32_t i = 0; i < n; i++) r[i] = i;
  std::mt19937 generator;
  for (uint32_t i = 0; i < n; i++) {
    std::uniform_int_distribution<uint32_t> distribution(0, n - i - 1);
    std::swap(r[n - i - 1], r[distribution(generator)]);
  }
  for (uint32_t v : r) printf("%u\n", v);
  return 0;
}
