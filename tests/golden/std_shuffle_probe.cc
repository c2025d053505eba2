// Probe used ONLY by make_golden.py: prints what libstdc++'s std::default_random_engine +
// std::uniform_int_distribution<size_t> produce for the reference shufflers' Fisher-Yates loop
// (semantics of samgraph/common/dist/dist_shuffler.cc:112-131; this file is our own code).
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
int main(int argc, char **argv) {
  size_t n = strtoull(argv[1], nullptr, 10);
  int epochs = atoi(argv[2]);
  std::vector<int> data(n);
  for (size_t i = 0; i < n; ++i) data[i] = (int)i;
  for (int e = 0; e < epochs; ++e) {
    auto g = std::default_random_engine(e);
    for (size_t i = n - 1; i > 0; i--) {
      std::uniform_int_distribution<size_t> d(0, i);
      std::swap(data[i], data[d(g)]);
    }
    for (size_t i = 0; i < n; ++i) printf("%d ", data[i]);
    printf("\n");
  }
  return 0;
}
