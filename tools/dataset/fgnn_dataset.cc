// fgnn_dataset -- host-side preparation of the files the sampling engine reads next to indptr.bin / indices.bin
// (dataset layout: SURVEY.md 2.4; the engine loads them in csrc/engine/eng_dataset.cc).  One binary, sub-commands:
//
//   fgnn_dataset cache-by-degree   <dir>                 -> cache_by_degree.bin   (u32[N] node ranking)
//   fgnn_dataset cache-by-random   <dir>                 -> cache_by_random.bin
//   fgnn_dataset cache-by-heuristic <dir>                -> cache_by_heuristic.bin (train set, its neighbours, rest by degree)
//   fgnn_dataset cache-by-degree-hop <dir>               -> cache_by_degree_hop.bin (degree inside the 2-hop reach first)
//   fgnn_dataset cache-by-fake-optimal <dir> [f0 f1 [batch]] -> cache_by_fake_optimal.bin (expected 2-hop touch count)
//   fgnn_dataset 32to64            <dir>                 -> indptr64.bin, indices64.bin, {train,test,valid}_set64.bin
//   fgnn_dataset prob-prefix-table <dir> [policy]        -> prob_prefix_table.bin (f32[E], per-row inclusive sums)
//   fgnn_dataset alias-table       <dir> [policy]        -> prob_table.bin (f32[E]) + alias_table.bin (u32[E], node ids)
//   fgnn_dataset coo-to-dataset    <dir> <coo.bin>       -> indptr.bin, indices.bin, {train,valid,test}_set.bin
//   fgnn_dataset check             <dir>                 -> validates meta.txt against the CSR files
//
// What each file must contain follows the reference's generators (utility/data-process/toolkit/cache/
// cache_by_degree.cc:29-62, cache_by_random.cc:29-50, cache_by_heuristic.cc:29-101, cache_by_degree_hop.cc:30-165, cache_by_fake_optimal.cc:66-185,
// generator/32to64.cc:33-82, weight/create_prob_prefix_table.cc:87-136,
// weight/create_alias_table.cc:95-190, generator/coo_to_dataset.cc:128-222, property/csr_checker.cc); the weight
// policies are kSrcSuffix (default), kInverseSrcDegreeRand, kDefault and kInverseBothDegreeRand -- the last two draw
// random weights, here from a generator seeded per row so that runs are reproducible (the reference seeds from
// std::random_device).  `dir` is the dataset folder holding meta.txt.
//
// Build: g++ -O2 -std=c++17 -fopenmp -o tools/dataset/fgnn_dataset tools/dataset/fgnn_dataset.cc
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <queue>
#include <random>
#include <sstream>
#include <string>
#include <vector>

namespace {

[[noreturn]] void Die(const std::string &msg) {
  fprintf(stderr, "fgnn_dataset: %s\n", msg.c_str());
  exit(1);
}

struct Dataset {
  std::string dir;
  std::map<std::string, size_t> meta;
  std::vector<uint32_t> indptr, indices;
  size_t num_node = 0, num_edge = 0;
};

std::map<std::string, size_t> ReadMeta(const std::string &dir) {
  std::ifstream in(dir + "meta.txt");
  if (!in) Die("cannot open " + dir + "meta.txt");
  std::map<std::string, size_t> m;
  std::string k;
  size_t v;
  while (in >> k >> v) m[k] = v;
  for (const char *need : {"NUM_NODE", "NUM_EDGE", "FEAT_DIM", "NUM_CLASS", "NUM_TRAIN_SET", "NUM_VALID_SET", "NUM_TEST_SET"})
    if (!m.count(need)) Die(std::string("meta.txt lacks ") + need);
  return m;
}

template <typename T>
std::vector<T> ReadFile(const std::string &path, size_t count) {
  std::ifstream in(path, std::ios::binary | std::ios::ate);
  if (!in) Die("cannot open " + path);
  const size_t bytes = (size_t)in.tellg();
  if (bytes != count * sizeof(T)) Die(path + ": expected " + std::to_string(count * sizeof(T)) + " bytes, found " + std::to_string(bytes));
  in.seekg(0);
  std::vector<T> v(count);
  in.read(reinterpret_cast<char *>(v.data()), (std::streamsize)bytes);
  return v;
}

template <typename T>
void WriteFile(const std::string &path, const std::vector<T> &v) {
  std::ofstream out(path, std::ios::binary | std::ios::trunc);
  if (!out) Die("cannot write " + path);
  out.write(reinterpret_cast<const char *>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
  printf("wrote %s (%zu bytes)\n", path.c_str(), v.size() * sizeof(T));
}

Dataset Load(std::string dir) {
  if (dir.empty() || dir.back() != '/') dir += '/';
  Dataset d;
  d.dir = dir;
  d.meta = ReadMeta(dir);
  d.num_node = d.meta["NUM_NODE"];
  d.num_edge = d.meta["NUM_EDGE"];
  d.indptr = ReadFile<uint32_t>(dir + "indptr.bin", d.num_node + 1);
  d.indices = ReadFile<uint32_t>(dir + "indices.bin", d.num_edge);
  return d;
}

// rows hold the sampled-from neighbours of a node, so the row length is the in-degree and the number of rows a node
// appears in is its out-degree (graph_loader.cc:115-146)
std::vector<uint32_t> OutDegrees(const Dataset &d) {
  std::vector<uint32_t> out(d.num_node, 0);
  for (size_t e = 0; e < d.num_edge; ++e) {
    if (d.indices[e] >= d.num_node) Die("indices.bin holds a node id >= NUM_NODE");
    ++out[d.indices[e]];
  }
  return out;
}

int CacheByDegree(const Dataset &d) {
  const std::vector<uint32_t> out = OutDegrees(d);
  std::vector<uint32_t> rank(d.num_node);
  for (size_t i = 0; i < d.num_node; ++i) rank[i] = (uint32_t)i;
  // descending (out-degree, id): what std::greater<pair<degree, id>> gives
  std::sort(rank.begin(), rank.end(), [&](uint32_t a, uint32_t b) { return out[a] != out[b] ? out[a] > out[b] : a > b; });
  WriteFile(d.dir + "cache_by_degree.bin", rank);
  return 0;
}

int CacheByRandom(const Dataset &d) {
  std::vector<uint32_t> rank(d.num_node);
  for (size_t i = 0; i < d.num_node; ++i) rank[i] = (uint32_t)i;
  std::mt19937 gen;  // default seed, as the reference: the file is the same on every machine with this libstdc++
  for (uint32_t i = 0; i < d.num_node; ++i) {
    std::uniform_int_distribution<uint32_t> pick(0, (uint32_t)d.num_node - i - 1);
    std::swap(rank[d.num_node - i - 1], rank[pick(gen)]);
  }
  WriteFile(d.dir + "cache_by_random.bin", rank);
  return 0;
}

std::vector<uint32_t> RankByDegreeDesc(const std::vector<uint32_t> &deg) {
  std::vector<uint32_t> rank(deg.size());
  for (size_t i = 0; i < deg.size(); ++i) rank[i] = (uint32_t)i;
  std::sort(rank.begin(), rank.end(), [&](uint32_t a, uint32_t b) { return deg[a] != deg[b] ? deg[a] > deg[b] : a > b; });
  return rank;
}

std::vector<uint32_t> TrainSet(const Dataset &d) {
  return ReadFile<uint32_t>(d.dir + "train_set.bin", d.meta.at("NUM_TRAIN_SET"));
}

// cache_by_heuristic.cc:54-88: the train set in file order, then its not yet listed neighbours in row order, then
// everything else by descending (out-degree, id)
int CacheByHeuristic(const Dataset &d) {
  const std::vector<uint32_t> by_degree = RankByDegreeDesc(OutDegrees(d));
  const std::vector<uint32_t> train = TrainSet(d);
  std::vector<uint32_t> rank;
  rank.reserve(d.num_node);
  std::vector<bool> added(d.num_node, false);
  for (uint32_t v : train) {
    if (v >= d.num_node) Die("train_set.bin holds a node id >= NUM_NODE");
    if (added[v]) Die("train_set.bin lists a node twice");
    added[v] = true;
    rank.push_back(v);
  }
  for (uint32_t v : train)
    for (uint32_t e = d.indptr[v]; e < d.indptr[v + 1]; ++e) {
      const uint32_t u = d.indices[e];
      if (!added[u]) {
        added[u] = true;
        rank.push_back(u);
      }
    }
  for (uint32_t v : by_degree)
    if (!added[v]) {
      added[v] = true;
      rank.push_back(v);
    }
  if (rank.size() != d.num_node) Die("node number mismatch");
  WriteFile(d.dir + "cache_by_heuristic.bin", rank);
  return 0;
}

// cache_by_degree_hop.cc:30-165: nodes the train set reaches within two hops are ranked first, by their out-degree
// counted over the rows of reached nodes only (flag 0x40000000 on the degree); the others follow by whole-graph
// out-degree.  (The reference leaves the row lengths of unreached nodes uninitialised, :86-93; zero is what it means.)
int CacheByDegreeHop(const Dataset &d) {
  std::vector<uint32_t> deg = OutDegrees(d);
  std::vector<uint8_t> reached(d.num_node, 0), frontier(d.num_node, 0);
  for (uint32_t v : TrainSet(d)) reached[v] = frontier[v] = 1;
  for (int hop = 0; hop < 2; ++hop) {
    std::vector<uint8_t> next(d.num_node, 0);
    for (size_t v = 0; v < d.num_node; ++v) {
      if (!frontier[v]) continue;
      for (uint32_t e = d.indptr[v]; e < d.indptr[v + 1]; ++e)
        if (!reached[d.indices[e]]) next[d.indices[e]] = 1;
    }
    for (size_t v = 0; v < d.num_node; ++v)
      if (next[v]) reached[v] = 1;
    frontier.swap(next);
  }
  std::vector<uint32_t> sub(d.num_node, 0);
  for (size_t v = 0; v < d.num_node; ++v)
    if (reached[v])
      for (uint32_t e = d.indptr[v]; e < d.indptr[v + 1]; ++e) ++sub[d.indices[e]];
  for (size_t v = 0; v < d.num_node; ++v)
    if (reached[v]) deg[v] = sub[v] | 0x40000000u;
  WriteFile(d.dir + "cache_by_degree_hop.bin", RankByDegreeDesc(deg));
  return 0;
}

// cache_by_fake_optimal.cc:66-164: expected number of batches (of `batch` train nodes; the reference uses 1) whose
// 2-hop sample touches a node, under fanout {f0, f1} = {25, 10} (layer 1 draws f1 of a train node's row, layer 0 draws
// f0 of every row reached): per batch, miss1[v] = prod over train nodes t listing v of max(0, 1 - f1/deg(t)), then
// miss2[w] = prod over touched h listing w of 1 - (1 - miss1[h]) * min(1, f0/deg(h)); train nodes of the batch are
// certain hits in both tables; expectation += 1 - miss1 * miss2.  Ranking = descending (expectation, id).  Products are
// taken in the reference's order (train index, then row position; touched nodes grouped by id % 1 thread = insertion
// order) so the doubles agree bit for bit with a one-thread run of the reference.
int CacheByFakeOptimal(const Dataset &d, double f0, double f1, size_t batch) {
  const std::vector<uint32_t> train = TrainSet(d);
  const size_t n = d.num_node;
  std::vector<double> expect(n, 0.0), miss1(n, 1.0), miss2(n, 1.0);
  std::vector<uint8_t> seen(n, 0);
  std::vector<uint32_t> touched;
  auto touch = [&](uint32_t v) {
    if (!seen[v]) {
      seen[v] = 1;
      touched.push_back(v);
    }
  };
  if (batch == 0) batch = 1;
  for (size_t b0 = 0; b0 < train.size(); b0 += batch) {
    const size_t b1 = std::min(b0 + batch, train.size());
    touched.clear();
    for (size_t i = b0; i < b1; ++i) touch(train[i]);
    for (size_t i = b0; i < b1; ++i) {
      const uint32_t t = train[i];
      const uint32_t deg = d.indptr[t + 1] - d.indptr[t];
      const double miss = std::max(0.0, 1 - f1 / static_cast<double>(deg));
      for (uint32_t e = d.indptr[t]; e < d.indptr[t + 1]; ++e) {
        miss1[d.indices[e]] *= miss;
        touch(d.indices[e]);
      }
    }
    for (size_t i = b0; i < b1; ++i) miss1[train[i]] = 0.0;
    const size_t hop1 = touched.size();  // train nodes + first hop; the loop below appends the second hop
    for (size_t j = 0; j < hop1; ++j) {
      const uint32_t h = touched[j];
      const uint32_t deg = d.indptr[h + 1] - d.indptr[h];
      const double path_miss = 1 - (1 - miss1[h]) * std::min(1.0, f0 / static_cast<double>(deg));
      for (uint32_t e = d.indptr[h]; e < d.indptr[h + 1]; ++e) {
        miss2[d.indices[e]] *= path_miss;
        touch(d.indices[e]);
      }
    }
    for (size_t i = b0; i < b1; ++i) miss2[train[i]] = 0.0;
    for (uint32_t c : touched) {
      if (!(miss1[c] == 1 && miss2[c] == 1)) expect[c] += 1 - miss1[c] * miss2[c];
      miss1[c] = miss2[c] = 1.0;
      seen[c] = 0;
    }
  }
  std::vector<uint32_t> rank(n);
  for (size_t i = 0; i < n; ++i) rank[i] = (uint32_t)i;
  std::sort(rank.begin(), rank.end(), [&](uint32_t a, uint32_t b) { return expect[a] != expect[b] ? expect[a] > expect[b] : a > b; });
  WriteFile(d.dir + "cache_by_fake_optimal.bin", rank);
  return 0;
}

// generator/32to64.cc:33-82: 64-bit copies of the topology and node sets for the loaders of the DGL / PyG baselines
int To64(const Dataset &d) {
  auto widen = [&](const std::vector<uint32_t> &v, const std::string &name) {
    WriteFile(d.dir + name, std::vector<uint64_t>(v.begin(), v.end()));
  };
  widen(d.indptr, "indptr64.bin");
  widen(d.indices, "indices64.bin");
  for (const char *set : {"train", "test", "valid"}) {
    std::string key = std::string("NUM_") + set + "_SET";
    for (auto &c : key) c = (char)toupper(c);
    widen(ReadFile<uint32_t>(d.dir + set + "_set.bin", d.meta.at(key)), std::string(set) + "_set64.bin");
  }
  return 0;
}

enum Policy { kDefault, kInverseBothDegreeRand, kInverseSrcDegreeRand, kSrcSuffix };

Policy ParsePolicy(const char *s) {
  if (!s) return kSrcSuffix;
  const std::string p(s);
  if (p == "kDefault") return kDefault;
  if (p == "kInverseBothDegreeRand") return kInverseBothDegreeRand;
  if (p == "kInverseSrcDegreeRand") return kInverseSrcDegreeRand;
  if (p == "kSrcSuffix") return kSrcSuffix;
  Die("unknown weight policy " + p);
}

struct Weigher {
  Policy policy;
  const std::vector<uint32_t> &out_deg;
  const Dataset &d;
  // weight of the edge src -> dst (dst = the row)
  float operator()(uint32_t src, uint32_t dst, std::mt19937 &gen) const {
    switch (policy) {
      case kDefault: return (float)std::uniform_int_distribution<uint32_t>(1, 10)(gen);
      case kInverseBothDegreeRand: {
        const uint32_t in_deg = d.indptr[dst + 1] - d.indptr[dst];
        return (float)(1.0 / std::uniform_int_distribution<uint32_t>(1, std::max(out_deg[src], in_deg))(gen));
      }
      case kInverseSrcDegreeRand: return (float)(1.0 / out_deg[src]);
      case kSrcSuffix: return out_deg[src] < 10 ? 100.0f : 1.0f;
    }
    return 1.0f;
  }
};

int ProbPrefixTable(const Dataset &d, Policy policy) {
  const std::vector<uint32_t> out = OutDegrees(d);
  const Weigher weigh{policy, out, d};
  std::vector<float> table(d.num_edge);
#pragma omp parallel for schedule(dynamic, 4096)
  for (size_t row = 0; row < d.num_node; ++row) {
    std::mt19937 gen((uint32_t)row);
    float sum = 0.0f;
    for (uint32_t e = d.indptr[row]; e < d.indptr[row + 1]; ++e) {
      sum += weigh(d.indices[e], (uint32_t)row, gen);
      table[e] = sum;
    }
  }
  WriteFile(d.dir + "prob_prefix_table.bin", table);
  return 0;
}

int AliasTable(const Dataset &d, Policy policy) {
  const std::vector<uint32_t> out = OutDegrees(d);
  const Weigher weigh{policy, out, d};
  std::vector<float> prob(d.num_edge, 0.0f);
  std::vector<uint32_t> alias(d.num_edge, 0u);  // entries with prob 1 keep alias 0, as the reference's files do
#pragma omp parallel for schedule(dynamic, 4096)
  for (size_t row = 0; row < d.num_node; ++row) {
    const uint32_t off = d.indptr[row], len = d.indptr[row + 1] - off;
    if (len == 0) continue;
    std::mt19937 gen((uint32_t)row);
    std::vector<float> w(len);
    float sum = 0.0f;
    for (uint32_t i = 0; i < len; ++i) {
      w[i] = weigh(d.indices[off + i], (uint32_t)row, gen);
      sum += w[i];
    }
    for (uint32_t i = 0; i < len; ++i) {
      w[i] /= sum;
      w[i] *= (float)len;
    }
    // Vose's construction with two FIFO queues: a "small" column is topped up by the front "large" column
    std::queue<uint32_t> small, large;
    for (uint32_t i = 0; i < len; ++i) (w[i] < 1.0f ? small : large).push(i);
    while (!small.empty() && !large.empty()) {
      const uint32_t s = small.front(), l = large.front();
      small.pop();
      large.pop();
      prob[off + s] = w[s];
      alias[off + s] = d.indices[off + l];
      w[l] -= (1 - w[s]);
      (w[l] < 1.0f ? small : large).push(l);
    }
    for (; !large.empty(); large.pop()) prob[off + large.front()] = 1.0f;
    for (; !small.empty(); small.pop()) prob[off + small.front()] = 1.0f;
  }
  WriteFile(d.dir + "prob_table.bin", prob);
  WriteFile(d.dir + "alias_table.bin", alias);
  return 0;
}

int CooToDataset(std::string dir, const std::string &coo_path) {
  if (dir.empty() || dir.back() != '/') dir += '/';
  auto meta = ReadMeta(dir);
  const size_t n = meta["NUM_NODE"], m = meta["NUM_EDGE"];
  if (n >= 0xffffffffull || m >= 0xffffffffull) Die("ids and offsets are 32-bit");
  const std::vector<uint32_t> coo = ReadFile<uint32_t>(coo_path, 2 * m);  // (src, dst) pairs
  // CSR by destination, sources ascending inside a row: counting sort by dst, then per-row sort
  std::vector<uint32_t> indptr(n + 1, 0), indices(m);
  for (size_t e = 0; e < m; ++e) {
    if (coo[2 * e] >= n || coo[2 * e + 1] >= n) Die("coo holds a node id >= NUM_NODE");
    ++indptr[coo[2 * e + 1] + 1];
  }
  for (size_t i = 0; i < n; ++i) indptr[i + 1] += indptr[i];
  std::vector<uint32_t> fill(indptr.begin(), indptr.end() - 1);
  for (size_t e = 0; e < m; ++e) indices[fill[coo[2 * e + 1]]++] = coo[2 * e];
#pragma omp parallel for schedule(dynamic, 4096)
  for (size_t i = 0; i < n; ++i) std::sort(indices.begin() + indptr[i], indices.begin() + indptr[i + 1]);
  WriteFile(dir + "indptr.bin", indptr);
  WriteFile(dir + "indices.bin", indices);
  // node sets: distinct nodes with at least one neighbour, drawn with the default-seeded mt19937
  std::vector<bool> taken(n, false);
  std::mt19937 gen;
  std::uniform_int_distribution<uint32_t> pick(0, (uint32_t)n - 1);
  size_t eligible = 0;
  for (size_t i = 0; i < n; ++i) eligible += indptr[i + 1] > indptr[i];
  if (meta["NUM_TRAIN_SET"] + meta["NUM_TEST_SET"] + meta["NUM_VALID_SET"] > eligible)
    Die("node sets ask for more nodes than have neighbours");
  auto draw = [&](size_t count) {
    std::vector<uint32_t> set;
    set.reserve(count);
    while (set.size() < count) {
      const uint32_t v = pick(gen);
      if (indptr[v + 1] > indptr[v] && !taken[v]) {
        set.push_back(v);
        taken[v] = true;
      }
    }
    return set;
  };
  const auto train = draw(meta["NUM_TRAIN_SET"]);  // order matters: train, test, valid share one generator
  const auto test = draw(meta["NUM_TEST_SET"]);
  const auto valid = draw(meta["NUM_VALID_SET"]);
  WriteFile(dir + "train_set.bin", train);
  WriteFile(dir + "valid_set.bin", valid);
  WriteFile(dir + "test_set.bin", test);
  return 0;
}

int Check(const Dataset &d) {
  if (d.indptr[0] != 0) Die("indptr[0] != 0");
  for (size_t i = 0; i < d.num_node; ++i)
    if (d.indptr[i + 1] < d.indptr[i]) Die("indptr decreases at row " + std::to_string(i));
  if (d.indptr[d.num_node] != d.num_edge) Die("indptr[N] != NUM_EDGE");
  for (size_t e = 0; e < d.num_edge; ++e)
    if (d.indices[e] >= d.num_node) Die("indices[" + std::to_string(e) + "] >= NUM_NODE");
  for (const char *set : {"train", "valid", "test"}) {
    std::string key = std::string("NUM_") + (set[1] == 'r' ? "TRAIN" : set[0] == 'v' ? "VALID" : "TEST") + "_SET";
    const auto s = ReadFile<uint32_t>(d.dir + set + "_set.bin", d.meta.at(key));
    for (uint32_t v : s)
      if (v >= d.num_node) Die(std::string(set) + "_set.bin holds a node id >= NUM_NODE");
  }
  printf("ok: %zu nodes, %zu edges\n", d.num_node, d.num_edge);
  return 0;
}

}  // namespace

int main(int argc, char **argv) {
  if (argc < 3) Die("usage: fgnn_dataset <cache-by-degree|cache-by-random|cache-by-heuristic|cache-by-degree-hop|cache-by-fake-optimal|32to64|prob-prefix-table|alias-table|coo-to-dataset|check> <dir> [arg]");
  const std::string cmd = argv[1];
  if (cmd == "coo-to-dataset") {
    if (argc < 4) Die("coo-to-dataset needs <dir> <coo.bin>");
    return CooToDataset(argv[2], argv[3]);
  }
  const Dataset d = Load(argv[2]);
  if (cmd == "cache-by-degree") return CacheByDegree(d);
  if (cmd == "cache-by-random") return CacheByRandom(d);
  if (cmd == "cache-by-heuristic") return CacheByHeuristic(d);
  if (cmd == "cache-by-degree-hop") return CacheByDegreeHop(d);
  if (cmd == "cache-by-fake-optimal")
    return CacheByFakeOptimal(d, argc > 4 ? atof(argv[3]) : 25, argc > 4 ? atof(argv[4]) : 10, argc > 5 ? (size_t)atoll(argv[5]) : 1);
  if (cmd == "32to64") return To64(d);
  if (cmd == "prob-prefix-table") return ProbPrefixTable(d, ParsePolicy(argc > 3 ? argv[3] : nullptr));
  if (cmd == "alias-table") return AliasTable(d, ParsePolicy(argc > 3 ? argv[3] : nullptr));
  if (cmd == "check") return Check(d);
  Die("unknown command " + cmd);
}
