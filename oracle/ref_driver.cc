/*
 * ref_driver.cc -- runs the REFERENCE's own CPU functions (compiled from the sources where they
 * lie under /root/reference; nothing is copied into this repo) on raw little-endian array files,
 * so that tests/golden/make_golden.py can record their outputs as golden vectors.
 *
 * TEST INFRASTRUCTURE ONLY.  Built by `make -C oracle _ref` into oracle/_ref/ref_driver, and only
 * in the build container (the GPU box has no /root/reference; it uses the committed fixtures).
 *
 * What is linked: cpu/cpu_sampling_khop0.cc, cpu/cpu_sampling_khop2.cc, cpu/cpu_random.cc,
 * cpu/cpu_extraction.cc (CPUExtract and CPUMockExtract), cpu/cpu_hashtable2.cc, run_config.cc, constant.cc, logging.cc -- all
 * unmodified.  The reference's Device/Tensor layer (device.cc, common.cc) needs cuda_runtime.h and
 * is NOT built and NOT replaced: the four symbols it would provide (Device::Get, CPU, GetEnv,
 * IsEnvSet) stay unresolved: the driver is a shared object opened with dlopen(RTLD_LAZY) by the
 * ten-line launcher ref_launch.c, and the paths exercised here never call them.  Because
 * CPUHashTable2's constructor allocates through Device, the driver places the object in raw
 * storage, points its two tables at malloc'd arrays and calls the reference's own InitTable /
 * Populate / MapNodes / MapEdges / Reset bodies through qualified (non-virtual) calls.
 *
 * Job file: one command per line, processed in ONE process so the reference's thread_local
 * std::mt19937 stream (cpu_random.cc:27) and the in-place CSR mutation of CPUSampleKHop2 carry
 * across calls exactly as they do inside the reference's sampling loop.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#define private public
#define protected public
#include "cpu/cpu_hashtable2.h"
#undef private
#undef protected
#include "cpu/cpu_function.h"
#include "run_config.h"

using namespace samgraph::common;
using samgraph::common::cpu::CPUHashTable2;

template <typename T>
static std::vector<T> load(const std::string &path) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) { std::cerr << "cannot open " << path << "\n"; exit(2); }
  size_t n = f.tellg();
  f.seekg(0);
  std::vector<T> v(n / sizeof(T));
  f.read(reinterpret_cast<char *>(v.data()), n);
  return v;
}

template <typename T>
static void save(const std::string &path, const T *p, size_t n) {
  std::ofstream f(path, std::ios::binary);
  f.write(reinterpret_cast<const char *>(p), n * sizeof(T));
}

extern "C" int ref_driver_main(const char *jobfile) {
  RunConfig::omp_thread_num = 1;

  std::vector<IdType> indptr, indices;
  CPUHashTable2 *ht = nullptr;

  std::ifstream job(jobfile);
  std::string line;
  while (std::getline(job, line)) {
    std::istringstream ss(line);
    std::string cmd;
    if (!(ss >> cmd) || cmd[0] == '#') continue;
    if (cmd == "graph") {
      std::string a, b;
      ss >> a >> b;
      indptr = load<IdType>(a);
      indices = load<IdType>(b);
    } else if (cmd == "khop0" || cmd == "khop2") {
      std::string in, out;
      size_t fanout;
      ss >> in >> fanout >> out;
      auto input = load<IdType>(in);
      std::vector<IdType> src(input.size() * fanout + 1), dst(input.size() * fanout + 1);
      size_t num_out = 0;
      if (cmd == "khop0")
        cpu::CPUSampleKHop0(indptr.data(), indices.data(), input.data(), input.size(), src.data(),
                            dst.data(), &num_out, fanout);
      else
        cpu::CPUSampleKHop2(indptr.data(), indices.data(), input.data(), input.size(), src.data(),
                            dst.data(), &num_out, fanout);
      save(out + ".src.bin", src.data(), num_out);
      save(out + ".dst.bin", dst.data(), num_out);
    } else if (cmd == "dump_indices") {
      std::string out;
      ss >> out;
      save(out, indices.data(), indices.size());
    } else if (cmd == "ht_create") {
      size_t max_items;
      ss >> max_items;
      ht = static_cast<CPUHashTable2 *>(malloc(sizeof(CPUHashTable2)));
      memset(static_cast<void *>(ht), 0, sizeof(CPUHashTable2));
      ht->_o2n_table = static_cast<CPUHashTable2::BucketO2N *>(malloc(max_items * sizeof(CPUHashTable2::BucketO2N)));
      ht->_n2o_table = static_cast<CPUHashTable2::BucketN2O *>(malloc(max_items * sizeof(CPUHashTable2::BucketN2O)));
      ht->_capacity = max_items;
      ht->CPUHashTable2::InitTable();
    } else if (cmd == "ht_reset") {
      ht->CPUHashTable2::Reset();
    } else if (cmd == "ht_populate") {
      std::string in;
      ss >> in;
      auto items = load<IdType>(in);
      ht->CPUHashTable2::Populate(items.data(), items.size());
    } else if (cmd == "ht_mapnodes") {
      std::string out;
      ss >> out;
      size_t n = ht->CPUHashTable2::NumItems();
      std::vector<IdType> nodes(n + 1);
      ht->CPUHashTable2::MapNodes(nodes.data(), n);
      save(out, nodes.data(), n);
    } else if (cmd == "ht_mapedges") {
      std::string a, b, out;
      ss >> a >> b >> out;
      auto src = load<IdType>(a);
      auto dst = load<IdType>(b);
      std::vector<IdType> ns(src.size() + 1), nd(src.size() + 1);
      ht->CPUHashTable2::MapEdges(src.data(), dst.data(), src.size(), ns.data(), nd.data());
      save(out + ".src.bin", ns.data(), src.size());
      save(out + ".dst.bin", nd.data(), src.size());
    } else if (cmd == "extract") {
      std::string a, b, out;
      size_t dim;
      int dtype;
      ss >> a >> b >> dim >> dtype >> out;
      auto src = load<char>(a);
      auto idx = load<IdType>(b);
      size_t esz = dtype == kF32 || dtype == kI32 ? 4 : (dtype == kF64 || dtype == kI64 ? 8 : (dtype == kF16 ? 2 : 1));
      std::vector<char> dstv(idx.size() * dim * esz + 1);
      cpu::CPUExtract(dstv.data(), src.data(), idx.data(), idx.size(), dim, static_cast<DataType>(dtype));
      save(out, dstv.data(), idx.size() * dim * esz);
    } else if (cmd == "mock_extract") {
      // CPUMockExtract (cpu/cpu_extraction.cc:44-62, 92-116): row ids masked to a 2^bits-row table (SAMGRAPH_EMPTY_FEAT)
      std::string a, b, out;
      size_t dim, bits;
      int dtype;
      ss >> a >> b >> dim >> dtype >> bits >> out;
      auto src = load<char>(a);
      auto idx = load<IdType>(b);
      size_t esz = dtype == kF32 || dtype == kI32 ? 4 : (dtype == kF64 || dtype == kI64 ? 8 : (dtype == kF16 ? 2 : 1));
      std::vector<char> dstv(idx.size() * dim * esz + 1);
      RunConfig::option_empty_feat = bits;
      cpu::CPUMockExtract(dstv.data(), src.data(), idx.data(), idx.size(), dim, static_cast<DataType>(dtype));
      RunConfig::option_empty_feat = 0;
      save(out, dstv.data(), idx.size() * dim * esz);
    } else {
      std::cerr << "unknown command " << cmd << "\n";
      return 2;
    }
  }
  return 0;
}


/* ---- timing entry points (bench.py's cpu_baseline, kind "reference") ---------------------------------------------
 * The reference's CPU sampling path exactly as DoCPUSample / DoFeatureExtract drive it (cpu/cpu_loops.cc:55-227):
 * Reset + Populate(seeds), then per layer (last fanout first) CPUSampleKHop0/2 -> Populate -> MapNodes -> MapEdges,
 * then CPU[Mock]Extract of the input nodes' feature rows -- the reference's own functions, its OpenMP pragmas with
 * RunConfig::omp_thread_num threads, its per-thread mt19937.  The table is built like CPUHashTable2's constructor
 * does (direct-indexed, one bucket per graph node) but in malloc'd storage (see the header comment). */
struct RefBench {
  CPUHashTable2 *ht;
  std::vector<IdType> src, dst, new_src, new_dst, unique, cur;
};

extern "C" void *ref_bench_create(size_t num_node, size_t max_edges_per_layer, size_t max_unique, int threads) {
  RunConfig::omp_thread_num = threads;
  auto *b = new RefBench();
  b->ht = static_cast<CPUHashTable2 *>(malloc(sizeof(CPUHashTable2)));
  memset(static_cast<void *>(b->ht), 0, sizeof(CPUHashTable2));
  b->ht->_o2n_table = static_cast<CPUHashTable2::BucketO2N *>(malloc(num_node * sizeof(CPUHashTable2::BucketO2N)));
  b->ht->_n2o_table = static_cast<CPUHashTable2::BucketN2O *>(malloc(num_node * sizeof(CPUHashTable2::BucketN2O)));
  if (!b->ht->_o2n_table || !b->ht->_n2o_table) return nullptr;
  b->ht->_capacity = num_node;
  b->ht->CPUHashTable2::InitTable();
  b->src.resize(max_edges_per_layer + 1);
  b->dst.resize(max_edges_per_layer + 1);
  b->new_src.resize(max_edges_per_layer + 1);
  b->new_dst.resize(max_edges_per_layer + 1);
  b->unique.resize(max_unique + 1);
  b->cur.resize(max_unique + 1);
  return b;
}

extern "C" void ref_bench_set_threads(int threads) { RunConfig::omp_thread_num = threads; }

extern "C" void ref_bench_destroy(void *h) {
  auto *b = static_cast<RefBench *>(h);
  if (!b) return;
  free(b->ht->_o2n_table);
  free(b->ht->_n2o_table);
  free(b->ht);
  delete b;
}

/* one mini-batch; sample_type 0 = khop0, 5 = khop2 (mutates indices); returns 0, edge / input-node counts by pointer */
extern "C" int ref_bench_batch(void *h, const IdType *indptr, IdType *indices, const IdType *seeds, size_t num_seeds,
                               const size_t *fanout, size_t num_layers, int sample_type, const void *feat,
                               size_t feat_dim, size_t empty_feat_bits, void *feat_out, size_t *out_edges,
                               size_t *out_inputs) {
  auto *b = static_cast<RefBench *>(h);
  CPUHashTable2 *ht = b->ht;
  ht->CPUHashTable2::Reset();
  ht->CPUHashTable2::Populate(seeds, num_seeds);
  memcpy(b->cur.data(), seeds, num_seeds * sizeof(IdType));
  size_t num_input = num_seeds, edges = 0;
  for (long l = (long)num_layers - 1; l >= 0; --l) {
    size_t num_out = 0;
    if (num_input * fanout[l] >= b->src.size()) return 1;
    if (sample_type == 0)
      cpu::CPUSampleKHop0(indptr, indices, b->cur.data(), num_input, b->src.data(), b->dst.data(), &num_out, fanout[l]);
    else
      cpu::CPUSampleKHop2(indptr, indices, b->cur.data(), num_input, b->src.data(), b->dst.data(), &num_out, fanout[l]);
    ht->CPUHashTable2::Populate(b->dst.data(), num_out);
    const size_t num_unique = ht->CPUHashTable2::NumItems();
    if (num_unique >= b->unique.size()) return 1;
    ht->CPUHashTable2::MapNodes(b->unique.data(), num_unique);
    ht->CPUHashTable2::MapEdges(b->src.data(), b->dst.data(), num_out, b->new_src.data(), b->new_dst.data());
    b->cur.swap(b->unique);
    num_input = num_unique;
    edges += num_out;
  }
  if (feat && feat_out) {
    if (empty_feat_bits) {
      RunConfig::option_empty_feat = empty_feat_bits;
      cpu::CPUMockExtract(feat_out, feat, b->cur.data(), num_input, feat_dim, kF32);
    } else {
      cpu::CPUExtract(feat_out, feat, b->cur.data(), num_input, feat_dim, kF32);
    }
  }
  *out_edges = edges;
  *out_inputs = num_input;
  return 0;
}
