// eng_dataset.h -- on-disk dataset (reference engine.cc:73-264, constant.cc:23-50): raw
// little-endian arrays mapped into host memory; everything the children need is MAP_SHARED so it
// survives fork without duplication.
#pragma once
#include <string>

#include "eng_config.h"

namespace sam {

struct HostArray {
  void *ptr = nullptr;
  size_t bytes = 0;
  bool from_file = false;
};

struct Dataset {
  size_t num_node = 0, num_edge = 0, feat_dim = 0, num_class = 0;
  size_t num_train = 0, num_test = 0, num_valid = 0;
  HostArray indptr, indices, feat, label, train_set, test_set, valid_set, prob_prefix, prob_table, alias_table,
      ranking_file;
  uint32_t *ranking_nodes = nullptr;  // shared anonymous mapping for pre_sample, or the ranking file
  size_t feat_rows = 0;               // num_node, or 2^SAMGRAPH_EMPTY_FEAT
  void Load(const RunConfig &rc);
};

// Zero-filled host memory shared by the processes of a job: MAP_SHARED|MAP_ANONYMOUS inherited through fork, or -- with
// SAMGRAPH_SHM_PREFIX set -- a named POSIX shared-memory object that processes started independently (torchrun) meet in
// (eng_dataset.cc).  The creator of a region fills it and calls SharedPublish; SharedCreate blocks everybody else until then.
struct SharedRegion {
  void *ptr;
  bool creator;
};
SharedRegion SharedCreate(size_t bytes);
// layout of a named region ("/<SAMGRAPH_SHM_PREFIX>.<k>"): a header page, then the user bytes
constexpr size_t kShmHeaderBytes = 4096;
constexpr uint64_t kShmMagic = 0x46474e4e53484d31ull;  // "FGNNSHM1"; followed by the user byte count (u64)
void SharedPublish(void *ptr);
void *SharedAnonymous(size_t bytes);  // a region without initial content (published at once)
HostArray MapFile(const std::string &path, size_t expect_bytes, bool required);

}  // namespace sam
