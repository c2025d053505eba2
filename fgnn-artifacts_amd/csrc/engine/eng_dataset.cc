#include "eng_dataset.h"

#include <algorithm>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <cctype>
#include <cerrno>
#include <cstring>
#include <fstream>
#include <unordered_map>
#include <vector>

namespace sam {

// ---- memory shared by the processes of one job -------------------------------------------------------------------
// Default: MAP_SHARED|MAP_ANONYMOUS, inherited through fork (the reference's layout: config + data_init in the parent,
// workers forked afterwards).  With SAMGRAPH_SHM_PREFIX=<name> the k-th region a process asks for is the POSIX shared
// memory object "/<name>.<k>" instead: processes that were NOT forked from a common parent (one process per GPU started
// by torchrun) run the same config + data_init and therefore meet in the same regions.  Whoever creates a region
// initialises and publishes it, everybody else blocks in SharedCreate until then.  A header page in front of the user
// pointer carries the hand-shake; it exists in both modes so that the two differ in nothing else.
namespace {
constexpr size_t kShmHeader = kShmHeaderBytes;
struct ShmHeader {
  uint64_t magic;
  uint64_t bytes;
  int ready;
};
std::vector<std::string> &OwnedNames() {
  static std::vector<std::string> v;
  return v;
}
void UnlinkOwned() {
  for (auto &n : OwnedNames()) shm_unlink(n.c_str());
  OwnedNames().clear();
}
}  // namespace

SharedRegion SharedCreate(size_t bytes) {
  if (bytes == 0) bytes = 4096;
  const size_t total = bytes + kShmHeader;
  static int counter = 0;
  const char *prefix = getenv("SAMGRAPH_SHM_PREFIX");
  SharedRegion r{nullptr, true};
  void *p = MAP_FAILED;
  if (!prefix || !*prefix) {
    p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    SAM_CHECK(p != MAP_FAILED) << "mmap of " << total << " bytes failed";
  } else {
    const std::string name = std::string("/") + prefix + "." + std::to_string(counter++);
    int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd >= 0) {
      // the creator removes the names when it exits (the mappings stay valid); SAMGRAPH_SHM_KEEP=1 leaves that to the
      // launcher, for jobs whose processes may attach after the creator has gone
      const char *keep = getenv("SAMGRAPH_SHM_KEEP");
      if (!(keep && atoi(keep))) {
        if (OwnedNames().empty()) atexit(UnlinkOwned);
        OwnedNames().push_back(name);
      }
      SAM_CHECK(ftruncate(fd, (off_t)total) == 0) << "cannot size " << name << " to " << total << " bytes";
    } else {
      SAM_CHECK(errno == EEXIST) << "shm_open " << name << ": " << strerror(errno);
      r.creator = false;
      fd = shm_open(name.c_str(), O_RDWR, 0600);
      SAM_CHECK(fd >= 0) << "shm_open " << name << ": " << strerror(errno);
      Timer t;
      for (;;) {  // the creator sizes the object right after creating it
        struct stat st;
        SAM_CHECK(fstat(fd, &st) == 0);
        if ((size_t)st.st_size == total) break;
        SAM_CHECK(st.st_size == 0) << name << " has " << st.st_size << " bytes, expected " << total
                                   << ": the processes of a job must run the same configuration";
        SAM_CHECK(t.Passed() < 600.0) << name << " was never sized by its creator";
        usleep(200);
      }
    }
    p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    SAM_CHECK(p != MAP_FAILED) << "mmap of " << name << " (" << total << " bytes) failed";
    close(fd);
  }
  auto *h = static_cast<ShmHeader *>(p);
  r.ptr = static_cast<char *>(p) + kShmHeader;
  if (r.creator) {
    h->magic = kShmMagic;
    h->bytes = bytes;
  } else {
    Timer t;
    while (__atomic_load_n(&h->ready, __ATOMIC_ACQUIRE) != 1) {
      SAM_CHECK(t.Passed() < 1800.0) << "shared region " << counter - 1 << " was never published by its creator";
      usleep(200);
    }
    SAM_CHECK(h->magic == kShmMagic && h->bytes == bytes);
  }
  return r;
}

void SharedPublish(void *ptr) {
  auto *h = reinterpret_cast<ShmHeader *>(static_cast<char *>(ptr) - kShmHeader);
  SAM_CHECK(h->magic == kShmMagic);
  __atomic_store_n(&h->ready, 1, __ATOMIC_RELEASE);
}

void *SharedAnonymous(size_t bytes) {
  SharedRegion r = SharedCreate(bytes);
  if (r.creator) SharedPublish(r.ptr);
  return r.ptr;
}

HostArray MapFile(const std::string &path, size_t expect_bytes, bool required) {
  HostArray a;
  int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) {
    SAM_CHECK(!required) << "cannot open " << path;
    return a;
  }
  struct stat st;
  SAM_CHECK(fstat(fd, &st) == 0);
  SAM_CHECK_EQ((size_t)st.st_size, expect_bytes) << path << " has the wrong size ";
  if (expect_bytes) {
    // PRIVATE + writable: khop2 never writes the host copy, but keep the mapping independent of the file
    a.ptr = mmap(nullptr, expect_bytes, PROT_READ, MAP_SHARED, fd, 0);
    SAM_CHECK(a.ptr != MAP_FAILED) << "mmap " << path;
  }
  close(fd);
  a.bytes = expect_bytes;
  a.from_file = true;
  return a;
}

// ---- NUMA placement of the host feature table ------------------------------------------------------------------------
// Every trainer GPU of the job pulls its miss rows out of this ONE table (36-55 GB/s of random 512-byte reads per
// trainer): first-touched by whoever reads the file it would sit on that process's socket, all trainers would
// load one socket's memory controllers and the GPUs of the other socket would cross the socket link for every row.
// SAMGRAPH_HOST_FEAT_NUMA: interleave (default when more than one node has memory: pages round-robin over the nodes),
// node:<n> (bind to one node -- a single-trainer job next to its GPU), local (leave it to first touch).
static std::vector<int> NodesWithMemory() {
  std::vector<int> nodes;
  std::ifstream f("/sys/devices/system/node/has_memory");
  std::string s;
  if (!(f >> s)) return nodes;
  size_t i = 0;
  while (i < s.size()) {  // "0-1,4"
    const int a = atoi(s.c_str() + i);
    int b = a;
    while (i < s.size() && isdigit((unsigned char)s[i])) ++i;
    if (i < s.size() && s[i] == '-') {
      b = atoi(s.c_str() + ++i);
      while (i < s.size() && isdigit((unsigned char)s[i])) ++i;
    }
    for (int n = a; n <= b && n < 1024; ++n) nodes.push_back(n);
    if (i < s.size() && s[i] == ',') ++i;
  }
  return nodes;
}

std::string PlaceHostTable(void *ptr, size_t bytes, const char *what) {
  const char *e = getenv("SAMGRAPH_HOST_FEAT_NUMA");
  const std::string want = e && *e ? e : "interleave";
  const std::vector<int> nodes = NodesWithMemory();
  if (want == "local" || nodes.size() < 2 || bytes == 0) return "first touch (" + std::to_string(nodes.size()) + " node(s))";
  unsigned long mask[16] = {0};
  int mode = 3;  // MPOL_INTERLEAVE
  std::string desc = "interleaved over " + std::to_string(nodes.size()) + " nodes";
  if (want.compare(0, 5, "node:") == 0) {
    char *end = nullptr;
    const long n = strtol(want.c_str() + 5, &end, 10);
    if (end == want.c_str() + 5 || *end != '\0' || n < 0 || std::find(nodes.begin(), nodes.end(), (int)n) == nodes.end()) {
      SAM_LOG(kWarning) << what << ": SAMGRAPH_HOST_FEAT_NUMA=" << want << " does not name a NUMA node with memory; pages "
                        << "go where they are first touched";
      return "first touch (bad node in SAMGRAPH_HOST_FEAT_NUMA)";
    }
    mask[n / 64] |= 1ul << (n % 64);
    mode = 2;  // MPOL_BIND
    desc = "bound to node " + std::to_string(n);
  } else {
    for (int n : nodes) mask[n / 64] |= 1ul << (n % 64);
  }
  const uintptr_t a = reinterpret_cast<uintptr_t>(ptr) & ~uintptr_t(4095);
  const size_t len = (reinterpret_cast<uintptr_t>(ptr) + bytes - a + 4095) & ~size_t(4095);
  if (syscall(SYS_mbind, a, len, mode, mask, 1024ul + 1, 0u) != 0) {
    SAM_LOG(kWarning) << what << ": mbind refused (" << strerror(errno) << "); pages go where they are first touched";
    return std::string("first touch (mbind: ") + strerror(errno) + ")";
  }
  SAM_LOG(kInfo) << what << ": " << desc;
  return desc;
}

// Reads a whole file into MAP_SHARED|MAP_ANONYMOUS memory: unlike a file-backed mapping it can be
// hipHostRegister'ed (the trainers' gather kernels read miss rows straight from it) and it is shared by
// the forked children without copies.  The reference locks its file mappings in RAM as well
// (MAP_LOCKED, common.cc:98).
static HostArray ReadFileShared(const std::string &path, size_t expect_bytes, bool place = false) {
  HostArray a;
  int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) return a;
  struct stat st;
  SAM_CHECK(fstat(fd, &st) == 0);
  SAM_CHECK_EQ((size_t)st.st_size, expect_bytes) << path << " has the wrong size ";
  SharedRegion reg = SharedCreate(expect_bytes);
  a.ptr = reg.ptr;
  a.bytes = expect_bytes;
  a.from_file = true;
  if (reg.creator) {
    if (place) (void)PlaceHostTable(a.ptr, expect_bytes, "host feature table");  // before the first touch below
    size_t done = 0;
    while (done < expect_bytes) {
      ssize_t r = pread(fd, static_cast<char *>(a.ptr) + done, expect_bytes - done, (off_t)done);
      SAM_CHECK(r > 0) << "short read from " << path;
      done += (size_t)r;
    }
    SharedPublish(a.ptr);
  }
  close(fd);
  return a;
}

void Dataset::Load(const RunConfig &rc) {
  std::string dir = rc.dataset_path;
  if (dir.back() != '/') dir.push_back('/');
  std::unordered_map<std::string, size_t> meta;
  std::ifstream mf(dir + "meta.txt");
  SAM_CHECK(mf.good()) << "cannot open " << dir << "meta.txt";
  std::string k;
  size_t v;
  while (mf >> k >> v) meta[k] = v;
  for (const char *key : {"NUM_NODE", "NUM_EDGE", "FEAT_DIM", "NUM_CLASS", "NUM_TRAIN_SET", "NUM_TEST_SET",
                          "NUM_VALID_SET"})
    SAM_CHECK(meta.count(key)) << "meta.txt lacks " << key;
  num_node = meta["NUM_NODE"];
  num_edge = meta["NUM_EDGE"];
  feat_dim = meta["FEAT_DIM"];
  num_class = meta["NUM_CLASS"];
  num_train = meta["NUM_TRAIN_SET"];
  num_test = meta["NUM_TEST_SET"];
  num_valid = meta["NUM_VALID_SET"];
  SAM_CHECK(num_edge < (1ull << 32)) << "edge ids are 32 bit";

  indptr = MapFile(dir + "indptr.bin", (num_node + 1) * 4, true);
  indices = MapFile(dir + "indices.bin", num_edge * 4, true);
  train_set = MapFile(dir + "train_set.bin", num_train * 4, true);
  test_set = MapFile(dir + "test_set.bin", num_test * 4, false);
  valid_set = MapFile(dir + "valid_set.bin", num_valid * 4, false);

  // features: file, or an uninitialised buffer (engine.cc:138-155); SAMGRAPH_EMPTY_FEAT=k => 2^k rows
  feat_rows = rc.option_empty_feat ? (1ull << rc.option_empty_feat) : num_node;
  if (!rc.option_empty_feat) feat = ReadFileShared(dir + "feat.bin", num_node * feat_dim * 4, true);
  if (!feat.ptr) {
    feat.bytes = feat_rows * feat_dim * 4;
    SharedRegion reg = SharedCreate(feat.bytes);
    feat.ptr = reg.ptr;
    if (reg.creator) {
      // (the policy belongs to the shared object: whoever faults a page in later -- hipHostRegister pins them all in
      // every trainer -- gets it placed accordingly)
      (void)PlaceHostTable(feat.ptr, feat.bytes, "host feature table");
      SharedPublish(feat.ptr);
    }
  }
  label = ReadFileShared(dir + "label.bin", num_node * 8);
  if (!label.ptr) {
    label.bytes = num_node * 8;
    label.ptr = SharedAnonymous(label.bytes);
  }
  if (rc.sample_type == kWeightedKHopPrefix)
    prob_prefix = MapFile(dir + "prob_prefix_table.bin", num_edge * 4, true);
  if (rc.sample_type == kWeightedKHop || rc.sample_type == kWeightedKHopHashDedup) {  // engine.cc:175-186
    prob_table = MapFile(dir + "prob_table.bin", num_edge * 4, true);
    alias_table = MapFile(dir + "alias_table.bin", num_edge * 4, true);
  }

  if (rc.UseGPUCache()) {
    const char *file = nullptr;
    switch (rc.cache_policy) {
      case kCacheByDegree: file = "cache_by_degree.bin"; break;
      case kCacheByHeuristic: file = "cache_by_heuristic.bin"; break;
      case kCacheByDegreeHop: file = "cache_by_degree_hop.bin"; break;
      case kCacheByFakeOptimal: file = "cache_by_fake_optimal.bin"; break;
      case kCacheByRandom: file = "cache_by_random.bin"; break;
      case kCacheByPreSample: case kCacheByPreSampleStatic: break;
      default: SAM_FATAL << "cache policy " << rc.cache_policy << " is not built";
    }
    if (file) {
      ranking_file = MapFile(dir + file, num_node * 4, true);
      ranking_nodes = static_cast<uint32_t *>(ranking_file.ptr);
    } else {
      // written by sampler 0's presample, read by every sampler and trainer (dist_engine.cc:115-127)
      ranking_nodes = static_cast<uint32_t *>(SharedAnonymous(num_node * 4));
    }
  }
  SAM_LOG(kInfo) << "dataset " << dir << ": " << num_node << " nodes, " << num_edge << " edges, dim " << feat_dim;
}

}  // namespace sam
