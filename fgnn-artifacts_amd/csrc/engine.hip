// engine.hip -- per-batch driver: the MI355X restatement of DoGPUSample
// (reference samgraph/common/cuda/cuda_loops.cc:50-267), DoGetCacheMissIndex
// (dist/dist_loops.cc:271-323) and DoGPUFeatureExtract (cuda_loops.cc:726-770).
//
// Everything a batch needs is enqueued on one stream with device-resident sizes; the host gets one
// pinned fgnn_batch_meta per batch.  Bit-identical to oracle fgnn_oracle_do_sample.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "fgnn_device.h"

constexpr int kSlots = 6;  // batches that may be in flight at once (each on its own stream if the caller wishes); a
                           // multiple of 1, 2, 3 and 6 streams, so that a caller rotating over that many streams
                           // brings every slot back on the stream that used it last (no event between its uses)

struct fgnn_sampler {
  fgnn_sampler_config cfg;
  size_t max_nodes;                       // PredictNumNodes(batch, fanout, L)
  size_t in_cap[FGNN_MAX_LAYERS];         // worst-case #seeds of layer l
  size_t edge_cap[FGNN_MAX_LAYERS];       // worst-case #edges of layer l
  size_t max_edge_cap;
  size_t ws_bytes;
  // Per in-flight batch ("slot" = sequence number % kSlots): its own dedup table, scratch and temporaries.  The table
  // is reset right after its last use in the batch (a generation bump, hashtable.hip).
  struct Slot {
    fgnn_hashtable *ht = nullptr;
    uint32_t *tmp_dst = nullptr;          // [max_edge_cap] sampled neighbours (global ids)
    void *ws = nullptr;                   // kernel scratch
    // `done`: the slot's last batch has finished its sampling stage; `csr`: it has enqueued its last sampler kernel
    // (khop2 only).  Both are recorded only where the last hand-over crossed streams (see `csr_cross` below).
    hipEvent_t done = nullptr, csr = nullptr;
    hipStream_t last_st = nullptr;        // stream of the slot's last batch
    bool was_used = false;
    bool done_recorded = false, csr_recorded = false;  // the event covers the slot's last batch
    bool expect_cross = false;            // the slot's last reuse came from another stream: record `done` for the next
    // a batch whose sampling chain has been enqueued and whose tail (the last layer's dedup fill, fix-ups, the table's
    // generation bump) has not yet: fgnn_sampler_sample_begin / _end
    struct Pend {
      bool active = false;
      uint64_t seq = 0;
      fgnn_batch *out = nullptr;
      hipStream_t st = nullptr;           // the chain's stream: the tail goes there too
      bool ran = false;                   // the batch had seeds: layer 0's fill is owed
      size_t ecap0 = 0;                   // worst-case edges of layer 0 for THIS batch
      bool resolved0 = false, inserted0 = false;
      fgnn::FixTail owed;                 // the remap fix-up the layer-1 fill left for the next launch to carry
    } pend;
    fgnn::ScanWsHost scan_sample;         // look-back descriptors of the single-pass sampler
    uint32_t *rank_bitmap = nullptr;      // with-replacement samplers: seed ranking bitmap over the node ids (all zero
                                          // between batches), fgnn::RankWs
  } slot[kSlots];
  // host-side sequencing (calls may come from several threads, one per stream): call `seq` may start once
  // call seq - kSlots has returned; for khop2 (which swaps CSR entries in place) the sampler kernels of call
  // seq are enqueued only after call seq - 1 has enqueued its last sampler kernel, so the CSR is mutated in
  // batch order no matter how the batches overlap.
  std::mutex mu;
  std::condition_variable cv;
  uint64_t next_seq = 0;                  // for the unordered entry point
  uint64_t returned = 0;                  // calls [0, returned) have returned
  uint64_t csr_passed = 0;                // calls [0, csr_passed) have enqueued their last sampler kernel
  bool done_flag[kSlots] = {};
  bool csr_flag[kSlots] = {};
  // Events cost host time per batch (a record and a wait each), and most callers never need them: a single-stream
  // caller orders everything by the stream, a caller rotating over 2, 3 or 6 streams meets every slot on its own
  // stream again.  So a slot's `done` is recorded only if the slot's LAST reuse came from another stream, the CSR
  // hand-over `csr` only if the last batch followed its predecessor on another stream (what happened last is the guess
  // for what happens next); a reuse that finds no recorded event waits for the device -- it then covers more than
  // needed: correct, and only at a change of pattern (e.g. the pre-sampling epoch's stream -> batch streams).  No stream
  // handle of an earlier call is ever used again.
  std::atomic<bool> csr_cross{false};
  // weighted_khop_prefix: 5-ary search trees over the long rows of the prefix table (prefix_tree.hip), built once
  fgnn::PrefixTreeHost *ptree = nullptr;
  int opt_split_l0 = -1;        // FGNN_KHOP_SPLIT_L0 = 0 (profiling build): fused last layer
  int opt_unordered = 0;        // FGNN_KHOP2_UNORDERED = 1 (profiling build): wrong results under overlap
};

struct fgnn_batch {
  const fgnn_sampler *owner;
  uint32_t *row[FGNN_MAX_LAYERS], *col[FGNN_MAX_LAYERS], *data[FGNN_MAX_LAYERS];
  uint32_t *input_nodes, *output_nodes;
  uint32_t *cidx[4];                      // miss_src, miss_dst, cache_src, cache_dst
  void *feat, *label;
  size_t feat_dim, feat_rows_cap;
  int feat_dtype, label_dtype;
  size_t num_output;                      // host copy of the batch size
  fgnn_batch_meta *d_meta, *h_meta;
  void *ws;                               // scratch for the cache split
  size_t ws_bytes;
  hipEvent_t done;
  hipEvent_t t0, t1, t2;                  // optional (fgnn_batch_enable_timing): t0..t1 bracket the feature gather of
                                          // fgnn_batch_extract / the miss-row gather of fgnn_batch_extract_cached,
                                          // t1..t2 the cached-row gather
  bool timed2;
  bool stamped;                           // the last fgnn_batch_extract launch posted its workgroups' clocks to h_stamps
  unsigned long long *h_stamps;           // pinned: per-workgroup start / end clocks of the one-launch cached extraction
  size_t stamp_cap, stamp_grid, stamp_link;
  bool timing, timed;
  bool meta_copied;                       // the last extract launch of this batch also copied the summary to h_meta
  fgnn::ScanWsHost *scan;                 // look-back descriptors of the one-launch cache split
  uint32_t feat_row_mask;                 // SAMGRAPH_EMPTY_FEAT mock extraction (all ones = off)
};

namespace fgnn {
namespace {

// a caller-chosen feat_rows_cap below the worst case (fgnn_batch_create): the gather clamps to the capacity, this
// marks the batch so that the host does not take the truncated tensor for the whole one (fgnn_hip.h, `overflow`)
__global__ void batch_rows_overflow_kernel(fgnn_batch_meta *m, uint32_t cap) {
  if (m->num_input > cap) m->overflow = 1u;
}

size_t dtype_size(int dtype) {
  switch (dtype) {
    case FGNN_I8: case FGNN_U8: return 1;
    case FGNN_F16: return 2;
    case FGNN_F32: case FGNN_I32: return 4;
    case FGNN_F64: case FGNN_I64: return 8;
    default: return 0;
  }
}

}  // namespace
}  // namespace fgnn

using namespace fgnn;

extern "C" fgnn_sampler *fgnn_sampler_create(const fgnn_sampler_config *cfg, int *h_err) {
  auto fail = [&](int code) -> fgnn_sampler * {
    if (h_err) *h_err = code;
    return nullptr;
  };
  if (!cfg || !cfg->indptr || !cfg->indices || cfg->num_layers == 0 || cfg->num_layers > FGNN_MAX_LAYERS ||
      cfg->max_batch_size == 0)
    return fail(FGNN_EINVAL);
  switch (cfg->sample_type) {
    case FGNN_KHOP0:
    case FGNN_KHOP1:
    case FGNN_KHOP2:
      break;
    case FGNN_WEIGHTED_KHOP:
      if (!cfg->prob_table || !cfg->alias_table) return fail(FGNN_EINVAL);
      break;
    case FGNN_WEIGHTED_KHOP_HASH_DEDUP:
      if (!cfg->prob_table || !cfg->alias_table) return fail(FGNN_EINVAL);
      for (size_t l = 0; l < cfg->num_layers; ++l)
        if (cfg->fanout[l] > 50) return fail(FGNN_EINVAL);  // the reference's 50-slot table
      break;
    case FGNN_WEIGHTED_KHOP_PREFIX:
      if (!cfg->prob_prefix) return fail(FGNN_EINVAL);
      break;
    case FGNN_RANDOM_WALK:
      if (cfg->walk_len == 0 || cfg->num_walks == 0) return fail(FGNN_EINVAL);
      break;
    default:
      return fail(FGNN_EINVAL);
  }
  auto *s = new (std::nothrow) fgnn_sampler();
  if (!s) return fail(FGNN_EHIP);
  s->cfg = *cfg;
  s->opt_split_l0 = fgnn::tune_int("FGNN_KHOP_SPLIT_L0", -1);
  s->opt_unordered = fgnn::tune_int("FGNN_KHOP2_UNORDERED", 0);
  // worst-case sizes, layer L-1 first (cuda_loops.cc:87)
  size_t count = cfg->max_batch_size;
  s->max_edge_cap = 0;
  for (long l = (long)cfg->num_layers - 1; l >= 0; --l) {
    if (cfg->fanout[l] == 0) { delete s; return fail(FGNN_EINVAL); }
    s->in_cap[l] = count;
    s->edge_cap[l] = count * cfg->fanout[l];
    if (s->edge_cap[l] > s->max_edge_cap) s->max_edge_cap = s->edge_cap[l];
    count += s->edge_cap[l];
  }
  s->max_nodes = count;
  if (s->max_edge_cap >= 0x7fffffffull || s->max_nodes >= 0x7fffffffull) { delete s; return fail(FGNN_EINVAL); }
  // dedup scratch (pos + sums) plus the fused sampler's own block offsets behind it
  s->ws_bytes = 2 * fgnn_scratch_bytes(s->max_edge_cap > s->max_nodes ? s->max_edge_cap : s->max_nodes) +
                (s->max_edge_cap / 16 + 64) * sizeof(uint32_t);
  for (size_t l = 0; l < cfg->num_layers; ++l) {
    size_t need = 0;
    if (cfg->sample_type == FGNN_WEIGHTED_KHOP_PREFIX || cfg->sample_type == FGNN_KHOP1 ||
        cfg->sample_type == FGNN_WEIGHTED_KHOP)
      need = fgnn_weighted_scratch_bytes(s->in_cap[l], cfg->fanout[l]);
    if (cfg->sample_type == FGNN_RANDOM_WALK) need = fgnn_random_walk_scratch_bytes(s->in_cap[l], cfg->fanout[l]);
    if (need > s->ws_bytes) s->ws_bytes = need;
  }
  for (auto &sl : s->slot) {
    int err = FGNN_OK;
    sl.ht = fgnn_hashtable_create_ex(s->max_nodes, s->max_edge_cap, &err);  // fills are at most a layer's edges
    bool ok = sl.ht != nullptr;
    ok = ok && hipMalloc(&sl.tmp_dst, s->max_edge_cap * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc(&sl.ws, s->ws_bytes) == hipSuccess;
    for (hipEvent_t *e : {&sl.done, &sl.csr})
      ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
    ok = ok && sl.scan_sample.create(s->max_nodes / 64 + 2) == FGNN_OK;
    if (cfg->num_node && (cfg->sample_type == FGNN_WEIGHTED_KHOP_PREFIX || cfg->sample_type == FGNN_KHOP1 ||
                          cfg->sample_type == FGNN_WEIGHTED_KHOP)) {
      const size_t bytes = fgnn::rank_ws_bytes(cfg->num_node);
      ok = ok && hipMalloc(&sl.rank_bitmap, bytes) == hipSuccess && hipMemset(sl.rank_bitmap, 0, bytes) == hipSuccess;
    }
    if (!ok) {
      fgnn_sampler_destroy(s);
      return fail(err != FGNN_OK ? err : FGNN_EHIP);
    }
  }
  // (FGNN_PREFIX_TREE=0, profiling build: every row searched like the reference, for A/Bs)
  if (cfg->sample_type == FGNN_WEIGHTED_KHOP_PREFIX && cfg->num_node && fgnn::tune_int("FGNN_PREFIX_TREE", 1) != 0)
    s->ptree = fgnn::prefix_tree_build(cfg->indptr, cfg->prob_prefix, cfg->num_node);  // null: no long rows / no memory
  if (h_err) *h_err = FGNN_OK;
  return s;
}

extern "C" int fgnn_sampler_prefix_tree_stats(const fgnn_sampler *s, size_t out[3]) {
  if (!s || !out) return FGNN_EINVAL;
  fgnn::prefix_tree_stats(s->ptree, out);
  return FGNN_OK;
}

extern "C" void fgnn_sampler_destroy(fgnn_sampler *s) {
  if (!s) return;
  fgnn::prefix_tree_destroy(s->ptree);
  for (auto &sl : s->slot) {
    if (sl.ht) fgnn_hashtable_destroy(sl.ht);
    sl.scan_sample.destroy();
    if (sl.tmp_dst) (void)hipFree(sl.tmp_dst);
    if (sl.rank_bitmap) (void)hipFree(sl.rank_bitmap);
    if (sl.ws) (void)hipFree(sl.ws);
    for (hipEvent_t e : {sl.done, sl.csr})
      if (e) (void)hipEventDestroy(e);
  }
  delete s;
}

extern "C" size_t fgnn_sampler_max_nodes(const fgnn_sampler *s) { return s ? s->max_nodes : 0; }
extern "C" size_t fgnn_sampler_max_edges(const fgnn_sampler *s, int layer) {
  return (s && layer >= 0 && (size_t)layer < s->cfg.num_layers) ? s->edge_cap[layer] : 0;
}

extern "C" void fgnn_batch_destroy(fgnn_batch *b) {
  if (!b) return;
  for (int l = 0; l < FGNN_MAX_LAYERS; ++l) {
    if (b->row[l]) (void)hipFree(b->row[l]);
    if (b->col[l]) (void)hipFree(b->col[l]);
    if (b->data[l]) (void)hipFree(b->data[l]);
  }
  for (int k = 0; k < 4; ++k)
    if (b->cidx[k]) (void)hipFree(b->cidx[k]);
  if (b->input_nodes) (void)hipFree(b->input_nodes);
  if (b->output_nodes) (void)hipFree(b->output_nodes);
  if (b->feat) (void)hipFree(b->feat);
  if (b->label) (void)hipFree(b->label);
  if (b->d_meta) (void)hipFree(b->d_meta);
  if (b->h_meta) (void)hipHostFree(b->h_meta);
  if (b->ws) (void)hipFree(b->ws);
  if (b->scan) {
    b->scan->destroy();
    delete b->scan;
  }
  if (b->done) (void)hipEventDestroy(b->done);
  if (b->t0) (void)hipEventDestroy(b->t0);
  if (b->t1) (void)hipEventDestroy(b->t1);
  if (b->t2) (void)hipEventDestroy(b->t2);
  if (b->h_stamps) (void)hipHostFree(b->h_stamps);
  delete b;
}

extern "C" fgnn_batch *fgnn_batch_create(const fgnn_sampler *s, size_t feat_dim, int feat_dtype, int label_dtype,
                                         size_t feat_rows_cap, int *h_err) {
  auto fail = [&](int code, fgnn_batch *b) -> fgnn_batch * {
    if (b) fgnn_batch_destroy(b);
    if (h_err) *h_err = code;
    return nullptr;
  };
  if (!s) return fail(FGNN_EINVAL, nullptr);
  if (feat_dim && (dtype_size(feat_dtype) == 0 || dtype_size(label_dtype) == 0)) return fail(FGNN_EINVAL, nullptr);
  auto *b = new (std::nothrow) fgnn_batch();
  if (!b) return fail(FGNN_EHIP, nullptr);
  std::memset(static_cast<void *>(b), 0, sizeof(*b));
  b->owner = s;
  b->feat_dim = feat_dim;
  b->feat_dtype = feat_dtype;
  b->label_dtype = label_dtype;
  b->feat_row_mask = 0xFFFFFFFFu;
  b->feat_rows_cap = feat_rows_cap ? feat_rows_cap : s->max_nodes;
  if (b->feat_rows_cap > s->max_nodes) b->feat_rows_cap = s->max_nodes;
  bool ok = true;
  for (size_t l = 0; l < s->cfg.num_layers && ok; ++l) {
    ok = ok && hipMalloc(&b->row[l], s->edge_cap[l] * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc(&b->col[l], s->edge_cap[l] * sizeof(uint32_t)) == hipSuccess;
    if (s->cfg.sample_type == FGNN_RANDOM_WALK)
      ok = ok && hipMalloc(&b->data[l], s->edge_cap[l] * sizeof(uint32_t)) == hipSuccess;
  }
  ok = ok && hipMalloc(&b->input_nodes, s->max_nodes * sizeof(uint32_t)) == hipSuccess;
  ok = ok && hipMalloc(&b->output_nodes, s->cfg.max_batch_size * sizeof(uint32_t)) == hipSuccess;
  for (int k = 0; k < 4 && ok; ++k) ok = hipMalloc(&b->cidx[k], s->max_nodes * sizeof(uint32_t)) == hipSuccess;
  if (feat_dim) {
    ok = ok && hipMalloc(&b->feat, b->feat_rows_cap * feat_dim * dtype_size(feat_dtype)) == hipSuccess;
    ok = ok && hipMalloc(&b->label, s->cfg.max_batch_size * dtype_size(label_dtype)) == hipSuccess;
  }
  ok = ok && hipMalloc(&b->d_meta, sizeof(fgnn_batch_meta)) == hipSuccess;
  ok = ok && hipHostMalloc(reinterpret_cast<void **>(&b->h_meta), sizeof(fgnn_batch_meta), hipHostMallocDefault) ==
                 hipSuccess;
  b->ws_bytes = fgnn_scratch_bytes(s->max_nodes);
  ok = ok && hipMalloc(&b->ws, b->ws_bytes) == hipSuccess;
  b->scan = new (std::nothrow) fgnn::ScanWsHost();
  ok = ok && b->scan && b->scan->create(4096) == FGNN_OK;
  ok = ok && hipEventCreateWithFlags(&b->done, hipEventDisableTiming) == hipSuccess;
  if (!ok) return fail(FGNN_EHIP, b);
  if (hipMemset(b->d_meta, 0, sizeof(fgnn_batch_meta)) != hipSuccess) return fail(FGNN_EHIP, b);
  std::memset(b->h_meta, 0, sizeof(fgnn_batch_meta));
  if (h_err) *h_err = FGNN_OK;
  return b;
}

namespace {

// marks the hand-over points of call `seq` even on an early error return, so later calls never wait forever
struct SeqGuard {
  fgnn_sampler *s;
  uint64_t seq;
  hipStream_t st;
  bool csr_marked = false;
  bool finished = false;  // the success path has reset the slot's table and closed the slot itself
  bool detached = false;  // the batch's chain is enqueued, its tail follows in a later call (Slot::pend): nothing to close yet
  // end of the slot's use by this batch: remember the stream, record `done` if slots have been seen to change streams
  void close_slot() {
    fgnn_sampler::Slot &sl = s->slot[seq % kSlots];
    sl.done_recorded = sl.expect_cross && hipEventRecord(sl.done, st) == hipSuccess;
    sl.last_st = st;
    sl.was_used = true;
  }
  // the batch's last sampler kernel has been enqueued on `st` (khop2): record the hand-over event if batches have been
  // seen to follow each other on different streams, then let the next call go on
  void pass_csr(bool record_allowed) {
    if (csr_marked) return;
    fgnn_sampler::Slot &sl = s->slot[seq % kSlots];
    sl.csr_recorded = record_allowed && s->csr_cross.load(std::memory_order_relaxed) &&
                      hipEventRecord(sl.csr, st) == hipSuccess;
    sl.last_st = st;  // read by the next call (under the mutex released in mark_csr)
    mark_csr();
  }
  // An early error return leaves the slot's table with this batch's pending buckets and notes: the slot's next batch
  // must not dedup against them.
  void abandon() {
    fgnn_sampler::Slot &sl = s->slot[seq % kSlots];
    (void)fgnn::hashtable_next_generation(sl.ht, st, false);
    pass_csr(s->cfg.sample_type == FGNN_KHOP2);
    close_slot();
  }
  void mark_csr() {
    if (csr_marked) return;
    csr_marked = true;
    std::lock_guard<std::mutex> lk(s->mu);
    s->csr_flag[seq % kSlots] = true;
    while (s->csr_flag[s->csr_passed % kSlots]) {
      s->csr_flag[s->csr_passed % kSlots] = false;
      ++s->csr_passed;
    }
    s->cv.notify_all();
  }
  ~SeqGuard() {
    if (detached) return;
    if (!finished) abandon();
    mark_csr();
    std::lock_guard<std::mutex> lk(s->mu);
    s->done_flag[seq % kSlots] = true;
    while (s->done_flag[s->returned % kSlots]) {
      s->done_flag[s->returned % kSlots] = false;
      ++s->returned;
    }
    s->cv.notify_all();
  }
};

// DoGPUSample (cuda_loops.cc:50-267) for batch `seq`.  owed_fix != null: the caller takes over the last layer's remap
// fix-up (it lets a later launch of the batch carry it, FixTail); null: the batch's edge lists are final on return.
//
// Two halves.  CHAIN: everything up to and including the batch's LAST sampler launch -- for khop2, whose kernels rewrite
// CSR rows, the part that must run in batch order on the GPU: S(L-1) -> dedup -> ... -> S(0), then the next batch's
// S(L-1).  TAIL: the last layer's dedup fill, the fix-ups, the table's generation bump (and whatever the caller
// appends: cache split, gather, message pack) -- nothing of the next batch waits for it.  A caller that enqueues whole
// batches one after the other puts the next batch's chain BEHIND this batch's tail in its own enqueueing order: ~6
// launches, 25-40 us of host time during which the GPU has finished S(0) and the chain idles (an arch5 sampler process:
// 30 us between S(0) of batch k and S(L-1) of batch k + 1, a third of its 100 us per batch).  phase = kChain then kTail
// (fgnn_sampler_sample_begin / _end) lets a caller enqueue chain(k + 1) BEFORE tail(k); kWhole is both, back to back.
enum { kWhole = 0, kChain = 1, kTail = 2 };
int sample_impl(fgnn_sampler *s, uint64_t seq, const uint32_t *d_seeds, size_t num_seeds, uint64_t batch_key,
                fgnn_batch *out, void *stream, fgnn::FixTail *owed_fix, int phase = kWhole) {
  if (owed_fix) *owed_fix = fgnn::no_fix_tail();
  const bool do_chain = phase != kTail, do_tail = phase != kChain;
  if (!s || !out || out->owner != s) return FGNN_EINVAL;
  if (do_chain && ((!d_seeds && num_seeds) || num_seeds > s->cfg.max_batch_size)) return FGNN_EINVAL;
  const hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t L = s->cfg.num_layers;
  const bool mutates = s->cfg.sample_type == FGNN_KHOP2;
  const bool ordered = mutates && s->opt_unordered == 0;
  fgnn_sampler::Slot &sl = s->slot[seq % kSlots];
  if (do_chain) {
    // the slot is free once call seq - kSlots has returned (its device work is ordered below)
    std::unique_lock<std::mutex> lk(s->mu);
    // sequence numbers must be consecutive and used once; a gap would wait forever, so give up loudly instead
    if (!s->cv.wait_for(lk, std::chrono::seconds(60), [&] { return seq < s->returned + kSlots; })) return FGNN_EINVAL;
    if (seq < s->returned) return FGNN_EINVAL;
  } else if (!sl.pend.active || sl.pend.seq != seq || sl.pend.out != out || sl.pend.st != st) {
    return FGNN_EINVAL;  // no chain of this batch is waiting for its tail (on this stream)
  }
  SeqGuard guard{s, seq, st};
  if (!do_chain) {
    guard.csr_marked = true;  // (the chain half has passed the CSR on)
    sl.pend.active = false;
  }
  fgnn::ScanErrorSink sink(&out->d_meta->overflow);  // a timed-out cross-workgroup wait marks the batch invalid
  fgnn_hashtable *ht = sl.ht;
  uint32_t *tmp_dst = sl.tmp_dst;
  void *ws = sl.ws;
  int rc = FGNN_OK;
  const bool khop_fused = s->cfg.sample_type == FGNN_KHOP2 || s->cfg.sample_type == FGNN_KHOP0;
  // a layer's remap fix-up is not launched by itself: the next fill's insert launch carries it (FixTail) -- nothing of
  // the next layer's sampling reads the remapped edges -- and the last layer's goes to the caller or runs at the end
  fgnn::FixTail owed = fgnn::no_fix_tail();
  bool ran = false, resolved0 = false, inserted0 = false;
  size_t ecap0 = 0;

  // FillWithDuplicates + remap of layer l; its last pass also records num_dst / num_src / num_input of the layer
  auto fill = [&](long l, size_t ecap, bool inserted, bool resolved, bool table_free) -> int {
    size_t *d_ne = reinterpret_cast<size_t *>(&out->d_meta->num_edge[l]);
    fgnn::FixTail mine = fgnn::no_fix_tail();
    const int r = hashtable_fill_duplicates_ex(ht, tmp_dst, 0, d_ne, ecap, out->row[l], ws, s->ws_bytes, stream,
                                               LayerSummary{&out->d_meta->num_dst[l], &out->d_meta->num_src[l],
                                                            &out->d_meta->num_input},
                                               inserted, nullptr, /*final_fill=*/l == 0, resolved, &mine, &owed,
                                               table_free && l != 0);  // (the last fill decides for itself: nothing follows it)
    owed = mine;
    return r;
  };

  if (do_chain) {
    out->num_output = num_seeds;
    out->meta_copied = false;
    // the slot's scratch and table were last used kSlots batches ago: ordered by the stream itself when that was this
    // stream, by the slot's event otherwise
    sl.expect_cross = sl.was_used && sl.last_st != st;
    // Events are recorded where the LAST hand-over crossed streams (the guess for the next one).  A hand-over that crosses
    // without a recorded event -- a change of pattern: the pre-sampling epoch's stream -> the batch streams -- waits for
    // the device instead.  (Until round 5 the event was recorded late on the other stream's handle, which the caller may
    // have destroyed by then: a dangling hipStream_t is undefined behaviour, not an error return.)
    if (sl.expect_cross) {
      if (sl.done_recorded) FGNN_HIP_CHECK(hipStreamWaitEvent(st, sl.done, 0));
      else FGNN_HIP_CHECK(hipDeviceSynchronize());
    }
    if (ordered && seq > 0) {
      // khop2 swaps CSR entries in place: its kernels run in batch order even when batches overlap
      {
        std::unique_lock<std::mutex> lk(s->mu);
        if (!s->cv.wait_for(lk, std::chrono::seconds(60), [&] { return s->csr_passed >= seq; })) return FGNN_EINVAL;
      }
      fgnn_sampler::Slot &prev = s->slot[(seq - 1) % kSlots];
      const bool cross = prev.last_st != st;
      s->csr_cross.store(cross, std::memory_order_relaxed);  // this batch records its own hand-over if it needed one
      if (cross) {
        if (prev.csr_recorded) FGNN_HIP_CHECK(hipStreamWaitEvent(st, prev.csr, 0));
        else FGNN_HIP_CHECK(hipDeviceSynchronize());  // change of pattern (see above)
      }
    }
    // new nodes are appended straight into the batch's input_nodes buffer (input_nodes = unique, cuda_loops.cc:258)
    rc = fgnn_hashtable_set_n2o(ht, out->input_nodes);
    if (rc != FGNN_OK) return rc;
    // Reset state + FillWithUnique(seeds) + output_nodes copy + summary header: done by the first sampler launch itself
    // for the k-hop samplers (BatchStart), by one small launch otherwise
    const bool start_in_sampler = khop_fused && num_seeds > 0;
    if (!start_in_sampler) {
      rc = fgnn_hashtable_start_batch(ht, d_seeds, num_seeds, out->output_nodes, out->d_meta, batch_key, (uint32_t)L,
                                      stream);
      if (rc != FGNN_OK) return rc;
    }
    // Samplers that never look the dedup table up (all but the fused k-hop ones) may run EVERY fill through the
    // partitioned, table-free path -- decided once per batch: if one layer's worst case is too large for it (beyond the
    // one-launch count+assign, ~12.6 M items), a fill that fell back to the global insert would dedup against a table the
    // earlier partitioned fills never wrote to.  Then every fill but the last takes the global path
    bool table_free = !khop_fused;
    for (size_t l = L, ic = num_seeds; table_free && l-- > 0;) {
      const size_t ec = ic * s->cfg.fanout[l];
      table_free = ec == 0 || fgnn::hashtable_can_partition(ht, ec);
      ic += ec;
    }
    const uint32_t *cur = d_seeds;
    const uint32_t *d_cur_n = nullptr;  // first layer: host count
    size_t cur_n_host = num_seeds;
    size_t in_cap = num_seeds;          // tighter than the create-time worst case when the batch is short
    for (long l = (long)L - 1; l >= 0 && num_seeds; --l) {
      const size_t fan = s->cfg.fanout[l];
      const size_t ecap = in_cap * fan;
      bool resolved = false, split = false;
      size_t *d_ne = reinterpret_cast<size_t *>(&out->d_meta->num_edge[l]);
      if (s->cfg.sample_type == FGNN_WEIGHTED_KHOP_PREFIX || s->cfg.sample_type == FGNN_KHOP1 ||
          s->cfg.sample_type == FGNN_WEIGHTED_KHOP)
      {
        // the frontier is a list of unique node ids: seed order by bitmap ranking, no sort (sample_weighted.hip)
        // FGNN_RANK_BITMAP=0 (profiling build, A/B only): order the seeds with scan.hip's sort like the stateless
        // C entry points
        static const bool use_rank = fgnn::tune_int("FGNN_RANK_BITMAP", 1) != 0;
        const fgnn::RankWs rank{use_rank ? sl.rank_bitmap : nullptr, &sl.scan_sample};
        rc = fgnn::sample_with_replacement_ex(
            s->cfg.sample_type, s->cfg.indptr, s->cfg.indices,
            s->cfg.sample_type == FGNN_WEIGHTED_KHOP_PREFIX ? s->cfg.prob_prefix : s->cfg.prob_table, s->cfg.alias_table,
            cur, cur_n_host, d_cur_n, in_cap, fan, out->col[l], tmp_dst, d_ne, FGNN_SRC_LOCAL, s->cfg.seed, batch_key,
            (uint32_t)l, ws, s->ws_bytes, stream, s->cfg.num_node, rank.bitmap ? &rank : nullptr,
            fgnn::prefix_tree_view(s->ptree));
      }
      else if (s->cfg.sample_type == FGNN_WEIGHTED_KHOP_HASH_DEDUP)
        rc = fgnn::sample_hash_dedup(s->cfg.indptr, s->cfg.indices, s->cfg.prob_table, s->cfg.alias_table, cur,
                                     cur_n_host, d_cur_n, in_cap, fan, out->col[l], tmp_dst, d_ne, FGNN_SRC_LOCAL,
                                     s->cfg.seed, batch_key, (uint32_t)l, ws, s->ws_bytes, stream, &sl.scan_sample);
      else if (s->cfg.sample_type == FGNN_RANDOM_WALK)
        // fanout[l] == RunConfig::num_neighbor (CHECK_EQ at cuda_loops.cc:129)
        rc = fgnn::sample_random_walk_ex(s->cfg.indptr, s->cfg.indices, cur, cur_n_host, d_cur_n, in_cap, s->cfg.walk_len,
                                         s->cfg.restart_prob, s->cfg.num_walks, fan, out->col[l], tmp_dst, out->data[l],
                                         d_ne, FGNN_SRC_LOCAL, s->cfg.seed, batch_key, (uint32_t)l, ws, s->ws_bytes,
                                         stream, &sl.scan_sample);
      else {
        // k-hop: the sampler inserts every edge it emits into the dedup table itself (pass 1 of FillWithDuplicates)
        const fgnn::BatchStart start{ht->n2o, out->output_nodes, out->d_meta, batch_key, (uint32_t)L, (uint32_t)l};
        const bool first = start_in_sampler && l == (long)L - 1;
        // khop2's last layer runs as sampler kernel + insert kernel instead of the fused one.  khop2 rewrites CSR rows,
        // so the sampler kernels of consecutive batches form ONE chain however the batches overlap: layer-(L-1) sampler
        // -> its dedup -> ... -> layer-0 sampler -> next batch.  The layer-0 launch is the long one, and half of it is
        // the dedup insert of its edges, which nothing in the chain waits for: split off, the next batch's sampling
        // starts ~25 us earlier (papers100M shape, three batches in flight: 0.133 -> 0.118 ms per batch; one more launch
        // and one re-read of the layer's neighbour list; profiles/r02_split_ab.txt).  Not when this launch also inserts
        // the seeds: their local ids would replace pending edges without a note.
        split = ordered && l == 0 && !first && s->opt_split_l0 != 0;
        if (split)
          rc = fgnn::sample_khop_plain(mutates, s->cfg.indptr, s->cfg.indices, cur, cur_n_host, d_cur_n, in_cap, fan,
                                       out->col[l], tmp_dst, d_ne, s->cfg.seed, batch_key, (uint32_t)l, ws, s->ws_bytes,
                                       stream, &sl.scan_sample);
        else {
          // last fill of the batch: the insert hands its outcome to the dedup pass, which then never touches the table
          resolved = l == 0 && !first && fgnn::hashtable_can_resolve(ht, ecap);
          rc = sample_khop_fused(mutates, s->cfg.indptr, s->cfg.indices, cur, cur_n_host, d_cur_n, in_cap, fan,
                                 out->col[l], tmp_dst, d_ne, s->cfg.seed, batch_key, (uint32_t)l, ht, ws, s->ws_bytes,
                                 stream, &sl.scan_sample, first ? &start : nullptr, resolved);
        }
      }
      if (rc != FGNN_OK) return rc;
      const bool inserted = khop_fused && !split;
      if (l == 0) {
        // last sampler kernel of this batch: the next batch may touch the CSR; the layer's fill belongs to the tail
        if (mutates) guard.pass_csr(true);
        ran = true;
        ecap0 = ecap;
        resolved0 = resolved;
        inserted0 = inserted;
        break;
      }
      rc = fill(l, ecap, inserted, resolved, table_free);
      if (rc != FGNN_OK) return rc;
      in_cap += ecap;
      cur = out->input_nodes;
      d_cur_n = fgnn_hashtable_d_num_items(ht);
      cur_n_host = 0;
    }
    if (mutates) guard.pass_csr(true);  // (a batch without seeds still takes its turn)
    if (!do_tail) {
      sl.pend.active = true;
      sl.pend.seq = seq;
      sl.pend.out = out;
      sl.pend.ran = ran;
      sl.pend.ecap0 = ecap0;
      sl.pend.resolved0 = resolved0;
      sl.pend.inserted0 = inserted0;
      sl.pend.owed = owed;
      sl.pend.st = st;
      sl.last_st = st;
      guard.mark_csr();  // (samplers that do not touch the CSR take their turn here)
      guard.detached = true;
      return launch_status(__func__);
    }
  } else {
    ran = sl.pend.ran;
    ecap0 = sl.pend.ecap0;
    resolved0 = sl.pend.resolved0;
    inserted0 = sl.pend.inserted0;
    owed = sl.pend.owed;
  }
  // ---- tail
  if (ran) {
    rc = fill(0, ecap0, inserted0, resolved0, false);
    if (rc != FGNN_OK) return rc;
  }
  if (owed.mapped) {
    if (owed_fix) *owed_fix = owed;
    else if ((rc = fgnn::hashtable_map_fix(owed, stream)) != FGNN_OK) return rc;
  }
  // Reset (cuda_hashtable.cu:714-723) for the slot's next batch: a generation bump, no memory traffic
  rc = fgnn::hashtable_next_generation(ht, stream, false);
  if (rc != FGNN_OK) return rc;
  guard.close_slot();
  guard.finished = true;
  return launch_status(__func__);
}

int batch_cache_index(fgnn_batch *b, const uint32_t *cache_table, void *stream, const fgnn::FixTail *carry) {
  b->meta_copied = false;
  fgnn::ScanErrorSink sink(&b->d_meta->overflow);
  // num_miss / num_cache are adjacent in the summary: the split kernel writes them in place
  return fgnn::get_miss_cache_index_ex(cache_table, b->input_nodes, 0, &b->d_meta->num_input, b->owner->max_nodes,
                                       b->cidx[0], b->cidx[1], b->cidx[2], b->cidx[3], &b->d_meta->num_miss, b->ws,
                                       b->ws_bytes, stream, b->scan,
                                       reinterpret_cast<unsigned long long *>(&b->d_meta->t_sampled), carry);
}

}  // namespace

extern "C" int fgnn_sampler_sample_ordered(fgnn_sampler *s, uint64_t seq, const uint32_t *d_seeds, size_t num_seeds,
                                           uint64_t batch_key, fgnn_batch *out, void *stream) {
  return sample_impl(s, seq, d_seeds, num_seeds, batch_key, out, stream, nullptr);
}

extern "C" int fgnn_sampler_sample(fgnn_sampler *s, const uint32_t *d_seeds, size_t num_seeds, uint64_t batch_key,
                                   fgnn_batch *out, void *stream) {
  if (!s) return FGNN_EINVAL;
  uint64_t seq;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    seq = s->next_seq++;
  }
  return fgnn_sampler_sample_ordered(s, seq, d_seeds, num_seeds, batch_key, out, stream);
}

// the two halves of a batch as calls of their own (sample_impl): a caller with several batches in flight enqueues
// begin(k + 1) BEFORE end(k), so that khop2's cross-batch sampler chain never waits for the host to get through a
// batch's tail.  end = the tail [+ the cache-index split, which carries the last remap fix-up]; the caller then appends
// extraction / message pack and fgnn_batch_finish on the same stream.
extern "C" int fgnn_sampler_sample_begin(fgnn_sampler *s, const uint32_t *d_seeds, size_t num_seeds, uint64_t batch_key,
                                         fgnn_batch *out, void *stream, uint64_t *seq_out) {
  if (!s || !seq_out) return FGNN_EINVAL;
  uint64_t seq;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    seq = s->next_seq++;
  }
  *seq_out = seq;
  return sample_impl(s, seq, d_seeds, num_seeds, batch_key, out, stream, nullptr, kChain);
}

extern "C" int fgnn_sampler_sample_begin_ordered(fgnn_sampler *s, uint64_t seq, const uint32_t *d_seeds,
                                                 size_t num_seeds, uint64_t batch_key, fgnn_batch *out, void *stream) {
  return sample_impl(s, seq, d_seeds, num_seeds, batch_key, out, stream, nullptr, kChain);
}

extern "C" int fgnn_sampler_sample_end(fgnn_sampler *s, uint64_t seq, fgnn_batch *out, const uint32_t *cache_table,
                                       void *stream) {
  fgnn::FixTail owed = fgnn::no_fix_tail();
  int rc = sample_impl(s, seq, nullptr, 0, 0, out, stream, cache_table ? &owed : nullptr, kTail);
  if (rc == FGNN_OK && cache_table) rc = batch_cache_index(out, cache_table, stream, &owed);
  return rc;
}

// DoGPUSample + DoGetCacheMissIndex (dist_loops_arch5.cc:86-105: what an arch5 sampler does per batch) in one call, with
// the internal sequence counter of fgnn_sampler_sample: the last layer's remap fix-up rides on the split launch
extern "C" int fgnn_sampler_sample_indexed(fgnn_sampler *s, const uint32_t *d_seeds, size_t num_seeds,
                                           uint64_t batch_key, fgnn_batch *out, const uint32_t *cache_table,
                                           void *stream) {
  if (!s) return FGNN_EINVAL;
  uint64_t seq;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    seq = s->next_seq++;
  }
  fgnn::FixTail owed = fgnn::no_fix_tail();
  int rc = sample_impl(s, seq, d_seeds, num_seeds, batch_key, out, stream, cache_table ? &owed : nullptr);
  if (rc == FGNN_OK && cache_table) rc = batch_cache_index(out, cache_table, stream, &owed);
  return rc;
}

// whole batch in one call: sample -> [cache index] -> extract -> summary copy (saves host round trips per step)
extern "C" int fgnn_sampler_run_batch(fgnn_sampler *s, uint64_t seq, const uint32_t *d_seeds, size_t num_seeds,
                                      uint64_t batch_key, fgnn_batch *out, const uint32_t *cache_table,
                                      const void *feat, const void *label, void *stream) {
  fgnn::FixTail owed = fgnn::no_fix_tail();
  // with a cache split to follow, the last layer's remap fix-up rides on that launch
  int rc = sample_impl(s, seq, d_seeds, num_seeds, batch_key, out, stream, cache_table ? &owed : nullptr);
  if (rc == FGNN_OK && cache_table) rc = batch_cache_index(out, cache_table, stream, &owed);
  if (rc == FGNN_OK && (feat || label)) rc = fgnn_batch_extract(out, feat, label, stream);
  if (rc == FGNN_OK) rc = fgnn_batch_finish(out, stream);
  return rc;
}

extern "C" int fgnn_batch_cache_index(fgnn_batch *b, const uint32_t *cache_table, void *stream) {
  if (!b || !cache_table) return FGNN_EINVAL;
  return batch_cache_index(b, cache_table, stream, nullptr);
}

extern "C" int fgnn_batch_enable_timing(fgnn_batch *b, int on) {
  if (!b) return FGNN_EINVAL;
  if (on && !b->t0) {
    FGNN_HIP_CHECK(hipEventCreate(&b->t0));
    FGNN_HIP_CHECK(hipEventCreate(&b->t1));
    FGNN_HIP_CHECK(hipEventCreate(&b->t2));
  }
  b->timing = on != 0;
  b->timed = b->timed2 = false;
  return FGNN_OK;
}

extern "C" float fgnn_batch_gather_ms(fgnn_batch *b) {
  float ms = -1.0f;
  if (b && b->timed && hipEventElapsedTime(&ms, b->t0, b->t1) != hipSuccess) ms = -1.0f;
  return ms;
}

extern "C" float fgnn_batch_gather_kernel_ms(fgnn_batch *b) {
  if (!b || !b->timed || !b->stamped || !b->h_stamps) return -1.0f;
  unsigned long long n = b->h_stamps[2 * fgnn::kGatherStampBlocks];
  if (n == 0) return -1.0f;
  if (n > fgnn::kGatherStampBlocks) n = fgnn::kGatherStampBlocks;
  unsigned long long t0 = ~0ull, t1 = 0;
  for (unsigned long long k = 0; k < n; ++k) {
    t0 = b->h_stamps[2 * k] < t0 ? b->h_stamps[2 * k] : t0;
    t1 = b->h_stamps[2 * k + 1] > t1 ? b->h_stamps[2 * k + 1] : t1;
  }
  return t1 >= t0 ? (float)((double)(t1 - t0) * 1e-5) : -1.0f;
}

extern "C" int fgnn_batch_extract_cached_ms(fgnn_batch *b, float out[2]) {
  if (!b || !out) return FGNN_EINVAL;
  out[0] = out[1] = -1.0f;
  if (!b->timed2) return FGNN_OK;
  if (b->stamp_grid) {
    // one launch: a band's duration = first start .. last end over its workgroups (100 MHz wall clock)
    auto span = [&](size_t lo, size_t hi) -> float {
      unsigned long long t0 = ~0ull, t1 = 0;
      for (size_t k = lo; k < hi; ++k) {
        t0 = b->h_stamps[2 * k] < t0 ? b->h_stamps[2 * k] : t0;
        t1 = b->h_stamps[2 * k + 1] > t1 ? b->h_stamps[2 * k + 1] : t1;
      }
      return hi > lo && t1 >= t0 ? (float)((double)(t1 - t0) * 1e-5) : -1.0f;
    };
    out[0] = span(0, b->stamp_link);
    out[1] = span(b->stamp_link, b->stamp_grid);
    return FGNN_OK;
  }
  if (hipEventElapsedTime(&out[0], b->t0, b->t1) != hipSuccess) out[0] = -1.0f;
  if (hipEventElapsedTime(&out[1], b->t1, b->t2) != hipSuccess) out[1] = -1.0f;
  return FGNN_OK;
}

extern "C" float fgnn_batch_extract_launch_ms(fgnn_batch *b) {
  float ms = -1.0f;
  if (b && b->timed2 && hipEventElapsedTime(&ms, b->t0, b->t2) != hipSuccess) ms = -1.0f;
  return ms;
}

extern "C" int fgnn_batch_set_feat_row_mask(fgnn_batch *b, uint32_t mask) {
  if (!b) return FGNN_EINVAL;
  b->feat_row_mask = mask;
  return FGNN_OK;
}

namespace {

// label rows + summary copy as a tail of the batch's last feature gather (cache_gather.hip), where that gather can carry one
bool make_tail(fgnn_batch *b, const void *src, const void *label, fgnn::GatherTail *t) {
  if (!fgnn::gather_takes_tail(b->feat, src, b->feat_rows_cap, b->feat_dim, b->feat_dtype)) return false;
  *t = fgnn::GatherTail{nullptr, nullptr, nullptr, 0, 0, reinterpret_cast<uint32_t *>(b->h_meta),
                        reinterpret_cast<const uint32_t *>(b->d_meta), (uint32_t)(sizeof(fgnn_batch_meta) / 4), nullptr};
  if (label && b->num_output) {
    t->label_out = b->label;
    t->label_src = label;
    t->label_index = b->output_nodes;
    t->num_label = (uint32_t)b->num_output;
    t->label_esz = (uint32_t)dtype_size(b->label_dtype);
  }
  return true;
}

}  // namespace

extern "C" int fgnn_batch_extract(fgnn_batch *b, const void *feat, const void *label, void *stream) {
  if (!b || !b->feat_dim) return FGNN_EINVAL;
  int rc = FGNN_OK;
  auto st = static_cast<hipStream_t>(stream);
  bool tailed = false;
  if (feat) {
    if (b->feat_rows_cap < b->owner->max_nodes)
      hipLaunchKernelGGL(batch_rows_overflow_kernel, dim3(1), dim3(1), 0, st, b->d_meta, (uint32_t)b->feat_rows_cap);
    fgnn::GatherTail tail;
    tailed = make_tail(b, feat, label, &tail);
    b->stamped = false;
    if (b->timing && tailed) {
      // the launch's own duration from its workgroups' clock words (pinned memory), next to the HIP-event bracket: the
      // events also hold the launch gap and the wait for wave slots behind other batches' kernels
      if (b->stamp_cap < fgnn::kGatherStampBlocks + 1) {
        if (b->h_stamps) (void)hipHostFree(b->h_stamps);
        b->h_stamps = nullptr;
        b->stamp_cap = 0;
        FGNN_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&b->h_stamps),
                                     2 * (fgnn::kGatherStampBlocks + 1) * sizeof(unsigned long long), hipHostMallocDefault));
        b->stamp_cap = fgnn::kGatherStampBlocks + 1;
      }
      b->h_stamps[2 * fgnn::kGatherStampBlocks] = 0;
      tail.stamps = b->h_stamps;
      b->stamped = true;
    }
    if (b->timing) FGNN_HIP_CHECK(hipEventRecord(b->t0, st));
    rc = fgnn::gather_rows_ex(b->feat, feat, b->input_nodes, nullptr, 0, &b->d_meta->num_input, b->feat_rows_cap,
                              b->feat_dim, b->feat_dtype, b->feat_row_mask, stream, tailed ? &tail : nullptr, 0,
                              fgnn::kSharedGpuGatherWgPerCu);
    if (b->timing) {
      FGNN_HIP_CHECK(hipEventRecord(b->t1, st));
      b->timed = true;
    }
    if (rc == FGNN_OK && tailed) b->meta_copied = true;
  }
  if (rc == FGNN_OK && !tailed && label && b->num_output)
    rc = fgnn_gather_rows(b->label, label, b->output_nodes, nullptr, b->num_output, nullptr, b->num_output, 1,
                          b->label_dtype, stream);
  return rc;
}

extern "C" int fgnn_batch_extract_cached(fgnn_batch *b, const void *cache_rows, const void *full_feat,
                                         const void *label, void *stream) {
  if (!b || !b->feat_dim) return FGNN_EINVAL;
  // rows are scattered by destination index: a feature buffer below the worst case cannot be clamped, refuse it
  if (b->feat_rows_cap < b->owner->max_nodes) return FGNN_EINVAL;
  int rc = FGNN_OK;
  auto st = static_cast<hipStream_t>(stream);
  const bool timing = b->timing && full_feat && cache_rows;
  fgnn::GatherTail tail;
  const void *last_src = cache_rows ? cache_rows : full_feat;
  const bool tailed = last_src && make_tail(b, last_src, label, &tail);
  // ONE launch (CombineMissData + CombineCacheData, cuda_cache_manager_device.cu:165-210,339-442,
  // with ExtractMissData's fetch fused in): a band of workgroups pulls the miss rows over the host link while the rest
  // of the grid streams the hit rows out of the HBM cache; labels and the summary ride in the HBM band
  fgnn::ExtractJob job;
  std::memset(static_cast<void *>(&job), 0, sizeof(job));
  job.out = b->feat;
  job.miss_rows = full_feat;
  job.cache_rows = cache_rows;
  job.miss_src = b->cidx[0]; job.miss_dst = b->cidx[1]; job.cache_src = b->cidx[2]; job.cache_dst = b->cidx[3];
  job.d_counts = &b->d_meta->num_miss;  // num_miss, num_cache: adjacent words of the summary
  job.cap = b->feat_rows_cap;
  job.dim = b->feat_dim;
  job.dtype = b->feat_dtype;
  job.miss_mask = b->feat_row_mask;
  if (tailed) job.tail = tail;
  if (full_feat && cache_rows && tailed && fgnn::extract_can_fuse(job)) {
    job.link_wgs = fgnn::pointer_is_host(full_feat) ? FGNN_LINK_WGS_SHARED : 0;  // this GPU samples too
    if (timing) {
      const size_t grid = fgnn::extract_fused_grid(job);
      if (grid > b->stamp_cap) {
        if (b->h_stamps) (void)hipHostFree(b->h_stamps);
        b->h_stamps = nullptr;
        b->stamp_cap = 0;
        // pinned host memory: every workgroup posts its two clock words there, no copy behind the launch
        FGNN_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&b->h_stamps), 2 * grid * sizeof(unsigned long long),
                                     hipHostMallocDefault));
        b->stamp_cap = grid;
      }
      job.stamps = b->h_stamps;
      FGNN_HIP_CHECK(hipEventRecord(b->t0, st));
    }
    size_t grid = 0;
    rc = fgnn::extract_fused(job, stream, &grid);
    if (timing) {
      FGNN_HIP_CHECK(hipEventRecord(b->t2, st));
      b->stamp_grid = grid;
      (void)fgnn::extract_fused_grid(job, &b->stamp_link);
      b->timed2 = true;
    }
    if (rc == FGNN_OK) b->meta_copied = true;
    return rc;
  }
  // rows that are not whole 16-byte chunks, or one source missing: one launch per list
  b->stamp_grid = 0;
  if (timing) FGNN_HIP_CHECK(hipEventRecord(b->t0, st));
  if (full_feat)
    rc = fgnn::gather_rows_ex(b->feat, full_feat, b->cidx[0], b->cidx[1], 0, &b->d_meta->num_miss, b->feat_rows_cap,
                              b->feat_dim, b->feat_dtype, b->feat_row_mask, stream,
                              tailed && !cache_rows ? &tail : nullptr,
                              fgnn::kSharedGpuHostGrid);  // this GPU samples too: a host-source launch stays small
  if (timing) FGNN_HIP_CHECK(hipEventRecord(b->t1, st));
  if (rc == FGNN_OK && cache_rows)  // CombineCacheData
    rc = fgnn::gather_rows_ex(b->feat, cache_rows, b->cidx[2], b->cidx[3], 0, &b->d_meta->num_cache, b->feat_rows_cap,
                              b->feat_dim, b->feat_dtype, 0xFFFFFFFFu, stream, tailed ? &tail : nullptr);
  if (timing) {
    FGNN_HIP_CHECK(hipEventRecord(b->t2, st));
    b->timed2 = true;
  }
  if (rc == FGNN_OK && tailed) b->meta_copied = true;
  if (rc == FGNN_OK && !tailed && label && b->num_output)
    rc = fgnn_gather_rows(b->label, label, b->output_nodes, nullptr, b->num_output, nullptr, b->num_output, 1,
                          b->label_dtype, stream);
  return rc;
}

// sampler + trainer side of one batch on ONE GPU with a feature cache: sample -> cache index -> CombineMissData (rows
// fetched from `full_feat`, typically registered host memory) + CombineCacheData -> finish
extern "C" int fgnn_sampler_run_batch_cached(fgnn_sampler *s, uint64_t seq, const uint32_t *d_seeds, size_t num_seeds,
                                             uint64_t batch_key, fgnn_batch *out, const uint32_t *cache_table,
                                             const void *cache_rows, const void *full_feat, const void *label,
                                             void *stream) {
  if (!cache_table) return FGNN_EINVAL;
  fgnn::FixTail owed = fgnn::no_fix_tail();
  int rc = sample_impl(s, seq, d_seeds, num_seeds, batch_key, out, stream, &owed);
  if (rc == FGNN_OK) rc = batch_cache_index(out, cache_table, stream, &owed);
  if (rc == FGNN_OK) rc = fgnn_batch_extract_cached(out, cache_rows, full_feat, label, stream);
  if (rc == FGNN_OK) rc = fgnn_batch_finish(out, stream);
  return rc;
}

// The batch loop of one GPU that samples and extracts, in native code: what the reference's loop threads do
// (RunSampleCopySubLoopOnce, cuda_loops_arch1.cc:38-84: next batch of the shuffled train set -> DoGPUSample ->
// DoGPUFeatureExtract -> Submit), with the batches rotating over the plan's buffers and streams so that whole batches
// overlap.  One host thread; a buffer is collected (its summary copied out) right before it is reused and at the end.
extern "C" int fgnn_sampler_run_range(fgnn_sampler *s, const fgnn_run_plan *p, uint64_t first_seq, size_t count,
                                      fgnn_batch_meta *h_metas, float *h_gather_ms, double *h_enqueue_s) {
  if (!s || !p || !p->batches || !p->streams || p->num_batches == 0 || p->num_streams == 0 || !p->d_train ||
      p->batch_size == 0 || p->batch_size > s->cfg.max_batch_size || p->num_train == 0 || (count && !h_metas))
    return FGNN_EINVAL;
  if (p->cached && !p->cache_table) return FGNN_EINVAL;
  const size_t steps = (p->num_train + p->batch_size - 1) / p->batch_size;
  double busy = 0.0;
  auto collect = [&](uint64_t i) -> int {
    fgnn_batch *b = p->batches[i % p->num_batches];
    const int rc = fgnn_batch_wait(b, &h_metas[i - first_seq]);
    if (rc != FGNN_OK) return rc;
    if (h_gather_ms) {
      float *t = h_gather_ms + 2 * (i - first_seq);
      t[0] = t[1] = -1.0f;
      if (p->cached) (void)fgnn_batch_extract_cached_ms(b, t);
      else {
        t[0] = fgnn_batch_gather_ms(b);
        t[1] = fgnn_batch_gather_kernel_ms(b);
      }
    }
    return FGNN_OK;
  };
  int rc = FGNN_OK;
  // chain(i) is enqueued BEFORE tail(i - 1) (sample_impl): with whole batches enqueued one after the other the next
  // batch's first sampler launch sits behind ~6 launches of this batch's tail in the host's order, and on a slow host
  // khop2's cross-batch chain waits for the host, not for the GPU.  Needs a second buffer (the tail of batch i - 1 is
  // still owed when batch i starts)
  const bool pipelined = p->num_batches >= 2;
  constexpr uint64_t kNone = ~0ull;
  uint64_t owed_rest = kNone;  // the batch whose chain is enqueued and whose tail is not
  auto rest = [&](uint64_t i) -> int {  // tail + cache split + extraction + finish of batch i
    fgnn_batch *b = p->batches[i % p->num_batches];
    void *st = p->streams[i % p->num_streams];
    int r = fgnn_sampler_sample_end(s, i, b, p->cache_table, st);
    if (r == FGNN_OK && p->cached) r = fgnn_batch_extract_cached(b, p->cache_rows, p->full_feat, p->label, st);
    else if (r == FGNN_OK && (p->feat || p->label)) r = fgnn_batch_extract(b, p->feat, p->label, st);
    if (r == FGNN_OK) r = fgnn_batch_finish(b, st);
    return r;
  };
  for (uint64_t i = first_seq; i < first_seq + count && rc == FGNN_OK; ++i) {
    if (i - first_seq >= p->num_batches && (rc = collect(i - p->num_batches)) != FGNN_OK) break;
    const size_t step = (size_t)(i % steps);
    const size_t b0 = step * p->batch_size;
    const size_t n = p->num_train - b0 < p->batch_size ? p->num_train - b0 : p->batch_size;
    fgnn_batch *b = p->batches[i % p->num_batches];
    void *st = p->streams[i % p->num_streams];
    const auto t0 = std::chrono::steady_clock::now();
    if (pipelined) {
      rc = fgnn_sampler_sample_begin_ordered(s, i, p->d_train + b0, n, step, b, st);
      if (rc == FGNN_OK) {
        if (owed_rest != kNone) rc = rest(owed_rest);
        owed_rest = i;
      }
    } else if (p->cached)
      rc = fgnn_sampler_run_batch_cached(s, i, p->d_train + b0, n, step, b, p->cache_table, p->cache_rows, p->full_feat,
                                         p->label, st);
    else
      rc = fgnn_sampler_run_batch(s, i, p->d_train + b0, n, step, b, p->cache_table, p->feat, p->label, st);
    busy += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  if (owed_rest != kNone) {  // the last batch's tail (also after an error: its slot must be closed)
    const auto t0 = std::chrono::steady_clock::now();
    const int rc2 = rest(owed_rest);
    if (rc == FGNN_OK) rc = rc2;
    busy += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  // drain: every buffer that still holds an uncollected batch (also after an error: the buffers are reusable afterwards)
  const uint64_t enq_end = first_seq + count;
  const uint64_t drain0 = count > p->num_batches ? enq_end - p->num_batches : first_seq;
  for (uint64_t i = drain0; i < enq_end; ++i) {
    const int rc2 = rc == FGNN_OK ? collect(i) : fgnn_batch_wait(p->batches[i % p->num_batches], nullptr);
    if (rc == FGNN_OK) rc = rc2;
  }
  if (h_enqueue_s) *h_enqueue_s = busy;
  return rc;
}

extern "C" int fgnn_batch_finish(fgnn_batch *b, void *stream) {
  if (!b) return FGNN_EINVAL;
  auto st = static_cast<hipStream_t>(stream);
  // unless the batch's last extract launch has written the summary to h_meta itself (GatherTail)
  if (!b->meta_copied)
    FGNN_HIP_CHECK(hipMemcpyAsync(b->h_meta, b->d_meta, sizeof(fgnn_batch_meta), hipMemcpyDeviceToHost, st));
  b->meta_copied = false;
  FGNN_HIP_CHECK(hipEventRecord(b->done, st));
  return FGNN_OK;
}

extern "C" fgnn_batch_meta *fgnn_batch_host_meta(const fgnn_batch *b) { return b ? b->h_meta : nullptr; }
extern "C" int fgnn_batch_meta_copied(fgnn_batch *b) {
  if (!b) return FGNN_EINVAL;
  b->meta_copied = true;
  return FGNN_OK;
}

extern "C" int fgnn_batch_wait(fgnn_batch *b, fgnn_batch_meta *h_meta) {
  if (!b) return FGNN_EINVAL;
  FGNN_HIP_CHECK(hipEventSynchronize(b->done));
  if (h_meta) *h_meta = *b->h_meta;
  return FGNN_OK;
}

extern "C" const uint32_t *fgnn_batch_row(const fgnn_batch *b, int l) {
  return (b && l >= 0 && l < FGNN_MAX_LAYERS) ? b->row[l] : nullptr;
}
extern "C" const uint32_t *fgnn_batch_col(const fgnn_batch *b, int l) {
  return (b && l >= 0 && l < FGNN_MAX_LAYERS) ? b->col[l] : nullptr;
}
extern "C" const uint32_t *fgnn_batch_data(const fgnn_batch *b, int l) {
  return (b && l >= 0 && l < FGNN_MAX_LAYERS) ? b->data[l] : nullptr;
}
extern "C" const uint32_t *fgnn_batch_input_nodes(const fgnn_batch *b) { return b ? b->input_nodes : nullptr; }
extern "C" const uint32_t *fgnn_batch_output_nodes(const fgnn_batch *b) { return b ? b->output_nodes : nullptr; }
extern "C" const void *fgnn_batch_feat(const fgnn_batch *b) { return b ? b->feat : nullptr; }
extern "C" const void *fgnn_batch_label(const fgnn_batch *b) { return b ? b->label : nullptr; }
extern "C" const uint32_t *fgnn_batch_cache_index_ptr(const fgnn_batch *b, int which) {
  return (b && which >= 0 && which < 4) ? b->cidx[which] : nullptr;
}
extern "C" const fgnn_batch_meta *fgnn_batch_device_meta(const fgnn_batch *b) { return b ? b->d_meta : nullptr; }
