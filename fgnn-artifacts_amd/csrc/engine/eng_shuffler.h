// eng_shuffler.h -- per-epoch shuffle of the train set and its split into batches.
// Reference: dist/dist_shuffler.cc:36-179 (arch5: every sampler shuffles the WHOLE train set on the
// host with std::default_random_engine(epoch) and takes a contiguous step range),
// cuda/cuda_shuffler.cc:40-154 (single process; wall-clock seed there, epoch seed here),
// dist/dist_shuffler_aligned.cc:36-146 (arch6 / arch7: the train set is padded to a multiple of the worker count with
// its own first ids, every worker takes an equal contiguous share of the shuffled array).
// The permutation is bit-identical to the reference's for the same seed (oracle
// fgnn_oracle_shuffle_minstd0, pinned against libstdc++ in tests/golden/shuffle.npz).
#pragma once
#include <cstdint>
#include <thread>
#include <vector>

#include "eng_common.h"

namespace sam {

// std::minstd_rand0 + libstdc++ uniform_int_distribution<size_t>(0,i): in place, cumulative
void ShuffleMinstd0(uint32_t *data, size_t n, uint64_t seed);

struct ShufflePartition {  // one sampler's share of an epoch
  size_t padded_size, local_size, num_local_step, epoch_step, step_offset, dataset_offset, last_batch_size;
};

class Shuffler {
 public:
  // dist_shuffler.cc:47-79 (contiguous step ranges, the last sampler takes the remainder) or, aligned,
  // dist_shuffler_aligned.cc:45-71 (equal shares of the set padded to a multiple of the sampler count)
  static ShufflePartition Partition(size_t num_data, size_t batch_size, int sampler_id, int num_sampler, bool aligned);
  // sampler_id/num_sampler = 0/1 for the single-process engines
  Shuffler(const uint32_t *train_set, size_t num_data, size_t num_epoch, size_t batch_size, int sampler_id,
           int num_sampler, hipStream_t stream, bool aligned = false);
  // steps per epoch over all workers of the aligned split (known before any worker exists)
  static size_t AlignedNumStep(size_t num_data, size_t batch_size, size_t num_worker);
  ~Shuffler();
  // waits for the helper thread that prepares the next epoch (it touches the tracked streams); GetBatch after this
  // still works, the join in ReShuffle is then a no-op
  void Quiesce() { if (prep_.joinable()) prep_.join(); }
  // next batch of this sampler: device pointer into the epoch's slice + size; false when all epochs are done
  bool GetBatch(const uint32_t **d_batch, size_t *size);
  uint64_t Epoch() const { return cur_epoch_; }
  uint64_t Step() const { return cur_step_ + step_offset_; }  // global step (dist_shuffler.h:38-41, dist_shuffler_aligned.h:41)
  size_t NumStep() const { return epoch_step_; }      // steps per epoch over all samplers
  size_t NumLocalStep() const { return num_step_; }   // steps of this sampler
  bool IsLastBatch() const { return cur_step_ == num_step_ - 1; }
  // SAMGRAPH_SANITY_CHECK (dist_shuffler.cc:139-144,169-176): every batch handed out is checked on the GPU for invalid
  // ids and for ids already handed out in this epoch; a violation is fatal like the reference's device assert
  void EnableSanityCheck(size_t num_node);
  // A stream on which the caller enqueues work that reads the batches handed out (the engine's batch streams).  The
  // helper that prepares epoch e+1 recycles the device array of epoch e-2; before it overwrites that array it waits
  // for everything enqueued on the tracked streams (and on the constructor's stream).  With long epochs those batches
  // finished long ago and the wait returns at once; with a handful of steps per epoch and several batches in flight
  // (tiny train sets, many samplers) batches of epoch e-2 may still be QUEUED when epoch e begins -- without the wait
  // they would read epoch e+1's seeds.  Call before the first GetBatch.
  void TrackStream(hipStream_t st) { readers_.push_back(st); }

 private:
  void ReShuffle();
  // The reference shuffles on the host at every epoch boundary while the GPU waits (dist_shuffler.cc:98-137): 9 ms for
  // papers100M's 1.2 M train ids -- half of an epoch's sampling time on MI355X.  The permutation of epoch e+1 depends
  // only on epoch e's (in-place, cumulative, seed = epoch), so a helper thread prepares it -- host array and its device
  // copy -- while epoch e is being sampled; the boundary then costs a pointer swap.  The host array is double-buffered;
  // the device copy has THREE buffers, so that the helper for epoch e+2 never writes the buffer the last batches of
  // epoch e may still be reading when epoch e+1 begins: no flush of the in-flight batches at the epoch boundary either
  // (the reference drains its pipeline there, dist_loops_arch5.cc:131-137); the helper itself waits for the readers of
  // the array it recycles (TrackStream), off the sampling thread.
  void Prepare(uint64_t epoch);       // starts the helper for `epoch` (from the current host array)
  std::vector<uint32_t> data_, next_;
  uint32_t *d_data_ = nullptr, *d_next_ = nullptr, *d_prev_ = nullptr;
  std::thread prep_;
  hipStream_t copy_stream_ = nullptr;
  std::vector<hipStream_t> readers_;  // see TrackStream
  int device_ = 0;
  size_t num_data_, num_epoch_, batch_size_;
  size_t num_step_, epoch_step_, last_batch_size_, dataset_offset_, local_size_, step_offset_;
  uint64_t cur_epoch_ = 0;
  size_t cur_step_;
  bool initialized_ = false;
  hipStream_t stream_;
  size_t sanity_num_node_ = 0;
  uint32_t *d_sanity_bits_ = nullptr;  // 1 bit per node, cleared at every reshuffle
  uint32_t *d_sanity_flags_ = nullptr;
};

}  // namespace sam
