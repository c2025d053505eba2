#include "eng_shuffler.h"

#include <utility>

namespace sam {

void ShuffleMinstd0(uint32_t *data, size_t n, uint64_t seed) {
  uint64_t x = seed % 2147483647ull;
  if (x == 0) x = 1;
  const uint64_t urngrange = 2147483646ull - 1ull;
  for (size_t i = n ? n - 1 : 0; i > 0; --i) {
    const uint64_t uerange = (uint64_t)i + 1;
    SAM_CHECK(urngrange >= uerange) << "train set too large for minstd_rand0 downscaling";
    uint64_t ret;
    if (urngrange > (uint64_t)i) {
      const uint64_t scaling = urngrange / uerange;
      const uint64_t past = uerange * scaling;
      do {
        x = (x * 16807ull) % 2147483647ull;
        ret = x - 1;
      } while (ret >= past);
      ret /= scaling;
    } else {
      x = (x * 16807ull) % 2147483647ull;
      ret = x - 1;
    }
    const uint32_t t = data[i];
    data[i] = data[ret];
    data[ret] = t;
  }
}

size_t Shuffler::AlignedNumStep(size_t num_data, size_t batch_size, size_t num_worker) {
  return Partition(num_data, batch_size, 0, (int)num_worker, true).epoch_step;
}

ShufflePartition Shuffler::Partition(size_t num_data, size_t batch_size, int sampler_id, int num_sampler,
                                     bool aligned) {
  SAM_CHECK(batch_size > 0 && num_sampler > 0 && sampler_id >= 0 && sampler_id < num_sampler);
  ShufflePartition p;
  const size_t ns = (size_t)num_sampler, id = (size_t)sampler_id;
  if (aligned) {
    p.padded_size = RoundUpDiv(num_data, ns) * ns;
    p.local_size = p.padded_size / ns;
    p.num_local_step = RoundUpDiv(p.local_size, batch_size);
    p.epoch_step = p.num_local_step * ns;
    p.step_offset = p.num_local_step * id;
    p.dataset_offset = p.local_size * id;
    p.last_batch_size = p.local_size % batch_size == 0 ? batch_size : p.local_size % batch_size;
  } else {
    const size_t total_step = RoundUpDiv(num_data, batch_size);  // drop_last == false
    p.padded_size = num_data;
    p.epoch_step = total_step;
    p.step_offset = total_step / ns * id;
    p.dataset_offset = p.step_offset * batch_size;
    p.last_batch_size = num_data % batch_size == 0 ? batch_size : num_data % batch_size;
    if (id + 1 < ns) {
      p.last_batch_size = batch_size;
      p.num_local_step = total_step / ns;
      p.local_size = p.num_local_step * batch_size;
    } else {
      p.num_local_step = total_step - p.step_offset;
      p.local_size = num_data - p.step_offset * batch_size;
    }
  }
  return p;
}

Shuffler::Shuffler(const uint32_t *train_set, size_t num_data, size_t num_epoch, size_t batch_size, int sampler_id,
                   int num_sampler, hipStream_t stream, bool aligned)
    : data_(train_set, train_set + num_data), num_data_(num_data), num_epoch_(num_epoch), batch_size_(batch_size),
      stream_(stream) {
  const ShufflePartition p = Partition(num_data, batch_size, sampler_id, num_sampler, aligned);
  // aligned: pad with the first ids of the set (dist_shuffler_aligned.cc:50-59)
  SAM_CHECK(p.padded_size - num_data <= num_data);
  for (size_t i = 0; i < p.padded_size - num_data; ++i) data_.push_back(train_set[i]);
  num_data_ = p.padded_size;
  local_size_ = p.local_size;
  num_step_ = p.num_local_step;
  epoch_step_ = p.epoch_step;
  step_offset_ = p.step_offset;
  dataset_offset_ = p.dataset_offset;
  last_batch_size_ = p.last_batch_size;
  cur_step_ = num_step_;
  SAM_HIP(hipGetDevice(&device_));
  SAM_HIP(hipMalloc(&d_data_, (local_size_ ? local_size_ : 1) * sizeof(uint32_t)));
  SAM_HIP(hipMalloc(&d_next_, (local_size_ ? local_size_ : 1) * sizeof(uint32_t)));
  SAM_HIP(hipMalloc(&d_prev_, (local_size_ ? local_size_ : 1) * sizeof(uint32_t)));
  SAM_HIP(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));  // the helper's own: no null-stream semantics
  if (num_epoch_ > 0) Prepare(0);  // epoch 0's permutation is ready by the time the first batch is asked for
}

Shuffler::~Shuffler() {
  if (prep_.joinable()) prep_.join();
  if (d_data_) (void)hipFree(d_data_);
  if (d_next_) (void)hipFree(d_next_);
  if (d_prev_) (void)hipFree(d_prev_);
  if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
  if (d_sanity_bits_) (void)hipFree(d_sanity_bits_);
  if (d_sanity_flags_) (void)hipFree(d_sanity_flags_);
}

void Shuffler::EnableSanityCheck(size_t num_node) {
  if (d_sanity_bits_ || num_node == 0) return;
  sanity_num_node_ = num_node;
  SAM_HIP(hipMalloc(&d_sanity_bits_, fgnn_sanity_map_bytes(num_node)));
  SAM_HIP(hipMalloc(&d_sanity_flags_, sizeof(uint32_t)));
  SAM_HIP(hipMemset(d_sanity_bits_, 0, fgnn_sanity_map_bytes(num_node)));
  SAM_HIP(hipMemset(d_sanity_flags_, 0, sizeof(uint32_t)));
}

void Shuffler::Prepare(uint64_t epoch) {
  prep_ = std::thread([this, epoch] {
    next_ = data_;                                           // cumulative: epoch e+1 shuffles epoch e's array
    ShuffleMinstd0(next_.data(), num_data_, epoch);          // seed = epoch: every sampler gets the same permutation
    SAM_HIP(hipSetDevice(device_));
    // d_next_ held epoch - 3's slice (fresh memory for the first three epochs).  Its batches were all enqueued before
    // epoch - 2 began, so everything that can still read it is already on the readers' streams: wait for that, here on
    // the helper thread.  (Usually a no-op: a whole epoch has been sampled since.)
    if (epoch >= 3) {
      SAM_HIP(hipStreamSynchronize(stream_));
      for (hipStream_t st : readers_) SAM_HIP(hipStreamSynchronize(st));
    }
    SAM_HIP(hipMemcpyAsync(d_next_, next_.data() + dataset_offset_, local_size_ * sizeof(uint32_t),
                           hipMemcpyHostToDevice, copy_stream_));
    SAM_HIP(hipStreamSynchronize(copy_stream_));
  });
}

void Shuffler::ReShuffle() {
  if (!initialized_) {
    cur_epoch_ = 0;
    initialized_ = true;
  } else {
    cur_epoch_++;
  }
  cur_step_ = 0;
  if (cur_epoch_ >= num_epoch_) return;
  if (prep_.joinable()) prep_.join();  // normally long done (already joined after Quiesce)
  data_.swap(next_);
  uint32_t *oldest = d_prev_;
  d_prev_ = d_data_;  // its last batches may still be in flight
  d_data_ = d_next_;
  d_next_ = oldest;
  if (d_sanity_bits_) {  // a new epoch may hand every id out again (dist_shuffler.cc:139-144)
    SAM_HIP(hipMemsetAsync(d_sanity_bits_, 0, fgnn_sanity_map_bytes(sanity_num_node_), stream_));
    SAM_HIP(hipStreamSynchronize(stream_));
  }
  if (cur_epoch_ + 1 < num_epoch_) Prepare(cur_epoch_ + 1);
}

bool Shuffler::GetBatch(const uint32_t **d_batch, size_t *size) {
  cur_step_++;
  if (cur_step_ >= num_step_) ReShuffle();
  if (cur_epoch_ >= num_epoch_) return false;
  *d_batch = d_data_ + cur_step_ * batch_size_;
  *size = cur_step_ == num_step_ - 1 ? last_batch_size_ : batch_size_;
  if (d_sanity_bits_) {
    uint32_t flags = 0;
    SAM_CHECK(fgnn_sanity_check_batch(d_sanity_bits_, sanity_num_node_, *d_batch, *size, 0xFFFFFFFFu, d_sanity_flags_,
                                      stream_) == FGNN_OK);
    SAM_HIP(hipMemcpyAsync(&flags, d_sanity_flags_, sizeof(flags), hipMemcpyDeviceToHost, stream_));
    SAM_HIP(hipStreamSynchronize(stream_));
    SAM_CHECK((flags & 1u) == 0) << "sanity check: invalid id in the batch of epoch " << cur_epoch_ << " step " << Step();
    SAM_CHECK((flags & 4u) == 0) << "sanity check: id beyond num_node in the batch of epoch " << cur_epoch_ << " step "
                                 << Step();
    SAM_CHECK((flags & 2u) == 0) << "duplicate batch input (epoch " << cur_epoch_ << " step " << Step() << ")";
  }
  return true;
}

}  // namespace sam
