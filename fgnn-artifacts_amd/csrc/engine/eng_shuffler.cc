#include "eng_shuffler.h"

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

Shuffler::Shuffler(const uint32_t *train_set, size_t num_data, size_t num_epoch, size_t batch_size, int sampler_id,
                   int num_sampler, hipStream_t stream)
    : data_(train_set, train_set + num_data), num_data_(num_data), num_epoch_(num_epoch), batch_size_(batch_size),
      stream_(stream) {
  SAM_CHECK(batch_size > 0 && num_sampler > 0 && sampler_id >= 0 && sampler_id < num_sampler);
  // drop_last == false path of dist_shuffler.cc:47-79
  size_t total_step = (num_data + batch_size - 1) / batch_size;
  last_batch_size_ = num_data % batch_size == 0 ? batch_size : num_data % batch_size;
  if (sampler_id < num_sampler - 1) last_batch_size_ = batch_size;
  epoch_step_ = total_step;
  dataset_offset_ = (total_step / num_sampler * sampler_id) * batch_size;
  if (sampler_id == num_sampler - 1) {
    const size_t previous = total_step / num_sampler * sampler_id;
    num_step_ = total_step - previous;
    local_size_ = num_data - previous * batch_size;
  } else {
    num_step_ = total_step / num_sampler;
    local_size_ = num_step_ * batch_size;
  }
  cur_step_ = num_step_;
  SAM_HIP(hipMalloc(&d_data_, (local_size_ ? local_size_ : 1) * sizeof(uint32_t)));
}

Shuffler::~Shuffler() {
  if (d_data_) (void)hipFree(d_data_);
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
  ShuffleMinstd0(data_.data(), num_data_, cur_epoch_);  // seed = epoch: every sampler gets the same permutation
  SAM_HIP(hipMemcpyAsync(d_data_, data_.data() + dataset_offset_, local_size_ * sizeof(uint32_t),
                         hipMemcpyHostToDevice, stream_));
  SAM_HIP(hipStreamSynchronize(stream_));
}

bool Shuffler::GetBatch(const uint32_t **d_batch, size_t *size) {
  cur_step_++;
  if (cur_step_ >= num_step_) ReShuffle();
  if (cur_epoch_ >= num_epoch_) return false;
  *d_batch = d_data_ + cur_step_ * batch_size_;
  *size = cur_step_ == num_step_ - 1 ? last_batch_size_ : batch_size_;
  return true;
}

}  // namespace sam
