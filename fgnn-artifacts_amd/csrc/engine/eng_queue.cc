#include "eng_queue.h"

#include <sys/mman.h>

#include <chrono>
#include <thread>

namespace sam {

MemoryQueue::MemoryQueue(size_t slot_bytes, size_t num_slots) {
  SAM_CHECK(num_slots > 0 && num_slots <= kMaxSlots);
  slot_bytes = (slot_bytes + 255) & ~size_t(255);
  total_bytes_ = sizeof(QueueMeta) + slot_bytes * num_slots;
  void *p = mmap(nullptr, total_bytes_, PROT_READ | PROT_WRITE, MAP_ANONYMOUS | MAP_SHARED, -1, 0);
  SAM_CHECK(p != MAP_FAILED) << "cannot map " << total_bytes_ << " bytes for the memory queue";
  meta_ = static_cast<QueueMeta *>(p);
  meta_->send_cnt = 0;
  meta_->recv_cnt = 0;
  meta_->max_size = num_slots;
  meta_->mq_nbytes = slot_bytes;
  for (size_t i = 0; i < num_slots; ++i) {
    sem_init(meta_->sem_list + i, 1, 0);
    sem_init(meta_->release_list + i, 1, 1);
  }
}

void MemoryQueue::PinMemory() {
  SAM_HIP(hipHostRegister(meta_, total_bytes_, hipHostRegisterPortable | hipHostRegisterMapped));
}

void *MemoryQueue::GetPtr(size_t *key) {
  const size_t k = __sync_fetch_and_add(&meta_->send_cnt, 1);
  while (k >= *(volatile size_t *)&meta_->recv_cnt + meta_->max_size)
    std::this_thread::sleep_for(std::chrono::microseconds(1));
  SAM_CHECK(sem_wait(meta_->release_list + (k % meta_->max_size)) == 0);
  *key = k;
  return meta_->data + (k % meta_->max_size) * meta_->mq_nbytes;
}

void MemoryQueue::SimpleSend(size_t key) { SAM_CHECK(sem_post(meta_->sem_list + (key % meta_->max_size)) == 0); }

const void *MemoryQueue::Recv(size_t *key) {
  while (*(volatile size_t *)&meta_->recv_cnt == *(volatile size_t *)&meta_->send_cnt)
    std::this_thread::sleep_for(std::chrono::microseconds(1));
  const size_t k = __sync_fetch_and_add(&meta_->recv_cnt, 1);
  SAM_CHECK(sem_wait(meta_->sem_list + (k % meta_->max_size)) == 0);
  *key = k;
  return meta_->data + (k % meta_->max_size) * meta_->mq_nbytes;
}

bool MemoryQueue::TryRecv(const void **data, size_t *key) {
  if (*(volatile size_t *)&meta_->recv_cnt == *(volatile size_t *)&meta_->send_cnt) return false;
  *data = Recv(key);
  return true;
}

void MemoryQueue::Release(size_t key) { SAM_CHECK(sem_post(meta_->release_list + (key % meta_->max_size)) == 0); }

size_t MaxMessageBytes(size_t batch_size, const size_t *fanout, size_t num_layers, bool have_data) {
  size_t layer_cnt = batch_size, ret = sizeof(TransData);
  for (long l = (long)num_layers - 1; l >= 0; --l) {
    ret += sizeof(GraphData) + layer_cnt * fanout[l] * (have_data ? 3 : 2) * sizeof(uint32_t);
    layer_cnt += layer_cnt * fanout[l];
  }
  ret += batch_size * sizeof(uint32_t);      // output nodes
  ret += 3 * layer_cnt * sizeof(uint32_t);   // input nodes + (miss|cache) src/dst index pairs
  return ret + 64;
}

}  // namespace sam
