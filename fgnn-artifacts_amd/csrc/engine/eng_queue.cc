#include "eng_queue.h"

#include "eng_dataset.h"

#include <sys/mman.h>
#include <unistd.h>

#include <cstring>

#include <chrono>
#include <thread>

namespace sam {

MemoryQueue::MemoryQueue(size_t slot_bytes, size_t num_slots) {
  SAM_CHECK(num_slots > 0 && num_slots <= kMaxSlots);
  slot_bytes = (slot_bytes + 255) & ~size_t(255);
  total_bytes_ = sizeof(QueueMeta) + slot_bytes * num_slots;
  // shared by every process of the job: inherited through fork, or a named region (SAMGRAPH_SHM_PREFIX, eng_dataset.cc)
  SharedRegion reg = SharedCreate(total_bytes_);
  meta_ = static_cast<QueueMeta *>(reg.ptr);
  if (!reg.creator) {  // somebody else set the queue up: same geometry, or the job is misconfigured
    SAM_CHECK(meta_->max_size == num_slots && meta_->mq_nbytes == slot_bytes);
    return;
  }
  meta_->send_cnt = 0;
  meta_->recv_cnt = 0;
  meta_->max_size = num_slots;
  meta_->mq_nbytes = slot_bytes;
  for (size_t i = 0; i < num_slots; ++i) {
    meta_->pub_seq[i] = 0;
    meta_->rel_seq[i] = 0;
    meta_->payload_loc[i] = 0;
    meta_->checked[i] = 0;
    meta_->sender[i] = 0;
  }
  memset(meta_->check_verified, 0, sizeof(meta_->check_verified));
  memset(meta_->check_failed, 0, sizeof(meta_->check_failed));
  memset(meta_->rings, 0, sizeof(meta_->rings));
  meta_->ipc_broken = 0;
  SharedPublish(meta_);
}

bool MemoryQueue::CreateDeviceRing(int ring, uint32_t slots) {
  SAM_CHECK(ring >= 0 && ring < kMaxRings && !owns_ring_[ring]);
  if (slots == 0) return false;
  if (slots > (uint32_t)kMaxRingSlots) slots = kMaxRingSlots;
  if (slots > meta_->max_size) slots = (uint32_t)meta_->max_size;  // never more messages in flight than queue slots
  RingInfo &r = meta_->rings[ring];
  uint32_t made = 0;
  for (; made < slots; ++made) {
    void *p = nullptr;
    if (hipMalloc(&p, meta_->mq_nbytes) != hipSuccess) {
      (void)hipGetLastError();
      SAM_LOG(kWarning) << "device ring: HBM for " << made << " of " << slots << " slots only";
      break;
    }
    if (hipIpcGetMemHandle(&r.handle[made], p) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(p);
      SAM_LOG(kWarning) << "device ring: hipIpcGetMemHandle refused; messages use the host ring";
      for (uint32_t i = 0; i < made; ++i) {
        (void)hipFree(local_slot_[ring][i]);
        local_slot_[ring][i] = nullptr;
      }
      return false;
    }
    local_slot_[ring][made] = p;
  }
  if (made == 0) return false;
  slots = made;
  owns_ring_[ring] = true;
  int dev = 0;
  SAM_HIP(hipGetDevice(&dev));
  r.device = dev;
  r.pid = (int)getpid();
  r.slots = slots;
  for (auto &b : r.busy) b = 0;
  for (auto &b : r.spill) b = 0;
  r.sent_device = r.sent_host = r.spilled = 0;
  __sync_synchronize();
  r.ready = 1;
  svc_stop_ = false;
  svc_ = std::thread([this, ring, dev] {
    (void)hipSetDevice(dev);
    while (!svc_stop_) {
      // requests only ever appear after a receiver has flagged the mapping as broken
      if (*(volatile int *)&meta_->ipc_broken) {
        ServiceSpills(ring);
        std::this_thread::sleep_for(std::chrono::microseconds(100));
      } else {
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
      }
    }
  });
  return true;
}

void MemoryQueue::ServiceSpills(int ring) {
  RingInfo &r = meta_->rings[ring];
  for (uint32_t i = 0; i < r.slots; ++i) {
    if (__atomic_load_n(&r.spill[i], __ATOMIC_ACQUIRE) != 1) continue;
    // the device slot holds the complete message (headers included): overwrite the host slot with it
    char *host = meta_->data + (r.slot_key[i] % meta_->max_size) * meta_->mq_nbytes;
    SAM_HIP(hipMemcpy(host, local_slot_[ring][i], meta_->mq_nbytes, hipMemcpyDeviceToHost));
    ++r.spilled;
    __atomic_store_n(&r.spill[i], 2u, __ATOMIC_RELEASE);
  }
}

void *MemoryQueue::ClaimDeviceSlot(int ring, size_t key) {
  if (ring < 0 || ring >= kMaxRings) return nullptr;
  RingInfo &r = meta_->rings[ring];
  if (!owns_ring_[ring]) {  // no HBM ring (SAMGRAPH_DEVICE_RING_SLOTS=0, or refused): counted all the same
    meta_->payload_loc[key % meta_->max_size] = 0;
    ++r.sent_host;
    return nullptr;
  }
  for (uint32_t i = 0; i < r.slots && !*(volatile int *)&meta_->ipc_broken; ++i) {
    if (__atomic_load_n(&r.busy[i], __ATOMIC_ACQUIRE) == 0) {  // single claimer per ring: no CAS needed
      __atomic_store_n(&r.busy[i], 1u, __ATOMIC_RELAXED);
      r.slot_key[i] = key;
      meta_->payload_loc[key % meta_->max_size] = ((uint32_t)(ring + 1) << 8) | i;
      ++r.sent_device;
      return local_slot_[ring][i];
    }
  }
  meta_->payload_loc[key % meta_->max_size] = 0;
  ++r.sent_host;
  return nullptr;
}

const void *MemoryQueue::Payload(size_t key, const void *host_msg, bool *on_device) {
  const uint32_t loc = meta_->payload_loc[key % meta_->max_size];
  *on_device = loc != 0;
  if (!loc) return host_msg;
  const int ring = (int)(loc >> 8) - 1;
  const uint32_t slot = loc & 0xffu;
  SAM_CHECK(ring >= 0 && ring < kMaxRings);
  RingInfo &r = meta_->rings[ring];
  SAM_CHECK(r.ready && slot < r.slots);
  auto note = [&](int state) {
    if (map_state_[ring] < state) {
      int dev = -1;
      (void)hipGetDevice(&dev);
      map_device_[ring] = dev;
      map_state_[ring] = state;
    }
  };
  if (r.pid == (int)getpid()) {
    SAM_CHECK(local_slot_[ring][slot]);
    note(1);
    return local_slot_[ring][slot];
  }
  if (!mapped_slot_[ring][slot] && !*(volatile int *)&meta_->ipc_broken) {
    // SAMGRAPH_DEVICE_RING_FORCE_SPILL=1 (tests): behave as if the mapping had been refused
    const char *force = getenv("SAMGRAPH_DEVICE_RING_FORCE_SPILL");
    hipError_t e = (force && atoi(force)) ? hipErrorInvalidValue
                                          : hipIpcOpenMemHandle(&mapped_slot_[ring][slot], r.handle[slot],
                                                                hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      mapped_slot_[ring][slot] = nullptr;
      SAM_LOG(kWarning) << "device ring " << ring << ": cannot map the sampler's HBM ring (" << hipGetErrorString(e)
                        << "); messages go through the host ring from now on";
      __atomic_store_n(&meta_->ipc_broken, 1, __ATOMIC_RELEASE);
    }
  }
  if (!mapped_slot_[ring][slot]) {
    // ask the owner to copy this message into its host slot, then read it there
    __atomic_store_n(&r.spill[slot], 1u, __ATOMIC_RELEASE);
    Timer t;
    while (__atomic_load_n(&r.spill[slot], __ATOMIC_ACQUIRE) != 2) {
      SAM_CHECK(t.Passed() < 120.0) << "device ring " << ring << ": the sampler does not answer the spill request";
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    *on_device = false;
    note(3);
    return host_msg;
  }
  note(2);
  return mapped_slot_[ring][slot];
}

bool MemoryQueue::RingMapping(int ring, int64_t out[3]) const {
  if (ring < 0 || ring >= kMaxRings) return false;
  const RingInfo &r = meta_->rings[ring];
  out[0] = map_state_[ring];
  out[1] = r.ready ? r.device : -1;
  out[2] = map_state_[ring] ? map_device_[ring] : -1;
  return true;
}

void MemoryQueue::DrainDeviceRing(int ring, double timeout_s) {
  if (ring < 0 || ring >= kMaxRings || !owns_ring_[ring]) return;
  RingInfo &r = meta_->rings[ring];
  Timer t;
  for (;;) {
    bool busy = false;
    for (uint32_t i = 0; i < r.slots; ++i) busy |= __atomic_load_n(&r.busy[i], __ATOMIC_ACQUIRE) != 0;
    if (!busy) break;
    if (t.Passed() >= timeout_s) {
      if (timeout_s > 0)
        SAM_LOG(kWarning) << "device ring " << ring << ": messages still unread after " << timeout_s << " s";
      break;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  svc_stop_ = true;
  if (svc_.joinable()) svc_.join();
  SAM_LOG(kInfo) << "device ring " << ring << ": " << r.sent_device << " messages through HBM, " << r.sent_host
                 << " through the host ring, " << r.spilled << " copied back on request";
  for (uint32_t i = 0; i < r.slots; ++i) {
    (void)hipFree(local_slot_[ring][i]);
    local_slot_[ring][i] = nullptr;
  }
  owns_ring_[ring] = false;
}

bool MemoryQueue::RingStats(int ring, uint64_t out[6]) const {
  if (ring < 0 || ring >= kMaxRings) return false;
  const RingInfo &r = meta_->rings[ring];
  out[0] = r.ready ? r.slots : 0;
  out[1] = r.sent_device;
  out[2] = r.sent_host;
  out[3] = r.spilled;
  out[4] = meta_->check_verified[ring];
  out[5] = meta_->check_failed[ring];
  return true;
}

void MemoryQueue::PinMemory() {
  SAM_HIP(hipHostRegister(meta_, total_bytes_, hipHostRegisterPortable | hipHostRegisterMapped));
  void *d = nullptr;
  SAM_HIP(hipHostGetDevicePointer(&d, meta_, 0));
  dev_base_ = static_cast<const char *>(d);
}

// Hand-shakes by per-slot SEQUENCE numbers, not by the reference's per-slot semaphore pairs (memory_queue.cc:104-138):
// a semaphore cannot tell message k from message k + N of the same slot.  With more blocked receivers (or more
// in-flight slots per receiver, StartExtract keeps four) than slots, the sender of k + N can take the release post
// meant for the sender of k, and the receiver of k + N the data post meant for the receiver of k.  pub_seq[slot] ==
// k + 1 <=> message k is published in the slot; rel_seq[slot] == g <=> generations < g of the slot are released, i.e.
// message k may be written when rel_seq == k / N.  Waiters spin briefly, then sleep in short steps (the reference's
// own wait loops poll at 1 us).
namespace {
// spin ~a few microseconds, then sleep in steps that grow from 1 us to 16 us: a hand-off that actually waits (small
// queues, a slower peer) is picked up within a few microseconds of a 80-120 us batch, a long wait costs no core
// false: this process is closing the queue (MemoryQueue::Close) and the condition has not come true
template <typename Pred>
bool WaitFor(Pred ready, const int *aborted, const std::atomic<bool> &closing) {
  for (int i = 0; i < 4000; ++i) {
    if (ready()) return true;
    __builtin_ia32_pause();
  }
  int us = 1;
  while (!ready()) {
    if (__atomic_load_n(aborted, __ATOMIC_ACQUIRE)) SAM_FATAL << "message queue aborted: a process of the job has died";
    if (closing.load(std::memory_order_acquire)) return false;
    std::this_thread::sleep_for(std::chrono::microseconds(us));
    if (us < 16) us *= 2;
  }
  return true;
}
}  // namespace

void *MemoryQueue::GetPtr(size_t *key) {
  const size_t k = __atomic_fetch_add(&meta_->send_cnt, 1, __ATOMIC_ACQ_REL);
  const size_t slot = k % meta_->max_size, gen = k / meta_->max_size;
  // (a closing process never publishes the slot it claimed: nobody of this process will wait for it either)
  if (!WaitFor([&] { return __atomic_load_n(&meta_->rel_seq[slot], __ATOMIC_ACQUIRE) == gen; }, &meta_->aborted, closing_))
    return nullptr;
  *key = k;
  return meta_->data + slot * meta_->mq_nbytes;
}

void MemoryQueue::SimpleSend(size_t key) {
  __atomic_store_n(&meta_->pub_seq[key % meta_->max_size], key + 1, __ATOMIC_RELEASE);
}

const void *MemoryQueue::Recv(size_t *key) {
  // Blocks until the oldest message is PUBLISHED, then claims it (TryRecv): a receiver never holds a claim on a message
  // it is still waiting for, so one that gives up (Close) leaves nothing behind -- a claimed and never released slot
  // would block the sender that wraps around to it, in whatever process that sender lives.  (The reference's receivers
  // pre-claim their key and wait for it, memory_queue.cc:104-138; any trainer takes any message either way.)
  const void *data = nullptr;
  if (!WaitFor([&] { return TryRecv(&data, key); }, &meta_->aborted, closing_)) return nullptr;
  return data;
}

bool MemoryQueue::TryRecv(const void **data, size_t *key) {
  // Claims the oldest message only if it has been PUBLISHED.  send_cnt counts claimed slots (GetPtr), not published
  // ones, so "recv_cnt < send_cnt" alone would send the caller into a wait for a message whose sender may itself be
  // waiting for a slot the caller still holds.
  for (;;) {
    const size_t k = __atomic_load_n(&meta_->recv_cnt, __ATOMIC_ACQUIRE);
    if (k >= __atomic_load_n(&meta_->send_cnt, __ATOMIC_ACQUIRE)) return false;
    if (__atomic_load_n(&meta_->pub_seq[k % meta_->max_size], __ATOMIC_ACQUIRE) != k + 1) return false;
    size_t expect = k;
    if (!__atomic_compare_exchange_n(&meta_->recv_cnt, &expect, k + 1, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE))
      continue;  // another receiver took k: look at the next one
    *key = k;
    *data = meta_->data + (k % meta_->max_size) * meta_->mq_nbytes;
    return true;
  }
}

void MemoryQueue::Release(size_t key) {
  uint32_t &loc = meta_->payload_loc[key % meta_->max_size];
  if (loc) {  // hand the device slot back to its sampler
    RingInfo &r = meta_->rings[(loc >> 8) - 1];
    __atomic_store_n(&r.spill[loc & 0xffu], 0u, __ATOMIC_RELAXED);
    __atomic_store_n(&r.busy[loc & 0xffu], 0u, __ATOMIC_RELEASE);
    loc = 0;
  }
  __atomic_store_n(&meta_->rel_seq[key % meta_->max_size], key / meta_->max_size + 1, __ATOMIC_RELEASE);
}

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
