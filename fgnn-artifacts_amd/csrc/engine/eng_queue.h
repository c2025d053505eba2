// eng_queue.h -- sampler -> trainer hand-off: a lock-free MPMC ring of fixed-size slots in a
// MAP_SHARED|MAP_ANONYMOUS host mapping, hipHostRegister'ed by every child after fork, with the
// reference's send / receive counters (memory_queue.h:46-115, memory_queue.cc:33-138; per-slot sequence numbers where
// the reference has semaphore pairs, see eng_queue.cc)
// and the reference's wire format (task_queue.cc:68-88: TransData 40 B header, GraphData 24 B).
//
// MI355X change: the sampler does not issue one D2H copy per tensor with sizes known on the host
// (task_queue.cc:154-227); ONE kernel (pack.hip) reads the sizes on the device and writes the whole
// serialized message into the mapped slot, so the sampler never waits for a size.
#pragma once
#include <atomic>
#include <thread>

#include <cstddef>
#include <cstdint>

#include "eng_common.h"

namespace sam {

struct TransData {      // task_queue.cc:68-81
  bool have_data;
  int num_layer;
  uint64_t key;
  size_t input_size;
  size_t output_size;
  size_t num_miss;
  uint32_t data[0];
};
struct GraphData {      // task_queue.cc:83-88
  size_t num_src, num_dst, num_edge;
  uint32_t data[0];
};
static_assert(sizeof(TransData) == 40, "wire format");
static_assert(sizeof(GraphData) == 24, "wire format");

constexpr size_t kMaxSlots = 170;  // mq_size, memory_queue.h:46
constexpr int kMaxRings = 16;      // sampler processes that may own a device ring
constexpr int kMaxRingSlots = (int)kMaxSlots;  // slot index travels in 8 bits (payload_loc)

// Optional HBM hand-off: each sampler owns a small ring of message slots in ITS OWN HBM and exports it
// with hipIpcGetMemHandle; the payload arrays of a message are packed there, only the headers go to the host slot.
// A trainer maps the ring once and pulls the arrays device-to-device (over xGMI when it sits on another GPU) instead
// of the reference's D2H + H2D pair over the host link.  MPMC semantics are unchanged: any trainer takes any message,
// the host ring still orders and counts them, and a message whose sampler finds no free device slot simply travels
// through the host slot as before.
struct RingInfo {
  int ready;                 // 1 once the handle below is valid
  int device;
  int pid;                   // owner process: the same process uses the pointer directly (IPC cannot self-open)
  uint32_t slots;
  // one allocation and one handle PER SLOT: mapping a single multi-GB allocation through hipIpcOpenMemHandle left the
  // receiver hanging on this ROCm (80 slots x 46 MB in one piece; 16 x 46 MB worked), slot-sized pieces do not
  hipIpcMemHandle_t handle[kMaxRingSlots];
  uint32_t busy[kMaxRingSlots];  // 1 while a published message lives in the slot
  size_t sent_device, sent_host;  // messages of this sampler by payload location
  // fallback when a receiver cannot map the ring (hipIpcOpenMemHandle refused): it asks the owner to copy the slot
  // into the message's host slot -- 0 none, 1 requested, 2 done; the owner's service thread answers
  uint32_t spill[kMaxRingSlots];
  size_t slot_key[kMaxRingSlots];  // queue key of the message in the slot
  size_t spilled;
};

struct QueueMeta {                 // MQ_MetaData, memory_queue.h:65-115
  size_t send_cnt, recv_cnt, max_size, mq_nbytes;
  size_t pub_seq[kMaxSlots];        // key + 1 of the message last PUBLISHED in the slot, 0: none yet
  size_t rel_seq[kMaxSlots];        // generations of the slot released so far: message k may be written at k / N
  uint32_t payload_loc[kMaxSlots];  // 0: payload in the host slot; else ((ring + 1) << 8) | device slot
  int ipc_broken;                   // set by the first receiver that could not map a ring: samplers stop using theirs
  int aborted;                      // a process of the job died (MemoryQueue::Abort): blocked peers give up loudly
  // SAMGRAPH_HANDOFF_CHECK: a message whose sender appended a checksum of its words behind it (pack.hip); the receiver
  // recomputes it THROUGH THE ADDRESS IT READS THE PAYLOAD FROM (mapped HBM slot, host slot) before it uses the batch
  uint32_t checked[kMaxSlots];      // 1: the message in the slot carries the trailer; + which sampler sent it
  uint32_t sender[kMaxSlots];
  size_t check_verified[kMaxRings], check_failed[kMaxRings];  // per sending sampler, counted by the receivers
  RingInfo rings[kMaxRings];
  alignas(256) char data[0];
};

class MemoryQueue {
 public:
  MemoryQueue(size_t slot_bytes, size_t num_slots);  // in the parent, before fork
  void PinMemory();                                  // in each child (memory_queue.cc:47-49)
  void *GetPtr(size_t *key);                         // claim a slot for writing (blocks while the ring is full)
  void SimpleSend(size_t key);                       // publish
  const void *Recv(size_t *key);                     // blocks until a message is available
  // THIS process is shutting down while its own threads may still be blocked on the queue (an in-process engine stopped
  // before its last batch): their GetPtr / Recv return nullptr instead of waiting on.  The reference's loops poll
  // ShouldShutdown at 1 us (cuda_loops_arch3.cc:178-196); other processes of the job are not affected.
  void Close() { closing_.store(true, std::memory_order_release); }
  bool TryRecv(const void **data, size_t *key);      // never blocks: takes the oldest message only if it is PUBLISHED
  void Release(size_t key);                          // SharedData::~SharedData
  // a process of the job has died: every blocked GetPtr / Recv of every process logs and aborts instead of waiting for
  // a message (or a release) that will never come -- the reference's semaphore waits hang in that case
  void Abort() { __atomic_store_n(&meta_->aborted, 1, __ATOMIC_RELEASE); }
  // hand-off check bookkeeping (see QueueMeta)
  void MarkChecked(size_t key, int sender, bool on) {
    meta_->checked[key % meta_->max_size] = on ? 1u : 0u;
    meta_->sender[key % meta_->max_size] = (uint32_t)sender;
  }
  bool IsChecked(size_t key) const { return meta_->checked[key % meta_->max_size] != 0; }
  void CountCheck(size_t key, bool ok) {
    const uint32_t sdr = meta_->sender[key % meta_->max_size] % kMaxRings;
    __atomic_fetch_add(ok ? &meta_->check_verified[sdr] : &meta_->check_failed[sdr], 1, __ATOMIC_RELAXED);
  }
  // out[6] as samgraph_ext_queue_stats (include/samgraph_ext.h); false: no such ring
  bool RingStats(int ring, uint64_t out[6]) const;
  // out[3] as samgraph_ext_ring_mapping: how THIS process reads that ring's payloads
  bool RingMapping(int ring, int64_t out[3]) const;
  size_t SlotBytes() const { return meta_->mq_nbytes; }
  size_t NumSlots() const { return meta_->max_size; }
  // slots claimed for sending and not yet claimed by a receiver -- NOT the number of receivable messages (a claimed
  // slot may still be waiting for its payload): use TryRecv to ask for a message without blocking
  size_t Pending() const { return *(volatile size_t *)&meta_->send_cnt - *(volatile size_t *)&meta_->recv_cnt; }
  // device-visible address of a pointer into the (pinned) queue region, for kernels that read a host slot in place
  const void *DeviceVisiblePtr(const void *host_ptr) const {
    return dev_base_ + (static_cast<const char *>(host_ptr) - reinterpret_cast<const char *>(meta_));
  }

  // ---- device ring (see RingInfo) ----
  // sampler process, after fork, current device = the sampler's: allocates `slots` message slots in HBM and exports
  // them; false (with a warning) when the platform refuses the IPC handle -- payloads then stay on the host path
  bool CreateDeviceRing(int ring, uint32_t slots);
  // sampler: a free slot of its ring for message `key` (nullptr: none free, use the host slot)
  void *ClaimDeviceSlot(int ring, size_t key);
  // receiver: where the payload arrays of message `key` live -- the host slot itself (`host_msg`) or the ring slot,
  // mapped into this process on first use; *on_device tells which
  const void *Payload(size_t key, const void *host_msg, bool *on_device);
  // sampler, at shutdown: waits until every published device slot has been released, then frees the ring
  void DrainDeviceRing(int ring, double timeout_s);

 private:
  QueueMeta *meta_;
  size_t total_bytes_;
  const char *dev_base_ = nullptr;  // device-visible address of meta_ (set by PinMemory)
  void ServiceSpills(int ring);
  void *local_slot_[kMaxRings][kMaxRingSlots] = {};   // slots of the rings this process owns
  void *mapped_slot_[kMaxRings][kMaxRingSlots] = {};  // slots of other processes' rings, opened through IPC on first use
  bool owns_ring_[kMaxRings] = {};
  int map_state_[kMaxRings] = {};      // 0 none yet, 1 own ring, 2 mapped through IPC, 3 refused (read from the host slot)
  int map_device_[kMaxRings] = {};     // current device of the thread that read the ring first (valid when state != 0)
  std::thread svc_;                    // answers spill requests for the ring this process owns
  std::atomic<bool> svc_stop_{false};
  std::atomic<bool> closing_{false};   // Close(): process-local
};

// worst-case message size for a config (GetMaxMQSize, task_queue.cc:349-371)
size_t MaxMessageBytes(size_t batch_size, const size_t *fanout, size_t num_layers, bool have_data);

// what the pack kernel needs
struct PackArgs {
  const fgnn_batch_meta *d_meta;
  const uint32_t *input_nodes, *output_nodes;
  const uint32_t *cidx[4];
  const uint32_t *row[FGNN_MAX_LAYERS], *col[FGNN_MAX_LAYERS], *data[FGNN_MAX_LAYERS];
  int ship_input;        // !UseGPUCache || have_switcher (task_queue.cc:174)
  int ship_cache_index;  // UseGPUCache (task_queue.cc:184)
  int have_data;
  void *slot;            // host slot: headers (and the arrays too when `payload` is null)
  void *payload;         // device-ring slot for the arrays (same layout and offsets as the host slot), or null
  size_t slot_bytes;
  uint32_t *h_meta;      // pinned copy of the batch summary the pack kernel fills on the way (fgnn_batch_host_meta), or null
  uint32_t *msg_words;   // device word that receives the message length in 4-byte words (0: did not fit), or null
};
// enqueues the serialisation of one batch into `slot` (device-visible host memory)
int LaunchPack(const PackArgs &a, hipStream_t stream);

// receiver side: the arrays of a message that have to outlive the queue slot (ids, COO) are copied out by ONE kernel
// (the reference issues one cudaMemcpyAsync per array, task_queue.cc:257-347); the cache index arrays are not copied
// at all -- the gathers read them in place, from the sampler's HBM slot over xGMI or from the mapped host slot
struct UnpackArgs {
  static constexpr int kMaxSegments = 2 + 3 * FGNN_MAX_LAYERS;
  int num_segments;
  struct { uint32_t *dst; const uint32_t *src; size_t words; } seg[kMaxSegments];
};
int LaunchUnpack(const UnpackArgs &a, hipStream_t stream);

// SAMGRAPH_HANDOFF_CHECK.  Sender (verify == 0): position-weighted 64-bit sum of the message's words -> the two words
// behind the message (`msg` = where the payload was packed; length from *d_words, written by the pack kernel).
// Receiver (verify == 1): the same sum over `words` words read through `msg`, compared with the trailer; *d_result |= 1
// on a mismatch.
int LaunchMessageChecksum(uint32_t *msg, const uint32_t *d_words, size_t words, int verify, uint32_t *d_result,
                          hipStream_t stream);

}  // namespace sam
