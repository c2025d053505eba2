// eng_queue.h -- sampler -> trainer hand-off: a lock-free MPMC ring of fixed-size slots in a
// MAP_SHARED|MAP_ANONYMOUS host mapping, hipHostRegister'ed by every child after fork, with the
// reference's counters + process-shared semaphores (memory_queue.h:46-115, memory_queue.cc:33-138)
// and the reference's wire format (task_queue.cc:68-88: TransData 40 B header, GraphData 24 B).
//
// MI355X change: the sampler does not issue one D2H copy per tensor with sizes known on the host
// (task_queue.cc:154-227); ONE kernel (pack.hip) reads the sizes on the device and writes the whole
// serialized message into the mapped slot, so the sampler never waits for a size.
#pragma once
#include <semaphore.h>

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

struct QueueMeta {                 // MQ_MetaData, memory_queue.h:65-115
  size_t send_cnt, recv_cnt, max_size, mq_nbytes;
  sem_t sem_list[kMaxSlots];
  sem_t release_list[kMaxSlots];
  alignas(256) char data[0];
};

class MemoryQueue {
 public:
  MemoryQueue(size_t slot_bytes, size_t num_slots);  // in the parent, before fork
  void PinMemory();                                  // in each child (memory_queue.cc:47-49)
  void *GetPtr(size_t *key);                         // claim a slot for writing (blocks while the ring is full)
  void SimpleSend(size_t key);                       // publish
  const void *Recv(size_t *key);                     // blocks until a message is available
  bool TryRecv(const void **data, size_t *key);      // non-blocking: false if nothing has been claimed for sending
  void Release(size_t key);                          // SharedData::~SharedData
  size_t SlotBytes() const { return meta_->mq_nbytes; }
  size_t NumSlots() const { return meta_->max_size; }
  size_t Pending() const { return meta_->send_cnt - meta_->recv_cnt; }

 private:
  QueueMeta *meta_;
  size_t total_bytes_;
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
  void *slot;
  size_t slot_bytes;
};
// enqueues the serialisation of one batch into `slot` (device-visible host memory)
int LaunchPack(const PackArgs &a, hipStream_t stream);

}  // namespace sam
