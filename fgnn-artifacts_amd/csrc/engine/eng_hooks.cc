// eng_hooks.cc -- include/fgnn_engine_hooks.h
#include <signal.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstring>
#include <vector>

#include "eng_config.h"
#include "eng_dataset.h"
#include "eng_queue.h"
#include "eng_shuffler.h"
#include "fgnn_engine_hooks.h"

using namespace sam;

extern "C" void fgnn_host_shuffle_minstd0(uint32_t *data, size_t n, uint64_t seed) { ShuffleMinstd0(data, n, seed); }

extern "C" void fgnn_host_shuffler_partition(size_t num_data, size_t batch_size, int sampler_id, int num_sampler,
                                             int aligned, size_t out[7]) {
  const ShufflePartition p = Shuffler::Partition(num_data, batch_size, sampler_id, num_sampler, aligned != 0);
  out[0] = p.padded_size;
  out[1] = p.local_size;
  out[2] = p.num_local_step;
  out[3] = p.epoch_step;
  out[4] = p.step_offset;
  out[5] = p.dataset_offset;
  out[6] = p.last_batch_size;
}

extern "C" void fgnn_host_wire_sizes(size_t batch_size, const size_t *fanout, size_t num_layers, int have_data,
                                     size_t out[3]) {
  out[0] = sizeof(TransData);
  out[1] = sizeof(GraphData);
  out[2] = MaxMessageBytes(batch_size, fanout, num_layers, have_data != 0);
}

static uint64_t Mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

extern "C" int fgnn_host_queue_selftest(size_t slots, size_t slot_bytes, size_t messages, int producers,
                                        int consumers) {
  if (producers < 1 || consumers < 1 || slot_bytes < 64) return 2;
  MemoryQueue mq(slot_bytes, slots);
  // shared result area: seen[m] incremented by whoever receives message id m
  auto *seen = static_cast<uint32_t *>(mmap(nullptr, (messages + 1) * sizeof(uint32_t), PROT_READ | PROT_WRITE,
                                            MAP_SHARED | MAP_ANONYMOUS, -1, 0));
  if (seen == MAP_FAILED) return 2;
  std::vector<pid_t> kids;
  for (int p = 0; p < producers; ++p) {
    pid_t pid = fork();
    if (pid == 0) {
      for (size_t m = (size_t)p; m < messages; m += (size_t)producers) {
        size_t key;
        auto *w = static_cast<uint64_t *>(mq.GetPtr(&key));
        const size_t words = 2 + (Mix(m) % ((mq.SlotBytes() / 8) - 2));
        w[0] = m;
        w[1] = words;
        for (size_t i = 2; i < words; ++i) w[i] = Mix(m * 1315423911ull + i);
        mq.SimpleSend(key);
      }
      _exit(0);
    }
    kids.push_back(pid);
  }
  for (int c = 0; c < consumers; ++c) {
    pid_t pid = fork();
    if (pid == 0) {
      // consumer c takes ceil/floor share like the trainers do (train_graphsage.py:293-298)
      size_t mine = messages / (size_t)consumers + ((size_t)c < messages % (size_t)consumers ? 1 : 0);
      for (size_t k = 0; k < mine; ++k) {
        size_t key;
        auto *w = static_cast<const uint64_t *>(mq.Recv(&key));
        const size_t m = w[0], words = w[1];
        bool ok = m < messages && words >= 2 && words <= mq.SlotBytes() / 8;
        for (size_t i = 2; ok && i < words; ++i) ok = w[i] == Mix(m * 1315423911ull + i);
        if (ok) __sync_fetch_and_add(&seen[m], 1u);
        else __sync_fetch_and_add(&seen[messages], 1u);
        mq.Release(key);
      }
      _exit(0);
    }
    kids.push_back(pid);
  }
  int bad = 0;
  for (pid_t pid : kids) {
    int st = 0;
    waitpid(pid, &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) bad = 1;
  }
  if (seen[messages] != 0) bad = 1;
  for (size_t m = 0; m < messages; ++m)
    if (seen[m] != 1) bad = 1;
  munmap(seen, (messages + 1) * sizeof(uint32_t));
  return bad;
}

// The extraction thread's discipline (Engine::StartExtract): up to `depth` received messages are held UNRELEASED; a
// further one is taken only through TryRecv (published or nothing), with nothing held the consumer may block.  Must
// terminate for any slot count >= 2 -- a consumer that blocked for a claimed-but-unpublished message while holding
// slots would wait for a producer that waits for exactly those slots.
// A receiver blocked on an empty ring and a sender blocked on a full one must both give up (abort, like a failed CHECK)
// once the queue is marked aborted -- what samgraph_wait_one_child does when it sees a dead child.  Returns the number
// of the two children that ended by SIGABRT within the time limit (2 = both).
extern "C" int fgnn_host_queue_abort_selftest(void) {
  MemoryQueue mq(256, 2);
  pid_t receiver = fork();
  if (receiver == 0) {
    alarm(30);
    size_t key;
    (void)mq.Recv(&key);  // nothing is ever sent
    _exit(0);
  }
  MemoryQueue full(256, 1);
  pid_t sender = fork();
  if (sender == 0) {
    alarm(30);
    size_t key;
    (void)full.GetPtr(&key);
    full.SimpleSend(key);
    (void)full.GetPtr(&key);  // the only slot is never released
    _exit(0);
  }
  usleep(200000);
  mq.Abort();
  full.Abort();
  int aborted = 0;
  for (pid_t pid : {receiver, sender}) {
    int st = 0;
    if (waitpid(pid, &st, 0) == pid && WIFSIGNALED(st) && WTERMSIG(st) == SIGABRT) ++aborted;
  }
  return aborted;
}

extern "C" int fgnn_host_queue_selftest_deep(size_t slots, size_t slot_bytes, size_t messages, int producers,
                                             int consumers, int depth) {
  if (producers < 1 || consumers < 1 || slot_bytes < 64 || depth < 1) return 2;
  MemoryQueue mq(slot_bytes, slots);
  auto *seen = static_cast<uint32_t *>(mmap(nullptr, (messages + 1) * sizeof(uint32_t), PROT_READ | PROT_WRITE,
                                            MAP_SHARED | MAP_ANONYMOUS, -1, 0));
  if (seen == MAP_FAILED) return 2;
  std::vector<pid_t> kids;
  for (int p = 0; p < producers; ++p) {
    pid_t pid = fork();
    if (pid == 0) {
      alarm(60);  // a dead-locked ring ends the child instead of hanging the caller
      for (size_t m = (size_t)p; m < messages; m += (size_t)producers) {
        size_t key;
        auto *w = static_cast<uint64_t *>(mq.GetPtr(&key));
        if (Mix(m) % 7 == 0) usleep(200);  // a slot claimed long before it is published
        w[0] = m;
        w[1] = Mix(m);
        mq.SimpleSend(key);
      }
      _exit(0);
    }
    kids.push_back(pid);
  }
  for (int c = 0; c < consumers; ++c) {
    pid_t pid = fork();
    if (pid == 0) {
      alarm(60);
      const size_t mine = messages / (size_t)consumers + ((size_t)c < messages % (size_t)consumers ? 1 : 0);
      std::vector<size_t> held;  // keys, oldest first
      size_t taken = 0;
      while (taken < mine || !held.empty()) {
        if (taken < mine && (int)held.size() < depth) {
          const void *msg = nullptr;
          size_t key = 0;
          bool got = true;
          if (held.empty()) msg = mq.Recv(&key);
          else got = mq.TryRecv(&msg, &key);
          if (got) {
            auto *w = static_cast<const uint64_t *>(msg);
            if (w[0] < messages && w[1] == Mix(w[0])) __sync_fetch_and_add(&seen[w[0]], 1u);
            else __sync_fetch_and_add(&seen[messages], 1u);
            held.push_back(key);
            ++taken;
            continue;
          }
        }
        if (Mix(taken) % 5 == 0) usleep(100);  // "waiting for the GPU"
        mq.Release(held.front());
        held.erase(held.begin());
      }
      _exit(0);
    }
    kids.push_back(pid);
  }
  int bad = 0;
  for (pid_t pid : kids) {
    int st = 0;
    waitpid(pid, &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) bad = 1;
  }
  if (seen[messages] != 0) bad = 1;
  for (size_t m = 0; m < messages; ++m)
    if (seen[m] != 1) bad = 1;
  munmap(seen, (messages + 1) * sizeof(uint32_t));
  return bad;
}

extern "C" int fgnn_host_queue_named_role(size_t slots, size_t slot_bytes, size_t messages, int role, int index,
                                          int peers) {
  if (peers < 1 || index < 0 || index >= peers || slot_bytes < 64) return 2;
  MemoryQueue mq(slot_bytes, slots);
  if (role == 0) {
    for (size_t m = (size_t)index; m < messages; m += (size_t)peers) {
      size_t key;
      auto *w = static_cast<uint64_t *>(mq.GetPtr(&key));
      const size_t words = 2 + (Mix(m) % ((mq.SlotBytes() / 8) - 2));
      w[0] = m;
      w[1] = words;
      for (size_t i = 2; i < words; ++i) w[i] = Mix(m * 1315423911ull + i);
      mq.SimpleSend(key);
    }
    return 0;
  }
  int bad = 0;
  const size_t mine = messages / (size_t)peers + ((size_t)index < messages % (size_t)peers ? 1 : 0);
  for (size_t k = 0; k < mine; ++k) {
    size_t key;
    auto *w = static_cast<const uint64_t *>(mq.Recv(&key));
    const size_t m = w[0], words = w[1];
    bool ok = m < messages && words >= 2 && words <= mq.SlotBytes() / 8;
    for (size_t i = 2; ok && i < words; ++i) ok = w[i] == Mix(m * 1315423911ull + i);
    if (!ok) bad = 1;
    mq.Release(key);
  }
  return bad;
}

extern "C" void *fgnn_host_queue_open(size_t slots, size_t slot_bytes) { return new MemoryQueue(slot_bytes, slots); }
extern "C" void fgnn_host_queue_send(void *q, uint64_t key, uint64_t value) {
  auto *mq = static_cast<MemoryQueue *>(q);
  size_t k;
  auto *w = static_cast<uint64_t *>(mq->GetPtr(&k));
  w[0] = key;
  w[1] = value;
  mq->SimpleSend(k);
}
extern "C" void fgnn_host_queue_recv(void *q, uint64_t *key, uint64_t *value) {
  auto *mq = static_cast<MemoryQueue *>(q);
  size_t k;
  auto *w = static_cast<const uint64_t *>(mq->Recv(&k));
  *key = w[0];
  *value = w[1];
  mq->Release(k);
}
extern "C" void fgnn_host_queue_close(void *q) { delete static_cast<MemoryQueue *>(q); }

extern "C" int fgnn_host_queue_flip_word(const char *shm_name, size_t key, size_t word) {
  int fd = shm_open(shm_name, O_RDWR, 0600);
  if (fd < 0) return 2;
  struct stat st;
  if (fstat(fd, &st) != 0 || (size_t)st.st_size < kShmHeaderBytes + sizeof(QueueMeta)) {
    close(fd);
    return 1;
  }
  const size_t total = (size_t)st.st_size;
  void *p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return 2;
  const uint64_t *h = static_cast<const uint64_t *>(p);
  auto *m = reinterpret_cast<QueueMeta *>(static_cast<char *>(p) + kShmHeaderBytes);
  // is this region a queue?  (a job's regions are numbered, not named by content: the caller tries them in turn)
  const bool is_queue = h[0] == kShmMagic && m->max_size >= 1 && m->max_size <= kMaxSlots && m->mq_nbytes % 256 == 0 &&
                        m->mq_nbytes >= 256 && h[1] == sizeof(QueueMeta) + m->mq_nbytes * m->max_size &&
                        total == h[1] + kShmHeaderBytes;
  int rc = 1;
  if (is_queue && key < m->send_cnt && word * 4 + 4 <= m->mq_nbytes) {
    reinterpret_cast<uint32_t *>(m->data + (key % m->max_size) * m->mq_nbytes)[word] ^= 0x5A5A5A5Au;
    rc = 0;
  }
  munmap(p, total);
  return rc;
}

extern "C" int fgnn_host_config_probe(const char **keys, const char **vals, size_t n, size_t out[4]) {
  RunConfig rc;
  rc.Parse(keys, vals, n);
  out[0] = rc.fanout.size();
  out[1] = rc.fanout[0];
  out[2] = (size_t)rc.run_arch;
  out[3] = rc.UseGPUCache() ? 1 : 0;
  return 0;
}
