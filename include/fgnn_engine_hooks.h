/*
 * fgnn_engine_hooks.h -- host-only entry points of the engine library (c_lib.so) that let the CPU
 * test-suite exercise the engine's host logic without a GPU.  They run exactly the code the
 * samgraph_* path runs; nothing here is needed by an application.
 */
#ifndef FGNN_ENGINE_HOOKS_H
#define FGNN_ENGINE_HOOKS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the shufflers' per-epoch Fisher-Yates (dist/dist_shuffler.cc:108-131), in place */
void fgnn_host_shuffle_minstd0(uint32_t *data, size_t n, uint64_t seed);

/* one sampler's share of an epoch: aligned == 0 is DistShuffler (dist/dist_shuffler.cc:47-79), aligned != 0 is
 * DistAlignedShuffler (dist/dist_shuffler_aligned.cc:45-71, arch6 / arch7).  out = {padded train-set size, ids of this
 * sampler, its steps per epoch, steps per epoch over all samplers, its first global step, its first id's offset in
 * the shuffled array, size of its last batch} */
void fgnn_host_shuffler_partition(size_t num_data, size_t batch_size, int sampler_id, int num_sampler, int aligned,
                                  size_t out[7]);

/* sizes the wire format was compiled with: out[0] = sizeof(TransData) (40), out[1] = sizeof(GraphData) (24),
 * out[2] = worst-case message bytes for (batch_size, fanout[num_layers], have_data) (task_queue.cc:349-371) */
void fgnn_host_wire_sizes(size_t batch_size, const size_t *fanout, size_t num_layers, int have_data, size_t out[3]);

/* Multi-process self-test of the shared-memory hand-off ring (memory_queue.cc): forks `producers`
 * writer processes and `consumers` reader processes over a ring of `slots` slots of `slot_bytes`,
 * sends `messages` checksummed messages in total and verifies that every one arrives exactly once and
 * intact.  Returns 0 on success. */
int fgnn_host_queue_selftest(size_t slots, size_t slot_bytes, size_t messages, int producers, int consumers);

/* A receiver blocked on an empty ring and a sender blocked on a full one after MemoryQueue::Abort (the parent saw a
 * child die, samgraph_wait_one_child): returns how many of the two ended by SIGABRT (2 = both; the reference's
 * semaphore waits, memory_queue.cc:104-138, hang forever in that situation). */
int fgnn_host_queue_abort_selftest(void);
/* The same, with consumers that behave like the engine's extraction thread (eng_engine.cc: StartExtract): each holds up
 * to `depth` received messages unreleased and takes a further one only when it is already published (TryRecv); with
 * nothing held it blocks.  Must terminate for any slots >= 2 (a ring of 2-3 slots is what large fan-outs leave under
 * SAMGRAPH_MQ_BYTES).  Returns 0 on success. */
int fgnn_host_queue_selftest_deep(size_t slots, size_t slot_bytes, size_t messages, int producers, int consumers,
                                  int depth);

/* The same ring between processes that were NOT forked from a common parent (one process per GPU started by torchrun):
 * with SAMGRAPH_SHM_PREFIX set, every process calling this attaches to the same named ring (eng_dataset.cc:
 * SharedCreate).  role 0 = producer `index` of `peers` (sends its share of `messages`), role 1 = consumer `index` of
 * `peers` (receives its share, verifies the checksums).  Returns 0 on success. */
int fgnn_host_queue_named_role(size_t slots, size_t slot_bytes, size_t messages, int role, int index, int peers);

/* The ring as single calls, for control-plane rehearsals of a multi-process job without a GPU (bench.py --rehearse):
 * open attaches to / creates the job's ring (SAMGRAPH_SHM_PREFIX), send publishes one message {key, value}, recv
 * blocks for the next message. */
void *fgnn_host_queue_open(size_t slots, size_t slot_bytes);
void fgnn_host_queue_send(void *q, uint64_t key, uint64_t value);
void fgnn_host_queue_recv(void *q, uint64_t *key, uint64_t *value);
void fgnn_host_queue_close(void *q);

/* Test of the hand-off check (SAMGRAPH_HANDOFF_CHECK) from OUTSIDE the engine: flips bits of 32-bit word `word` of the
 * published message `key` in the host ring of a running job, found in the POSIX shared-memory object `shm_name`
 * ("/<SAMGRAPH_SHM_PREFIX>.<k>").  0: done; 1: that object is not a message ring, or holds no such message (try the
 * next k); 2: cannot be opened.  The engine itself has no switch that corrupts anything. */
int fgnn_host_queue_flip_word(const char *shm_name, size_t key, size_t word);

/* Parses a config through the same code as samgraph_config and returns 0; on an invalid config the
 * process aborts like the reference.  Writes steps-per-epoch style derived values for inspection:
 * out[0] = #layers, out[1] = fanout[0], out[2] = run_arch, out[3] = UseGPUCache. */
int fgnn_host_config_probe(const char **keys, const char **vals, size_t n, size_t out[4]);

#ifdef __cplusplus
}
#endif
#endif
