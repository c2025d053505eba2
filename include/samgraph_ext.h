/*
 * samgraph_ext.h -- entry points of c_lib.so that the reference's operation.h does NOT have.  They carry the samgraph_
 * prefix (the library exports `*samgraph_*` only, like the reference's samgraph.lds) and nothing in the reference-shaped
 * API depends on them: a script written for the reference never calls them.  bench.py uses them to make a multi-GPU
 * run explain itself.
 */
#ifndef SAMGRAPH_EXT_H
#define SAMGRAPH_EXT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Hand-off statistics of sampler `ring` (= its worker id), readable by every process of the job after
 * samgraph_data_init (the counters live in the shared queue region, eng_queue.h):
 *   out[0] slots of the sampler's HBM message ring (0: none -- every payload travels through the pinned host ring)
 *   out[1] messages whose payload went through the HBM ring     out[2] ... through the host ring
 *   out[3] messages copied back to the host slot on request (a receiver could not map the ring)
 *   out[4] messages a receiver verified end to end (SAMGRAPH_HANDOFF_CHECK)   out[5] ... that did NOT verify
 * Returns 0, or -1 when there is no queue / no such ring. */
int samgraph_ext_queue_stats(int ring, uint64_t out[6]);

/* How THIS process reads the payloads of sampler `ring` (per process, unlike the shared counters above):
 *   out[0] 0 = it has not read a message of that ring yet; 1 = the ring is its own (plain device pointer);
 *          2 = the sampler's HBM slots are mapped here (hipIpcOpenMemHandle with lazy peer access: payloads are read
 *              device to device, over xGMI when the two GPUs differ); 3 = the mapping was refused -- the owner copies
 *              every message back into the pinned host slot and this process reads it there
 *   out[1] the GPU the ring lives on          out[2] the GPU this process read it from (-1: none yet)
 * Returns 0, or -1 when there is no queue / no such ring. */
int samgraph_ext_ring_mapping(int ring, int64_t out[3]);

#ifdef __cplusplus
}
#endif
#endif
