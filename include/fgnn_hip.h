/*
 * fgnn_hip.h -- kernel-level C ABI of libfgnn_hip.so: the MI355X (gfx950) replacement for the
 * reference's L1 interface samgraph/common/cuda/cuda_function.h:30-111, cuda_hashtable.h:99-149 and
 * cuda_cache_manager.h:27-79 ("plain functions taking raw device pointers + a stream").
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name starts with h_;
 *  - `stream` is a hipStream_t passed as void*; calls only ENQUEUE work and never synchronise
 *    (the reference synchronises after every kernel, e.g. cuda_sampling_khop0.cu:193-244);
 *  - sizes can be given on the host (`num_*`) or, when the producing kernel has not finished yet,
 *    as a device scalar (`d_num_*`, may be NULL): the device value wins.  `*_cap` is the host-side
 *    upper bound used to size grids and scratch;
 *  - scratch comes from the caller (`ws`, `ws_bytes`; query with fgnn_scratch_bytes) so a launch
 *    sequence can be captured into a hipGraph (no allocation inside);
 *  - return 0 on success, a negative FGNN_E* code on a host-side argument error.  Device faults
 *    abort, like the reference's CHECK (logging.h:32-45).
 *  - ids are uint32 (IdType, common.h:35); FGNN_EMPTY_KEY = 0xFFFFFFFF (constant.h:71).
 */
#ifndef FGNN_HIP_H
#define FGNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FGNN_EMPTY_KEY 0xFFFFFFFFu
#define FGNN_OK 0
#define FGNN_EINVAL (-1)
#define FGNN_ENOSPC (-2) /* scratch too small */
#define FGNN_EHIP (-3)   /* a HIP runtime call failed */

/* DataType, common.h:38-46 */
enum { FGNN_F32 = 0, FGNN_F64 = 1, FGNN_F16 = 2, FGNN_U8 = 3, FGNN_I32 = 4, FGNN_I8 = 5, FGNN_I64 = 6 };
/* SampleType, common.h:50-58 (only used to derive RNG tags) */
enum { FGNN_KHOP0 = 0, FGNN_KHOP1 = 1, FGNN_WEIGHTED_KHOP = 2, FGNN_RANDOM_WALK = 3, FGNN_WEIGHTED_KHOP_PREFIX = 4,
       FGNN_KHOP2 = 5, FGNN_WEIGHTED_KHOP_HASH_DEDUP = 6 };

/* out_src contents of the samplers */
enum { FGNN_SRC_GLOBAL = 0, /* seed's global id, as the reference emits (khop2.cu:79) */
       FGNN_SRC_LOCAL = 1   /* seed's position in `input` == its local id after dedup, which lets
                               the engine skip the src half of GPUMapEdges (cuda_mapping.cu:56-66) */ };

const char *fgnn_version(void);
/* text of the last HIP runtime failure on this thread (FGNN_EHIP) */
const char *fgnn_last_error(void);
int fgnn_device_count(void);
/* Diagnostics (tools/phase_probe.py): install a device buffer of fgnn_debug_phase_log_bytes() bytes and the
 * single-pass kernels (sampler, dedup count+assign, cache split) stamp a 100 MHz wall clock per workgroup at
 * their phase boundaries: u64[kind 0..3][tile 0..4095][phase 0..7].  NULL (the default) switches it off. */
size_t fgnn_debug_phase_log_bytes(void);
void fgnn_debug_phase_log(unsigned long long *d_buf);
/* Diagnostics (tests/test_coresidency_gpu.py): a foreign tenant -- `workgroups` x 256 threads that keep their wave
 * slots for `usec` microseconds (<= 2 s) on `stream` and do nothing else. */
int fgnn_debug_occupy(size_t workgroups, unsigned usec, void *stream);
/* Diagnostics (bench.py, roofline_sample): num_items independent random 4-byte reads from array[0 .. num_elems), four in
 * flight per lane, as one launch -- what the memory system sustains for the sampling chain's access pattern on this
 * GPU right now (time it with events on `stream`; one read = one 64-byte fabric request when the array is far larger
 * than the caches).  d_sink: one device word (never written in practice). */
int fgnn_debug_random_reads(const uint32_t *array, size_t num_elems, size_t num_items, uint64_t salt, uint32_t *d_sink,
                            void *stream);
/* Diagnostics: how many tile aggregates of the single-pass kernels were recomputed by a waiting workgroup instead of
 * being read from the tile itself (current device, since the process started; synchronises).  0 on an idle GPU. */
unsigned long long fgnn_debug_scan_helps(void);
/* Diagnostics (tests): polls of one look-back descriptor before a waiting workgroup starts recomputing the missing
 * aggregate itself; 0 forces the helping path on every wait not satisfied at once, a negative value restores the
 * default.  Affects launches made after the call.  (The library reads no switch from the environment.) */
void fgnn_debug_set_scan_help_after(int polls);
/* Diagnostics (tests): distinct keys one bin of the partitioned last fill (hashtable_partition.hip) may hold in its LDS
 * table before the bin falls back to the global table; a small value forces the fall-back path, a negative one
 * restores the default (3/4 of the 8192 slots). */
void fgnn_debug_set_partition_lds_limit(int distinct_keys);
/* Diagnostics (tests): the stable (key, value) sort the stateless with-replacement samplers order their seeds with
 * (scan.hip; where the reference calls cub::DeviceRadixSort, cuda_sampling_weighted_khop_prefix.cu:200-215), on its
 * own: n uint32 pairs sorted by key in place, equal keys in their input order.  Allocates its scratch and waits. */
int fgnn_debug_sort_pairs(uint32_t *d_keys, uint32_t *d_vals, size_t n, void *stream);

/* Bytes of scratch that any single call below needs for `n_cap` items. */
size_t fgnn_scratch_bytes(size_t n_cap);

/* ---- sanity checks (SAMGRAPH_SANITY_CHECK) --------------------------------------------------- */
/* GPUSanityCheckList + GPUBatchSanityCheck (cuda/cuda_sanity_check.cu:28-88) in one launch: every id of `input` differs
 * from invalid_val and -- when seen_bits != NULL, a bitmap of fgnn_sanity_map_bytes(num_node) bytes that the caller
 * zeroes at the start of an epoch -- has not been seen before in this epoch (ids are marked as seen).  The outcome is
 * OR-ed into *d_flags (a device word the caller zeroes): 1 = an invalid id, 2 = a duplicate, 4 = an id >= num_node.
 * The reference assert()s inside the kernel; here the caller reads the word and decides. */
size_t fgnn_sanity_map_bytes(size_t num_node);
int fgnn_sanity_check_batch(uint32_t *seen_bits, size_t num_node, const uint32_t *input, size_t num_input,
                            uint32_t invalid_val, uint32_t *d_flags, void *stream);

/* ---- samplers ----------------------------------------------------------------------------- */

/* GPUSampleKHop0 (cuda_sampling_khop0.cu:178-253): fixed-fanout uniform sampling without
 * replacement by reservoir, compacted COO in seed-major order.  *d_num_out (size_t, device) gets the
 * edge count.  With ws_bytes >= fgnn_scratch_bytes(cap) + 4 * (cap + 16 + 1024 * (fanout + 3)) rows of more than 16 384
 * entries are drawn by all workgroups of the GPU before the sampler runs (one draw per row element, khop0.cu:41-90: a
 * hub row of 10^6 entries is otherwise one workgroup's job); smaller scratch: same results, the owner walks the row. */
int fgnn_sample_khop0(const uint32_t *indptr, const uint32_t *indices, const uint32_t *input,
                      size_t num_input, const uint32_t *d_num_input, size_t num_input_cap, size_t fanout,
                      uint32_t *out_src, uint32_t *out_dst, size_t *d_num_out, int src_mode,
                      uint64_t seed, uint64_t batch_key, uint32_t layer, void *ws, size_t ws_bytes,
                      void *stream);

/* GPUSampleKHop2 (cuda_sampling_khop2.cu:177-252): partial Fisher-Yates IN PLACE on the CSR row
 * (`indices` is mutated exactly as the reference mutates it), compacted COO. */
int fgnn_sample_khop2(const uint32_t *indptr, uint32_t *indices, const uint32_t *input,
                      size_t num_input, const uint32_t *d_num_input, size_t num_input_cap, size_t fanout,
                      uint32_t *out_src, uint32_t *out_dst, size_t *d_num_out, int src_mode,
                      uint64_t seed, uint64_t batch_key, uint32_t layer, void *ws, size_t ws_bytes,
                      void *stream);

/* GPUSampleWeightedKHopPrefix (cuda_sampling_weighted_khop_prefix.cu:148-255): with replacement,
 * x = U(0,1] * rowsum binary-searched in the per-row inclusive prefix sums `prob_prefix`; output
 * ordered by seed id ascending (the reference's stable radix sort by src), draws of a seed in draw
 * order with a draw dropped when it equals the seed's NEXT draw.  Scratch: fgnn_weighted_scratch_bytes. */
size_t fgnn_weighted_scratch_bytes(size_t num_input_cap, size_t fanout);
int fgnn_sample_weighted_khop_prefix(const uint32_t *indptr, const uint32_t *indices, const float *prob_prefix,
                                     const uint32_t *input, size_t num_input, const uint32_t *d_num_input,
                                     size_t num_input_cap, size_t fanout, uint32_t *out_src, uint32_t *out_dst,
                                     size_t *d_num_out, int src_mode, uint64_t seed, uint64_t batch_key,
                                     uint32_t layer, void *ws, size_t ws_bytes, void *stream);

/* GPUSampleKHop1 (cuda_sampling_khop1.cu:130-234): uniform WITH replacement, same sort-by-src + adjacent-duplicate
 * removal as the weighted samplers.  Scratch: fgnn_weighted_scratch_bytes. */
int fgnn_sample_khop1(const uint32_t *indptr, const uint32_t *indices, const uint32_t *input, size_t num_input,
                      const uint32_t *d_num_input, size_t num_input_cap, size_t fanout, uint32_t *out_src,
                      uint32_t *out_dst, size_t *d_num_out, int src_mode, uint64_t seed, uint64_t batch_key,
                      uint32_t layer, void *ws, size_t ws_bytes, void *stream);
/* GPUSampleWeightedKHop (cuda_sampling_weighted_khop.cu:132-236): alias method with replacement; alias_table holds
 * node ids (create_alias_table.cc:150-151).  Scratch: fgnn_weighted_scratch_bytes. */
int fgnn_sample_weighted_khop(const uint32_t *indptr, const uint32_t *indices, const float *prob_table,
                              const uint32_t *alias_table, const uint32_t *input, size_t num_input,
                              const uint32_t *d_num_input, size_t num_input_cap, size_t fanout, uint32_t *out_src,
                              uint32_t *out_dst, size_t *d_num_out, int src_mode, uint64_t seed, uint64_t batch_key,
                              uint32_t layer, void *ws, size_t ws_bytes, void *stream);
/* GPUSampleWeightedKHopHashDedup (cuda_sampling_weighted_khop_hash_dedup.cu:206-282): alias-method draws, a draw whose
 * value the seed already selected is rejected until `fanout` distinct neighbours are found (rows of length <= fanout
 * are taken whole); output in seed order, no sort.  fanout <= 50 (the reference's per-thread table).  A seed gives up
 * after 64 * fanout attempts (the reference would spin forever).  Scratch: fgnn_hash_dedup_scratch_bytes. */
size_t fgnn_hash_dedup_scratch_bytes(size_t num_input_cap);
int fgnn_sample_weighted_khop_hash_dedup(const uint32_t *indptr, const uint32_t *indices, const float *prob_table,
                                         const uint32_t *alias_table, const uint32_t *input, size_t num_input,
                                         const uint32_t *d_num_input, size_t num_input_cap, size_t fanout,
                                         uint32_t *out_src, uint32_t *out_dst, size_t *d_num_out, int src_mode,
                                         uint64_t seed, uint64_t batch_key, uint32_t layer, void *ws, size_t ws_bytes,
                                         void *stream);

/* GPUSampleRandomWalk + FrequencyHashmap::GetTopK (cuda_sampling_random_walk.cu:113-161,
 * cuda_frequency_hashmap.cu:1143-1367): num_walks restart walks of walk_len steps per seed; per seed
 * the K most frequently visited nodes (count desc, first visit asc); out_data = visit count.
 * Scratch: fgnn_random_walk_scratch_bytes. */
size_t fgnn_random_walk_scratch_bytes(size_t num_input_cap, size_t K);
int fgnn_sample_random_walk(const uint32_t *indptr, const uint32_t *indices, const uint32_t *input,
                            size_t num_input, const uint32_t *d_num_input, size_t num_input_cap, size_t walk_len,
                            double restart_prob, size_t num_walks, size_t K, uint32_t *out_src, uint32_t *out_dst,
                            uint32_t *out_data, size_t *d_num_out, int src_mode, uint64_t seed,
                            uint64_t batch_key, uint32_t layer, void *ws, size_t ws_bytes, void *stream);

/* ---- whole neighbourhoods (no sampling) ----------------------------------------------------- */

/* GPUExtractNeighbour (cuda_function.h:93-97, cuda_extract_neighbour.cu:111-169): every neighbour of every input node,
 * input order, CSR order inside a row (the reference emits in a tile-internal order; its consumers only dedup the list).
 * *d_num_out (size_t, device) = number of neighbours (exact below 2^32); at most out_cap ids are written (out_cap = 0:
 * count only).  Scratch: fgnn_extract_neighbour_scratch_bytes(num_input_cap). */
size_t fgnn_extract_neighbour_scratch_bytes(size_t num_input_cap);
int fgnn_extract_neighbour(const uint32_t *indptr, const uint32_t *indices, const uint32_t *input, size_t num_input,
                           const uint32_t *d_num_input, size_t num_input_cap, uint32_t *out, size_t out_cap,
                           size_t *d_num_out, void *ws, size_t ws_bytes, void *stream);
/* One layer of DoGPUSampleAllNeighbour (cuda_loops.cc:526-565: GPUExtractNeighbour + FillWithDupMutable), the "sampler"
 * of the static pre-sampling policy, with the dedup done by a direct-map stamp array over the node ids: every neighbour
 * u of frontier[0 .. n) with stamp[u] != mark gets stamp[u] = mark, freq[u] += 1 (freq may be NULL) and is appended to
 * next[] (order unspecified; ids beyond next_cap are counted but not stored).  *d_num_next (uint32, device) is ADDED to:
 * zero it before the call.  mark_frontier != 0: the frontier nodes themselves are stamped and counted first (the seeds,
 * FillWithUnique).  Levels are expanded from the previous level's `next` only, which reaches the same closed
 * neighbourhood as the reference's re-expansion of everything seen so far. */
int fgnn_neighbourhood_expand(const uint32_t *indptr, const uint32_t *indices, const uint32_t *frontier,
                              size_t num_frontier, const uint32_t *d_num_frontier, size_t frontier_cap, uint32_t *stamp,
                              uint32_t mark, uint32_t *freq, uint32_t *next, size_t next_cap, uint32_t *d_num_next,
                              int mark_frontier, void *stream);

/* ---- dedup / remap: OrderedHashTable (cuda_hashtable.h:99-149) ---------------------------- */

typedef struct fgnn_hashtable fgnn_hashtable;

/* max_items = PredictNumNodes(...) (common.cc:330-339).  Buckets are 8 bytes {key,value}; capacity is
 * the next power of two >= 2*max_items.  h_err != NULL receives the failing code. */
fgnn_hashtable *fgnn_hashtable_create(size_t max_items, int *h_err);
/* Same, for callers that know the largest fill (number of items of one fill_duplicates call): the spare bits of the
 * 32-bit bucket value then hold a generation, and fgnn_hashtable_reset becomes a counter bump instead of a wipe of the
 * whole table (the table is wiped once every 2^(31 - bits(max(max_items, max_fill_items))) - 1 resets).  Fills larger
 * than max_fill_items are refused (FGNN_EINVAL). */
fgnn_hashtable *fgnn_hashtable_create_ex(size_t max_items, size_t max_fill_items, int *h_err);
void fgnn_hashtable_destroy(fgnn_hashtable *ht);
size_t fgnn_hashtable_capacity(const fgnn_hashtable *ht);
/* Reset (cuda_hashtable.cu:714-723) */
int fgnn_hashtable_reset(fgnn_hashtable *ht, void *stream);
/* FillWithUnique (cuda_hashtable.cu:1017-1037): items are distinct; item i gets local id
 * num_items + i and is appended to the N2O list. */
int fgnn_hashtable_fill_unique(fgnn_hashtable *ht, const uint32_t *items, size_t num_items, void *stream);
/* FillWithDuplicates (cuda_hashtable.cu:725-807) + the dst half of GPUMapEdges
 * (cuda_mapping.cu:31-81), fused: every item not yet in the table gets the next local id in order of
 * FIRST OCCURRENCE; `mapped[i]` (may be NULL) receives the local id of items[i].
 * The N2O list (fgnn_hashtable_n2o) is extended; *d_num_unique (uint32, device) = new total. */
int fgnn_hashtable_fill_duplicates(fgnn_hashtable *ht, const uint32_t *items, size_t num_items,
                                   const size_t *d_num_items, size_t num_items_cap, uint32_t *mapped,
                                   void *ws, size_t ws_bytes, void *stream);
/* GPUMapEdges for arbitrary ids that are already in the table. */
int fgnn_hashtable_map(const fgnn_hashtable *ht, const uint32_t *items, size_t num_items,
                       const size_t *d_num_items, size_t num_items_cap, uint32_t *mapped, void *stream);
/* Redirects where the N2O list is written (storage for max_items ids; NULL = the table's own array): the
 * batch driver points it at the batch's input_nodes buffer so no copy is needed at the end. */
int fgnn_hashtable_set_n2o(fgnn_hashtable *ht, uint32_t *storage);
/* First fill of a batch on a reset table in ONE launch: item i gets local id i, items are also copied to
 * items_copy (may be NULL), *d_meta (may be NULL) is initialised with (key, num_layers, num_output). */
struct fgnn_batch_meta_s;
int fgnn_hashtable_start_batch(fgnn_hashtable *ht, const uint32_t *items, size_t num_items, uint32_t *items_copy,
                               struct fgnn_batch_meta_s *d_meta, uint64_t key, uint32_t num_layers, void *stream);
/* device pointers: N2O list (== `unique`, the next layer's input) and its length (uint32) */
const uint32_t *fgnn_hashtable_n2o(const fgnn_hashtable *ht);
const uint32_t *fgnn_hashtable_d_num_items(const fgnn_hashtable *ht);

/* ---- cache index split: GetMissCacheIndex (cuda_cache.cu:162-234) ------------------------- */

/* Stable two-way partition of nodes[] by table[node] == FGNN_EMPTY_KEY.
 * d_counts[0] = num_miss, d_counts[1] = num_cache (uint32, device). */
int fgnn_get_miss_cache_index(const uint32_t *table, const uint32_t *nodes, size_t num_nodes,
                              const uint32_t *d_num_nodes, size_t num_nodes_cap, uint32_t *miss_src,
                              uint32_t *miss_dst, uint32_t *cache_src, uint32_t *cache_dst,
                              uint32_t *d_counts, void *ws, size_t ws_bytes, void *stream);

/* GPUDynamicCacheManager::ReplaceCacheGPU (cuda_cache_manager_device.cu:212-246,632-708), the index half of the arch4
 * dynamic-cache prototype: table[old_nodes[i]] = FGNN_EMPTY_KEY for i < num_old, then table[new_nodes[i]] = i for
 * i < num_new (the rows of the batch that has just been extracted become the cache of the next one). */
int fgnn_cache_table_replace(uint32_t *table, const uint32_t *old_nodes, size_t num_old, const uint32_t *new_nodes,
                             size_t num_new, void *stream);

/* ---- row gathers: GPUExtract (cuda_extraction.cu:74-117), CombineMissData / CombineCacheData
 *      (cuda_cache_manager_device.cu:339-442) ------------------------------------------------- */

/* out[dst_index ? dst_index[i] : i, :] = src[src_index ? src_index[i] : i, :]  for i < n.
 * GPUExtract       = (src_index = index, dst_index = NULL)
 * CombineMissData  = (src_index = NULL,  dst_index = miss_dst)
 * CombineCacheData = (src_index = cache_src, dst_index = cache_dst)
 * src may be device memory or host memory registered/allocated as device-accessible. */
int fgnn_gather_rows(void *out, const void *src, const uint32_t *src_index, const uint32_t *dst_index,
                     size_t n, const uint32_t *d_n, size_t n_cap, size_t dim, int dtype, void *stream);
/* Same with the SOURCE row index ANDed with src_row_mask: the reference's mock extraction for feature tables that hold
 * only 2^k rows (SAMGRAPH_EMPTY_FEAT=k; cpu_extraction.cc:47-62, cuda_extraction.cu mock variants). */
int fgnn_gather_rows_masked(void *out, const void *src, const uint32_t *src_index, const uint32_t *dst_index,
                            size_t n, const uint32_t *d_n, size_t n_cap, size_t dim, int dtype, uint32_t src_row_mask,
                            void *stream);

/* fgnn_gather_rows_masked for a caller whose GPU ALSO runs the sampling chain (one process sampling and extracting:
 * the reference's arch2-4 / arch6 loops, cuda_loops_arch3.cc:178-196): rows read from HOST memory come over the host
 * link at ~50 GB/s whatever the grid, and a launch sized like an HBM gather keeps ~1 M slow reads in the memory pipeline
 * in front of every other kernel's misses -- the next batches' sampling chains then run 2-6x slower.  shared_gpu != 0
 * limits a host-source launch to 64 workgroups (HBM sources are not affected); shared_gpu == 0 is
 * fgnn_gather_rows_masked. */
int fgnn_gather_rows_shared(void *out, const void *src, const uint32_t *src_index, const uint32_t *dst_index, size_t n,
                            const uint32_t *d_n, size_t n_cap, size_t dim, int dtype, uint32_t src_row_mask,
                            int shared_gpu, void *stream);

/* The trainer-side feature extraction of one batch as ONE launch: ExtractMissData's row fetch
 * (cuda_cache_manager_host.cc:38-56) + CombineMissData + CombineCacheData (cuda_cache_manager_device.cu:165-210,339-442)
 * + the label rows (DoCPULabelExtractAndCopy, dist_loops.cc:886-929) + the copies that take a received message's arrays
 * out of its queue slot (ParseData, task_queue.cc:257-347):
 *   out[miss_dst[i], :]  = miss_rows[miss_src[i] & miss_row_mask, :]   i < num_miss
 *   out[cache_dst[i], :] = cache_rows[cache_src[i], :]                 i < num_cache
 *   label_out[i]         = label_src[label_index[i]]                   i < num_label   (label_out NULL: none)
 *   segs[k].dst[0:words] = segs[k].src[0:words]                        k < num_segs    (32-bit words)
 * link_workgroups > 0: miss_rows is host memory the GPU reads over the host link; that many workgroups (the "link
 * band") pull the miss rows while the rest of the grid streams the hit rows out of the HBM cache -- the link never
 * waits behind the HBM gather and the HBM gather never waits behind the link.  0: miss_rows is in HBM too, every
 * workgroup takes its share of both lists.  Counts are host values, or d_counts = {num_miss, num_cache} on the device
 * with `cap` the capacity of either list.  Rows must be a multiple of 16 bytes and 16-byte aligned (FGNN_EINVAL
 * otherwise: use fgnn_gather_rows per list); num_segs <= FGNN_MAX_COPY_SEGMENTS.
 * stamps: NULL, or u64[2 * fgnn_extract_fused_grid()] receiving every workgroup's start and end time (100 MHz wall
 * clock; workgroups [0, link_workgroups) are the link band): per-band durations of one launch. */
typedef struct {
  uint32_t *dst;
  const uint32_t *src;
  size_t words;
} fgnn_copy_segment;
#define FGNN_MAX_COPY_SEGMENTS 32
typedef struct {
  void *out;
  const void *miss_rows, *cache_rows;
  const uint32_t *miss_src, *miss_dst, *cache_src, *cache_dst;
  size_t num_miss, num_cache;
  const uint32_t *d_counts;
  size_t cap;
  size_t dim;
  int dtype;
  uint32_t miss_row_mask;
  void *label_out;
  const void *label_src;
  const uint32_t *label_index;
  size_t num_label;
  int label_dtype;
  const fgnn_copy_segment *segs;
  int num_segs;
  int link_workgroups;
  unsigned long long *stamps;
} fgnn_extract_job;
int fgnn_extract_fused(const fgnn_extract_job *job, void *stream);
/* workgroups the launch of `job` would use (the size of `stamps` / 2; 0: the job cannot take this path), and how many
 * of them form the link band (<= link_workgroups: a short miss list needs fewer) */
size_t fgnn_extract_fused_grid(const fgnn_extract_job *job);
size_t fgnn_extract_fused_link_grid(const fgnn_extract_job *job);
/* link_workgroups for a GPU that also runs the sampling chain / for a GPU that only extracts (an arch5 trainer).  Small
 * on purpose: 16 workgroups keep 256 KB of host reads in flight, twice what the link needs at its latency; more of them
 * only sit in the memory pipeline in front of every other access (56 GB/s with 16, 49 with 256: profiles/r06_a_*). */
#define FGNN_LINK_WGS_SHARED 16
#define FGNN_LINK_WGS_DEDICATED 16

/* ---- neighbourhood aggregation over a sampled block (consumer side of the path) ----------- */

/* out[dst_index[e], :] += edge_weight[e] * h[src_index[e], :]  for e < num_edge (edge_weight NULL = 1), fp32.
 * `out` must be initialised by the caller (zeros for a plain aggregation).  Forward of the mean / sum / weighted
 * aggregators: src_index = block row (neighbour), dst_index = block col (seed); backward: the two swapped with
 * h = grad_out.  Fastest when equal dst_index values are contiguous (the samplers' seed-major edge order). */
int fgnn_block_aggregate(const uint32_t *src_index, const uint32_t *dst_index, const float *edge_weight,
                         size_t num_edge, const float *h, size_t dim, float *out, void *stream);
/* Same with a row stride for `out` (out_ld >= dim elements: the sums land in a column block of a wider matrix, e.g. the
 * right half of [h_dst | mean h_u]) and, if in_degree != NULL, in_degree[dst_index[e]] += 1 for every edge on the way
 * (float counts into a caller-zeroed array: the mean's denominators without a pass of their own). */
int fgnn_block_aggregate_ex(const uint32_t *src_index, const uint32_t *dst_index, const float *edge_weight,
                            size_t num_edge, const float *h, size_t dim, float *out, size_t out_ld, float *in_degree,
                            void *stream);

/* ---- fused pieces of a GraphSAGE training step (consumer side, csrc/train_ops.hip) ----------------------------------
 * Elementwise work a training step otherwise does with two to four framework ops each; one launch apiece because the
 * step is replayed as a captured HIP graph, where every node costs the GPU 15-20 us whatever it does.  fp32, the same
 * arithmetic per element as the ops they replace (reference: the DGL / PyTorch layers and loop of
 * example/samgraph/multi_gpu/train_graphsage.py:24-51,300-330). */
/* SAGEConv('mean') forward, after fgnn_block_aggregate_ex has summed the neighbours into z[:, din:2 din) and counted the
 * in-degrees: inv[i] = 1 / max(deg[i], 1); z[i, din:] *= inv[i]; z[i, :din] = h[i, :din].  din % 4 == 0, 16-byte
 * aligned rows. */
int fgnn_sage_finish_z(float *z, size_t ld, const float *h, size_t h_ld, const float *deg, float *inv, size_t num_dst,
                       size_t din, void *stream);
/* its backward half: gh[i, :] = gz[i, :din] for i < num_dst, 0 for num_dst <= i < num_src (the self path; the neighbour
 * path is then ADDED by fgnn_block_aggregate with row / col swapped); gagg[i, :] = gz[i, din:] * inv[i]. */
int fgnn_sage_grad_prep(const float *gz, size_t gz_ld, const float *inv, float *gh, float *gagg, size_t num_dst,
                        size_t num_src, size_t din, void *stream);
/* y = relu(x) * keep / (1 - p) with keep ~ Bernoulli(1 - p): Philox keyed by (seed, *d_step, layer_tag, element).
 * d_step (device, nullable): the training step count fgnn_adam_step keeps.  n % 4 == 0.
 * backward: gx = gy / (1 - p) where y > 0, else 0. */
int fgnn_relu_dropout(const float *x, float *y, size_t n, float p, uint64_t seed, const unsigned long long *d_step,
                      uint32_t layer_tag, void *stream);
int fgnn_relu_dropout_backward(const float *y, const float *gy, float *gx, size_t n, float p, void *stream);
/* CrossEntropyLoss(reduction='mean') and its gradient in one launch: *loss = mean_i (logsumexp(x_i) - x_i[label_i]),
 * dlogits[i, c] = (softmax(x_i)[c] - [c == label_i]) / n.  The mean is summed in a fixed order (bit-reproducible).
 * A row whose label is outside [0, num_class) is IGNORED the way torch ignores ignore_index rows (default -100): no
 * loss, a zero gradient row, and n above is the number of the OTHER rows (no such row: loss 0, torch gives nan).
 * ws: fgnn_softmax_xent_scratch_bytes(n) bytes, its first 16 zeroed ONCE by the caller (the launches keep them zero). */
size_t fgnn_softmax_xent_scratch_bytes(size_t n);
int fgnn_softmax_xent(const float *logits, size_t ld, const long long *labels, size_t n, size_t num_class, float *loss,
                      float *dlogits, size_t dl_ld, void *ws, size_t ws_bytes, void *stream);
/* torch.optim.Adam's update (no amsgrad) for up to 8 tensors in one launch; h_* are HOST arrays of device pointers /
 * sizes.  d_step: device, two 64-bit words zeroed once by the caller -- [0] the number of steps taken (read, then
 * advanced by this launch), [1] scratch. */
int fgnn_adam_step(float *const *h_params, const float *const *h_grads, float *const *h_exp_avg,
                   float *const *h_exp_avg_sq, const size_t *h_numel, int count, float lr, float beta1, float beta2,
                   float eps, float weight_decay, unsigned long long *d_step, void *stream);

/* ---- pre-sampling cache policy (init-time) --------------------------------------------------
 * PreSampler (dist/pre_sampler.cc:75-162): freq[node] += 1 for every input node of every presample batch, then
 * rank = nodes by (frequency desc, id desc).  fgnn_cache_table_build = SampleCacheTableInit (dist_engine.cc:193-229):
 * table[rank[i]] = i for i < num_cached, FGNN_EMPTY_KEY elsewhere. */
int fgnn_presample_count(uint32_t *d_freq, const uint32_t *d_nodes, size_t num_nodes, const uint32_t *d_num_nodes,
                         size_t num_nodes_cap, void *stream);
size_t fgnn_presample_rank_scratch_bytes(size_t num_node);
int fgnn_presample_rank(const uint32_t *d_freq, size_t num_node, uint32_t *d_rank, void *ws, size_t ws_bytes,
                        void *stream);
int fgnn_cache_table_build(uint32_t *d_table, size_t num_node, const uint32_t *d_rank, size_t num_cached,
                           void *stream);

/* ---- batch driver -------------------------------------------------------------------------
 * One object per sampler GPU that enqueues a whole mini-batch without a single host round trip:
 * DoGPUSample (cuda_loops.cc:50-267 == dist/dist_loops.cc:51-269), DoGetCacheMissIndex
 * (dist_loops.cc:271-323) and DoGPUFeatureExtract (cuda_loops.cc:726-770).  The reference allocates
 * every intermediate from a pool and synchronises ~40 times per layer; here all buffers are sized
 * once for the worst case (PredictNumNodes, common.cc:330-339), counts stay on the device, and the
 * host reads one small pinned summary per batch. */

#define FGNN_MAX_LAYERS 8

typedef struct fgnn_sampler fgnn_sampler;
typedef struct fgnn_batch fgnn_batch;

typedef struct {
  const uint32_t *indptr;   /* device, u32[num_node + 1] */
  uint32_t *indices;        /* device, u32[num_edge]; mutated by khop2 like the reference */
  const float *prob_prefix; /* device, f32[num_edge] or NULL (weighted_khop_prefix only) */
  size_t num_node;
  int sample_type;          /* SampleType, common.h:50-58 */
  size_t num_layers;
  size_t fanout[FGNN_MAX_LAYERS];
  size_t max_batch_size;
  uint64_t seed;            /* Philox seed (replaces the clock seed of cuda_random_states.cu:105-107) */
  size_t walk_len, num_walks; /* random walk: RunConfig::random_walk_length / num_random_walk */
  double restart_prob;
  const float *prob_table;     /* device, f32[num_edge]: weighted_khop (alias method) only */
  const uint32_t *alias_table; /* device, u32[num_edge] node ids: weighted_khop only */
} fgnn_sampler_config;

/* Host-visible summary of one batch (Task / TrainGraph / MissCacheIndex sizes, common.h:186-222);
 * valid after fgnn_batch_wait. */
typedef struct fgnn_batch_meta_s {
  uint64_t key;
  uint64_t num_edge[FGNN_MAX_LAYERS]; /* TrainGraph::num_edge */
  uint32_t num_src[FGNN_MAX_LAYERS];  /* TrainGraph::num_src = #unique after the layer */
  uint32_t num_dst[FGNN_MAX_LAYERS];  /* TrainGraph::num_dst = #seeds of the layer */
  uint32_t num_layers;
  uint32_t num_input;                 /* |input_nodes| */
  uint32_t num_output;                /* |output_nodes| = batch size */
  uint32_t num_miss, num_cache;
  uint32_t overflow;                  /* non-zero: the batch is invalid (a capacity was exceeded, or a cross-workgroup
                                         wait inside a single-pass kernel timed out) -- callers must not use it */
  /* Device time stamps (the GPU's 100 MHz wall clock, 10 ns ticks) taken by the kernels themselves, so that a caller
   * who wants the reference's per-batch sample / cache-index times (kLogL1SampleTime, kLogL3CacheGetIndexTime) needs
   * no event records around the calls: first sampler launch of the batch started; cache-index split started
   * (= sampling done; 0 if fgnn_batch_cache_index was not called); a caller's closing kernel started (written by that
   * kernel, e.g. the engine's pack kernel; 0 otherwise). */
  uint64_t t_start, t_sampled, t_closed;
} fgnn_batch_meta;

/* Streams: the sampler keeps no stream handle beyond the call it was passed to (it compares handles to tell whether a
 * slot or the CSR order changes streams, it never enqueues on a remembered one): a caller may destroy a stream once the
 * work it enqueued there has finished.  Hand-overs between streams are ordered by events recorded at the END of the
 * earlier call wherever the previous hand-over crossed streams too; the first crossing after a change of pattern (the
 * pre-sampling stream -> the batch streams) waits for the device instead. */
fgnn_sampler *fgnn_sampler_create(const fgnn_sampler_config *cfg, int *h_err);
void fgnn_sampler_destroy(fgnn_sampler *s);
/* weighted_khop_prefix samplers build, at creation, a 5-ary search tree over every prefix-table row longer than 64
 * entries (a draw then costs ceil(log5 deg) 16-byte look-ups instead of the binary search's ceil(log2 deg),
 * cuda_sampling_weighted_khop_prefix.cu:66-86; same result on every non-decreasing row, rows that are not keep the
 * reference's search): out[0] = rows with a tree, out[1] = long rows refused (not non-decreasing), out[2] = bytes. */
int fgnn_sampler_prefix_tree_stats(const fgnn_sampler *s, size_t out[3]);
size_t fgnn_sampler_max_nodes(const fgnn_sampler *s);            /* PredictNumNodes(batch, fanout, L) */
size_t fgnn_sampler_max_edges(const fgnn_sampler *s, int layer); /* worst-case edges of graphs[layer] */

/* Output buffers of one in-flight batch.  feat_rows_cap = 0 sizes the feature buffer for the worst
 * case; feat_dim = 0 skips feature/label buffers (sampler-only use). */
fgnn_batch *fgnn_batch_create(const fgnn_sampler *s, size_t feat_dim, int feat_dtype, int label_dtype,
                              size_t feat_rows_cap, int *h_err);
void fgnn_batch_destroy(fgnn_batch *b);

/* DoGPUSample: Reset, FillWithUnique(seeds), then per layer (last fanout first) sample ->
 * FillWithDuplicates -> remap.  graphs[l]: row = local id of the sampled neighbour, col = local id of
 * the seed, data = random-walk visit count.  input_nodes = final unique list. */
int fgnn_sampler_sample(fgnn_sampler *s, const uint32_t *d_seeds, size_t num_seeds, uint64_t batch_key,
                        fgnn_batch *out, void *stream);
/* fgnn_sampler_sample followed by fgnn_batch_cache_index (cache_table may be NULL: sample only) as one call -- what an
 * arch5 sampler process does per batch (dist_loops_arch5.cc:86-105); one launch fewer than the two calls. */
int fgnn_sampler_sample_indexed(fgnn_sampler *s, const uint32_t *d_seeds, size_t num_seeds, uint64_t batch_key,
                                fgnn_batch *out, const uint32_t *cache_table, void *stream);
/* Thread-safe, explicitly ordered variant for overlapping batches: `seq` = 0,1,2,... is the batch's position
 * in the run; calls may come from several host threads (one per stream) in any timing, the library makes
 * khop2's in-place CSR swaps happen in `seq` order and keeps at most 6 batches in flight.  fgnn_sampler_sample
 * is this with an internal counter (do not mix the two on one sampler). */
int fgnn_sampler_sample_ordered(fgnn_sampler *s, uint64_t seq, const uint32_t *d_seeds, size_t num_seeds,
                                uint64_t batch_key, fgnn_batch *out, void *stream);
/* A batch in two calls, for callers that keep several batches in flight.  begin: everything up to and including the
 * batch's LAST sampler launch -- for khop2, whose kernels rewrite CSR rows in batch order, the part the NEXT batch's first
 * sampler launch waits for on the GPU.  end: the last layer's dedup fill, the fix-ups, the table's reset [+ the cache
 * index split when cache_table != NULL]; the caller then appends extraction / fgnn_batch_finish on the same stream.
 * Enqueue begin(k + 1) BEFORE end(k): whole batches enqueued one after the other put the next batch's first sampler
 * launch behind ~6 launches of this batch's tail in the host's enqueueing order, and the cross-batch chain then waits
 * for the host (an arch5 sampler process: 30 us of every 100).  begin: internal sequence counter (as fgnn_sampler_sample),
 * *seq_out = the batch's number for end; begin_ordered: explicit numbers (as fgnn_sampler_sample_ordered).  Results are
 * those of the one-call forms.  `stream` of end = the stream of begin. */
int fgnn_sampler_sample_begin(fgnn_sampler *s, const uint32_t *d_seeds, size_t num_seeds, uint64_t batch_key,
                              fgnn_batch *out, void *stream, uint64_t *seq_out);
int fgnn_sampler_sample_begin_ordered(fgnn_sampler *s, uint64_t seq, const uint32_t *d_seeds, size_t num_seeds,
                                      uint64_t batch_key, fgnn_batch *out, void *stream);
int fgnn_sampler_sample_end(fgnn_sampler *s, uint64_t seq, fgnn_batch *out, const uint32_t *cache_table, void *stream);
/* sample_ordered + cache_index (if cache_table) + extract (if feat/label) + finish in one call. */
int fgnn_sampler_run_batch(fgnn_sampler *s, uint64_t seq, const uint32_t *d_seeds, size_t num_seeds,
                           uint64_t batch_key, fgnn_batch *out, const uint32_t *cache_table, const void *feat,
                           const void *label, void *stream);
/* The batch loop of a GPU that samples and extracts, as ONE native call per range of batches (the reference runs this
 * loop in C++ threads: RunSampleCopySubLoopOnce, cuda_loops_arch1.cc:38-84; per batch: the next batch_size ids of the
 * shuffled train set -> DoGPUSample -> DoGetCacheMissIndex -> DoGPUFeatureExtract -> Submit).  Batch i (= its `seq`)
 * takes step i % steps_per_epoch of d_train (the last step of an epoch is short), batch_key = that step, output buffer
 * batches[i % num_batches], stream streams[i % num_streams]: whole batches overlap on the GPU, khop2's CSR swaps stay
 * in batch order.  cached == 0: fgnn_sampler_run_batch(cache_table, feat, label); cached != 0:
 * fgnn_sampler_run_batch_cached(cache_table, cache_rows, full_feat, label).  The call returns when every batch of the
 * range has finished: h_metas[count] receives the summaries in order, h_gather_ms (NULL or float[count][2]) the
 * times of the batches whose buffers have timing enabled ({gather by HIP events, gather by its own clock stamps} or
 * {link band, HBM band of the one-launch cached extraction}; -1 where not timed), *h_enqueue_s (NULL ok) the host time spent enqueueing. Sequence numbers must continue those of
 * earlier calls on the sampler.  num_streams of 1, 2, 3 or 6 needs no event between a slot's uses. */
typedef struct {
  const uint32_t *d_train;
  size_t num_train, batch_size;
  fgnn_batch **batches;
  size_t num_batches;
  void **streams;
  size_t num_streams;
  const uint32_t *cache_table;
  const void *feat, *label;
  int cached;
  const void *cache_rows, *full_feat;
} fgnn_run_plan;
int fgnn_sampler_run_range(fgnn_sampler *s, const fgnn_run_plan *plan, uint64_t first_seq, size_t count,
                           fgnn_batch_meta *h_metas, float *h_gather_ms, double *h_enqueue_s);
/* DoGetCacheMissIndex on input_nodes against a direct-map table u32[num_node]. */
int fgnn_batch_cache_index(fgnn_batch *b, const uint32_t *cache_table, void *stream);
/* DoGPUFeatureExtract: feat_out[i,:] = feat[input_nodes[i],:], label_out[i] = label[output_nodes[i]].
 * Either source may be NULL to skip it. */
/* Feature tables with 2^k rows (SAMGRAPH_EMPTY_FEAT): node ids are ANDed with `mask` before indexing the feature
 * source in fgnn_batch_extract / fgnn_batch_extract_cached (labels are not masked).  Default 0xFFFFFFFF. */
int fgnn_batch_set_feat_row_mask(fgnn_batch *b, uint32_t mask);
int fgnn_batch_extract(fgnn_batch *b, const void *feat, const void *label, void *stream);
/* Trainer-side DoCacheFeatureCopy (dist_loops.cc:713-846) with the miss rows gathered by the GPU:
 * feat_out[cache_dst] = cache_rows[cache_src]; feat_out[miss_dst] = full_feat[miss_src] where
 * full_feat is device-accessible (HBM, or pinned / registered host memory read over the host link). */
int fgnn_batch_extract_cached(fgnn_batch *b, const void *cache_rows, const void *full_feat, const void *label,
                              void *stream);
/* fgnn_sampler_run_batch with the cached extraction: sample_ordered + cache_index + extract_cached + finish */
int fgnn_sampler_run_batch_cached(fgnn_sampler *s, uint64_t seq, const uint32_t *d_seeds, size_t num_seeds,
                                  uint64_t batch_key, fgnn_batch *out, const uint32_t *cache_table,
                                  const void *cache_rows, const void *full_feat, const void *label, void *stream);
/* with fgnn_batch_enable_timing: out[0] = ms of the miss rows, out[1] = ms of the cached rows of the last
 * fgnn_batch_extract_cached (valid after fgnn_batch_wait; -1 when not bracketed).  The extraction is ONE launch: the two
 * figures are the durations of its link band and of its HBM band (first start .. last end over the band's workgroups,
 * device wall clock); fgnn_batch_extract_launch_ms is the HIP-event time around the whole launch. */
int fgnn_batch_extract_cached_ms(fgnn_batch *b, float out[2]);
float fgnn_batch_extract_launch_ms(fgnn_batch *b);
/* Optional HIP-event bracket around the feature gather launched by fgnn_batch_extract (on its stream);
 * fgnn_batch_gather_ms returns the elapsed time of the last bracketed launch after fgnn_batch_wait, or -1. */
int fgnn_batch_enable_timing(fgnn_batch *b, int on);
float fgnn_batch_gather_ms(fgnn_batch *b);
/* the same launch's OWN duration: first start .. last end over its workgroups (100 MHz device clock words the kernel
 * posts to pinned memory) -- what a kernel trace reports; the event bracket also holds the launch gap and the wait for
 * wave slots behind other batches' kernels.  -1 when not timed (or the gather did not take the 16-byte-row path). */
float fgnn_batch_gather_kernel_ms(fgnn_batch *b);
/* async copy of the summary to pinned host memory + event */
int fgnn_batch_finish(fgnn_batch *b, void *stream);
/* A caller whose own kernel runs last on the batch's stream anyway (the engine's message pack kernel) can let that
 * kernel copy the summary: write sizeof(fgnn_batch_meta) bytes from fgnn_batch_device_meta() to this pinned,
 * device-visible address, call fgnn_batch_meta_copied(), then fgnn_batch_finish() -- which then only records the event. */
fgnn_batch_meta *fgnn_batch_host_meta(const fgnn_batch *b);
int fgnn_batch_meta_copied(fgnn_batch *b);
/* blocks until the batch's event; copies the summary to *h_meta (may be NULL) */
int fgnn_batch_wait(fgnn_batch *b, fgnn_batch_meta *h_meta);

/* device pointers into a batch */
const uint32_t *fgnn_batch_row(const fgnn_batch *b, int layer);
const uint32_t *fgnn_batch_col(const fgnn_batch *b, int layer);
const uint32_t *fgnn_batch_data(const fgnn_batch *b, int layer);
const uint32_t *fgnn_batch_input_nodes(const fgnn_batch *b);
const uint32_t *fgnn_batch_output_nodes(const fgnn_batch *b);
const void *fgnn_batch_feat(const fgnn_batch *b);
const void *fgnn_batch_label(const fgnn_batch *b);
/* which: 0 miss_src, 1 miss_dst, 2 cache_src, 3 cache_dst */
const uint32_t *fgnn_batch_cache_index_ptr(const fgnn_batch *b, int which);
const fgnn_batch_meta *fgnn_batch_device_meta(const fgnn_batch *b);

#ifdef __cplusplus
}
#endif
#endif
