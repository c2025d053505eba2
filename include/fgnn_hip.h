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
enum { FGNN_KHOP0 = 0, FGNN_RANDOM_WALK = 3, FGNN_WEIGHTED_KHOP_PREFIX = 4, FGNN_KHOP2 = 5 };

/* out_src contents of the samplers */
enum { FGNN_SRC_GLOBAL = 0, /* seed's global id, as the reference emits (khop2.cu:79) */
       FGNN_SRC_LOCAL = 1   /* seed's position in `input` == its local id after dedup, which lets
                               the engine skip the src half of GPUMapEdges (cuda_mapping.cu:56-66) */ };

const char *fgnn_version(void);
/* text of the last HIP runtime failure on this thread (FGNN_EHIP) */
const char *fgnn_last_error(void);
int fgnn_device_count(void);

/* Bytes of scratch that any single call below needs for `n_cap` items. */
size_t fgnn_scratch_bytes(size_t n_cap);

/* ---- samplers ----------------------------------------------------------------------------- */

/* GPUSampleKHop0 (cuda_sampling_khop0.cu:178-253): fixed-fanout uniform sampling without
 * replacement by reservoir, compacted COO in seed-major order.  *d_num_out (size_t, device) gets the
 * edge count. */
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

/* ---- dedup / remap: OrderedHashTable (cuda_hashtable.h:99-149) ---------------------------- */

typedef struct fgnn_hashtable fgnn_hashtable;

/* max_items = PredictNumNodes(...) (common.cc:330-339).  Buckets are 8 bytes {key,value}; capacity is
 * the next power of two >= 2*max_items.  h_err != NULL receives the failing code. */
fgnn_hashtable *fgnn_hashtable_create(size_t max_items, int *h_err);
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

/* ---- row gathers: GPUExtract (cuda_extraction.cu:74-117), CombineMissData / CombineCacheData
 *      (cuda_cache_manager_device.cu:339-442) ------------------------------------------------- */

/* out[dst_index ? dst_index[i] : i, :] = src[src_index ? src_index[i] : i, :]  for i < n.
 * GPUExtract       = (src_index = index, dst_index = NULL)
 * CombineMissData  = (src_index = NULL,  dst_index = miss_dst)
 * CombineCacheData = (src_index = cache_src, dst_index = cache_dst)
 * src may be device memory or host memory registered/allocated as device-accessible. */
int fgnn_gather_rows(void *out, const void *src, const uint32_t *src_index, const uint32_t *dst_index,
                     size_t n, const uint32_t *d_n, size_t n_cap, size_t dim, int dtype, void *stream);

#ifdef __cplusplus
}
#endif
#endif
