/*
 * fgnn_oracle.h -- CPU restatement of the GNNLab/SamGraph sampling-and-extraction hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may link or call it.  The shipped path is the HIP library under
 * fgnn-artifacts_amd/csrc; it never falls back to anything in this directory.
 *
 * Every function cites the reference file:line (relative to the reference tree's
 * samgraph/common/) whose behaviour it restates.  Two "RNG modes" exist because the reference
 * itself has two random sources:
 *   FGNN_RNG_MT_CPU_TWIN : the reference's CPU twin -- one default-seeded std::mt19937 stream
 *                          consumed through libstdc++ std::uniform_int_distribution<uint32_t>
 *                          (cpu/cpu_random.cc:26-30).  Used to pin this restatement against the
 *                          reference's own compiled CPU objects (oracle/_ref, tests/golden).
 *   FGNN_RNG_PHILOX      : the build's reproducible replacement for the reference GPU path's
 *                          clock-seeded cuRAND XORWOW (cuda/cuda_random_states.cu:105-107):
 *                          counter-based Philox4x32-10 addressed by
 *                          (seed, batch key, tag, item index, draw index), `draw % n` exactly
 *                          where the CUDA kernels do `curand() % n`.
 * The sampling/dedup/remap/cache/gather algorithms are one code path for both modes.
 */
#ifndef FGNN_ORACLE_H
#define FGNN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FGNN_EMPTY_KEY 0xFFFFFFFFu /* Constant::kEmptyKey, constant.h:71 */

/* SampleType values, common.h:50-58 */
enum {
  FGNN_KHOP0 = 0,
  FGNN_KHOP1 = 1,
  FGNN_WEIGHTED_KHOP = 2,
  FGNN_RANDOM_WALK = 3,
  FGNN_WEIGHTED_KHOP_PREFIX = 4,
  FGNN_KHOP2 = 5,
  FGNN_WEIGHTED_KHOP_HASH_DEDUP = 6
};

/* DataType values, common.h:38-46 */
enum { FGNN_F32 = 0, FGNN_F64 = 1, FGNN_F16 = 2, FGNN_U8 = 3, FGNN_I32 = 4, FGNN_I8 = 5, FGNN_I64 = 6 };

enum { FGNN_RNG_MT_CPU_TWIN = 0, FGNN_RNG_PHILOX = 1 };

/* ---------------------------------------------------------------- RNG ---------------------- */

/* Philox4x32-10 (Salmon et al., SC'11; the same generator cuRAND/rocRAND ship as "Philox").  */
void fgnn_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* The build's draw addressing.  tag = (sample_type << 8) | layer; `item` = position of the seed in
 * the layer's input list.  One Philox block = counter (block, item, tag, lo32(batch_key)) under key
 * (lo32(seed), hi32(seed) ^ hi32(batch_key)); draw number j of an item is word (j & 3) of block
 * (j >> 2), so a consumer of consecutive draws pays one Philox evaluation per four draws. */
void fgnn_philox_draw(uint64_t seed, uint64_t batch_key, uint32_t tag, uint32_t item, uint32_t block,
                      uint32_t out[4]);
uint32_t fgnn_philox_u32(uint64_t seed, uint64_t batch_key, uint32_t tag, uint32_t item, uint32_t j);

/* std::mt19937 (default seed 5489) + libstdc++-11 uniform_int_distribution<uint32_t>(lo,hi). */
typedef struct {
  uint32_t mt[624];
  int idx;
} fgnn_mt19937;
void fgnn_mt19937_seed(fgnn_mt19937 *g, uint32_t seed);
uint32_t fgnn_mt19937_next(fgnn_mt19937 *g);
uint32_t fgnn_mt19937_uniform_int(fgnn_mt19937 *g, uint32_t lo, uint32_t hi); /* inclusive */

typedef struct {
  int mode;          /* FGNN_RNG_* */
  uint64_t seed;     /* philox */
  fgnn_mt19937 mt;   /* cpu twin */
} fgnn_rng;
void fgnn_rng_init(fgnn_rng *r, int mode, uint64_t seed);

/* ---------------------------------------------------------------- helpers ------------------ */

/* PredictNumNodes, common.cc:330-339 */
size_t fgnn_predict_num_nodes(size_t batch_size, const size_t *fanout, size_t num_fanout_to_comp);
/* TableSize, cuda/cuda_hashtable.cu:125-128 */
size_t fgnn_table_size(size_t num, size_t scale);

/* ---------------------------------------------------------------- samplers ----------------- */

/* cuda/cuda_sampling_khop0.cu:41-174 (GPU semantic) / cpu/cpu_sampling_khop0.cc:29-83 (twin).
 * Output: compacted COO, seed-major then slot order; out_src = seed id, out_dst = neighbour. */
void fgnn_oracle_sample_khop0(const uint32_t *indptr, const uint32_t *indices, const uint32_t *input,
                              size_t num_input, size_t fanout, uint32_t *out_src, uint32_t *out_dst,
                              size_t *num_out, fgnn_rng *rng, uint64_t batch_key, uint32_t layer);

/* cuda/cuda_sampling_khop2.cu:41-89 / cpu/cpu_sampling_khop2.cc:29-76.  MUTATES indices
 * (partial Fisher-Yates in place on the CSR row), exactly as the reference does. */
void fgnn_oracle_sample_khop2(const uint32_t *indptr, uint32_t *indices, const uint32_t *input,
                              size_t num_input, size_t fanout, uint32_t *out_src, uint32_t *out_dst,
                              size_t *num_out, fgnn_rng *rng, uint64_t batch_key, uint32_t layer);

/* cuda/cuda_sampling_weighted_khop_prefix.cu:41-92 + host fn 148-255: with replacement via
 * binary search of the per-row inclusive prefix-sum table, then a STABLE sort by src id and
 * removal of adjacent duplicate (src,dst) pairs.  Philox mode only (the CPU twin is an empty stub,
 * cpu/cpu_sampling_weighted_khop.cc:24-27). */
void fgnn_oracle_sample_weighted_khop_prefix(const uint32_t *indptr, const uint32_t *indices,
                                             const float *prob_prefix, const uint32_t *input,
                                             size_t num_input, size_t fanout, uint32_t *out_src,
                                             uint32_t *out_dst, size_t *num_out, fgnn_rng *rng,
                                             uint64_t batch_key, uint32_t layer);

/* cuda/cuda_sampling_khop1.cu:42-234: uniform WITH replacement (curand() % len per slot), then the same stable
 * sort by src + adjacent-duplicate removal.  Philox mode only. */
void fgnn_oracle_sample_khop1(const uint32_t *indptr, const uint32_t *indices, const uint32_t *input,
                              size_t num_input, size_t fanout, uint32_t *out_src, uint32_t *out_dst, size_t *num_out,
                              fgnn_rng *rng, uint64_t batch_key, uint32_t layer);

/* cuda/cuda_sampling_weighted_khop.cu:41-236: alias method with replacement (alias table holds NODE IDS,
 * utility/data-process/toolkit/weight/create_alias_table.cc:150-151), then sort + adjacent dedup. */
void fgnn_oracle_sample_weighted_khop(const uint32_t *indptr, const uint32_t *indices, const float *prob_table,
                                      const uint32_t *alias_table, const uint32_t *input, size_t num_input,
                                      size_t fanout, uint32_t *out_src, uint32_t *out_dst, size_t *num_out,
                                      fgnn_rng *rng, uint64_t batch_key, uint32_t layer);

/* cuda/cuda_sampling_weighted_khop_hash_dedup.cu:41-111: alias draws, per-seed rejection of already selected
 * VALUES until `fanout` distinct ones are found, padded + count + compact (no sort).  See the .c file for the
 * termination rule the reference lacks. */
#define FGNN_HASH_DEDUP_MAX_FANOUT 50u               /* the reference's per-thread table has 50 slots */
#define FGNN_HASH_DEDUP_MAX_ATTEMPTS(f) (64u * (uint32_t)(f))
void fgnn_oracle_sample_weighted_khop_hash_dedup(const uint32_t *indptr, const uint32_t *indices,
                                                 const float *prob_table, const uint32_t *alias_table,
                                                 const uint32_t *input, size_t num_input, size_t fanout,
                                                 uint32_t *out_src, uint32_t *out_dst, size_t *num_out, fgnn_rng *rng,
                                                 uint64_t batch_key, uint32_t layer);

/* cuda/cuda_sampling_random_walk.cu:43-109 + cuda/cuda_frequency_hashmap.cu:1143-1367:
 * num_walks restart-walks of walk_len steps per seed, visit-frequency top-K per seed.
 * Tie rule fixed to (count desc, first visit order asc) -- a legal outcome of the reference's race.
 * out_src = seed id, out_dst = visited node, out_data = visit count. */
void fgnn_oracle_sample_random_walk(const uint32_t *indptr, const uint32_t *indices,
                                    const uint32_t *input, size_t num_input, size_t walk_len,
                                    double restart_prob, size_t num_walks, size_t K,
                                    uint32_t *out_src, uint32_t *out_dst, uint32_t *out_data,
                                    size_t *num_out, fgnn_rng *rng, uint64_t batch_key,
                                    uint32_t layer);

/* ---------------------------------------------------------------- dedup / remap ------------ */

/* OrderedHashTable (cuda/cuda_hashtable.{h,cu}) == CPUHashTable2 (cpu/cpu_hashtable2.cc:53-194)
 * run with one thread: dedup by FIRST occurrence, local id = rank of first occurrence. */
typedef struct {
  uint32_t *o2n;       /* [num_node] global -> local, FGNN_EMPTY_KEY if absent (direct index) */
  uint32_t *n2o;       /* [capacity] local -> global */
  size_t num_node;
  size_t capacity;
  size_t num_items;
} fgnn_oracle_ht;

fgnn_oracle_ht *fgnn_oracle_ht_create(size_t num_node, size_t capacity);
void fgnn_oracle_ht_destroy(fgnn_oracle_ht *ht);
void fgnn_oracle_ht_reset(fgnn_oracle_ht *ht); /* cuda_hashtable.cu:714-723 / cpu_hashtable2.cc:176-184 */
/* FillWithUnique, cuda_hashtable.cu:149-174,1017-1037.  Returns 0, or -1 if an item repeats. */
int fgnn_oracle_ht_fill_unique(fgnn_oracle_ht *ht, const uint32_t *items, size_t n);
/* FillWithDuplicates, cuda_hashtable.cu:130-147,176-211,386-438,725-807. unique gets n2o[0:num_unique]. */
void fgnn_oracle_ht_fill_duplicates(fgnn_oracle_ht *ht, const uint32_t *items, size_t n,
                                    uint32_t *unique, size_t *num_unique);
/* GPUMapEdges, cuda/cuda_mapping.cu:31-81 */
void fgnn_oracle_map_edges(const fgnn_oracle_ht *ht, const uint32_t *src, const uint32_t *dst, size_t n,
                           uint32_t *new_src, uint32_t *new_dst);

/* ---------------------------------------------------------------- batch driver ------------- */

typedef struct {
  uint32_t *row;  /* local id of the sampled neighbour (TrainGraph::row, cuda_loops.cc:222) */
  uint32_t *col;  /* local id of the seed            (TrainGraph::col, cuda_loops.cc:218) */
  uint32_t *data; /* random-walk visit counts or NULL */
  size_t num_src, num_dst, num_edge;
} fgnn_oracle_graph;

typedef struct {
  size_t num_layers;
  fgnn_oracle_graph *graphs;  /* graphs[i] for fanout[i]; sampled from i = L-1 down to 0 */
  uint32_t *input_nodes;      /* final unique list */
  size_t num_input_nodes;
  size_t total_edges;         /* kLogL1NumSample, cuda_loops.cc:262 */
} fgnn_oracle_task;

typedef struct {
  int sample_type;
  size_t num_layers;
  const size_t *fanout;
  /* random walk */
  size_t walk_len, num_walks, num_neighbor;
  double restart_prob;
  const uint32_t *alias_table; /* FGNN_WEIGHTED_KHOP only (prob table goes in the prob_prefix argument) */
} fgnn_oracle_sample_cfg;

/* DoGPUSample, cuda/cuda_loops.cc:50-267 (== dist/dist_loops.cc:51-269, cpu/cpu_loops.cc:55-191). */
fgnn_oracle_task *fgnn_oracle_do_sample(const uint32_t *indptr, uint32_t *indices,
                                        const float *prob_prefix, const fgnn_oracle_sample_cfg *cfg,
                                        fgnn_oracle_ht *ht, const uint32_t *seeds, size_t num_seeds,
                                        fgnn_rng *rng, uint64_t batch_key);
void fgnn_oracle_task_free(fgnn_oracle_task *t);

/* ---------------------------------------------------------------- cache + gather ----------- */

/* Direct-map cache table, cuda/cuda_cache_manager_host.cc:80-100, dist/dist_engine.cc:193-229 */
void fgnn_oracle_cache_table_build(const uint32_t *ranking_nodes, size_t num_cached, size_t num_node,
                                   uint32_t *table);
/* GetMissCacheIndex, cuda/cuda_cache.cu:33-234: stable two-way partition (the host body of the same split,
 * dist/dist_cache_manager_host.cc:129-155, gives it with one OpenMP thread; that file needs cuda_runtime.h through
 * function.h and cannot be compiled into oracle/_ref: pinned by reading). */
void fgnn_oracle_get_miss_cache_index(const uint32_t *table, const uint32_t *nodes, size_t n,
                                      uint32_t *miss_src, uint32_t *miss_dst, size_t *num_miss,
                                      uint32_t *cache_src, uint32_t *cache_dst, size_t *num_cache);
/* GPUExtract / CPUExtract, cuda/cuda_extraction.cu:30-117, cpu/cpu_extraction.cc:31-116 */
void fgnn_oracle_extract(void *dst, const void *src, const uint32_t *index, size_t num_index, size_t dim,
                         int dtype);
/* GPUMockExtract / CPUMockExtract, cuda/cuda_extraction.cu:50-70, cpu/cpu_extraction.cc:44-62, 92-116 */
void fgnn_oracle_mock_extract(void *dst, const void *src, const uint32_t *index, size_t num_index, size_t dim,
                              int dtype, unsigned empty_feat_bits);
/* CombineMissData / CombineCacheData, cuda/cuda_cache_manager_device.cu:165-210 */
void fgnn_oracle_combine(void *out, const void *rows, const uint32_t *src_index /* NULL => i */,
                         const uint32_t *dst_index, size_t n, size_t dim, int dtype);

/* Presample ranking, dist/pre_sampler.cc:131-162: sort desc of (freq << 32 | node). */
void fgnn_oracle_presample_rank(const uint32_t *freq, size_t num_node, uint32_t *ranking_nodes);

/* GPUExtractNeighbour, cuda/cuda_extract_neighbour.cu:41-169: every neighbour of every input node (count_edge ->
 * ExclusiveSum -> compact_edge).  The reference emits in a tile-internal order (thread t of a 1024-item tile writes the
 * rows of items t, t+256, ... back to back); its only consumers feed the list to a dedup, so the restatement uses input
 * order, CSR order within a row.  out == NULL: count only.  Returns the number of neighbours. */
size_t fgnn_oracle_extract_neighbour(const uint32_t *indptr, const uint32_t *indices, const uint32_t *input,
                                     size_t num_input, uint32_t *out);
/* DoGPUSampleAllNeighbour, cuda/cuda_loops.cc:500-571 (the kCacheByPreSampleStatic pre-sampler's "sampler",
 * cuda/pre_sampler.cc:69-71): FillWithUnique(seeds); per layer GPUExtractNeighbour of ALL nodes seen so far ->
 * FillWithDupMutable; input_nodes = every node seen (the closed num_layers-hop neighbourhood of the seeds), here in
 * first-occurrence order.  input_nodes holds up to num_node ids.  Returns their number. */
size_t fgnn_oracle_sample_all_neighbour(const uint32_t *indptr, const uint32_t *indices, const uint32_t *seeds,
                                        size_t num_seeds, size_t num_layers, size_t num_node, uint32_t *input_nodes);

/* ---------------------------------------------------------------- shufflers ---------------- */

/* The shufflers' explicit Fisher-Yates: for i = n-1..1: swap(data[i], data[d(0,i)(g)]) with
 * g = std::default_random_engine(seed) (= minstd_rand0) and libstdc++-11
 * std::uniform_int_distribution<size_t>; cuda/cuda_shuffler.cc:84-107 and cpu/cpu_shuffler.cc:75-98
 * (seed = wall clock), dist/dist_shuffler.cc:108-131 (seed = epoch).  In place, cumulative across
 * epochs exactly like the reference (the array is never reset). */
void fgnn_oracle_shuffle_minstd0(uint32_t *data, size_t n, uint64_t seed);

/* DistShuffler step partition, dist/dist_shuffler.cc:36-79: for sampler `sampler_id` of
 * `num_sampler`, the offset (in items) into the shuffled train set, its number of local steps and
 * the size of its last batch. */
void fgnn_oracle_dist_shuffler_partition(size_t num_data, size_t batch_size, int sampler_id,
                                         int num_sampler, size_t *dataset_offset,
                                         size_t *num_local_step, size_t *local_data_size,
                                         size_t *last_batch_size, size_t *epoch_step);

/* DistAlignedShuffler (dist/dist_shuffler_aligned.cc:36-146): the arch6 / arch7 split */
void fgnn_oracle_aligned_shuffler_partition(size_t num_data, size_t batch_size, size_t worker_id,
                                            size_t num_worker, size_t *padded_size,
                                            size_t *data_per_worker, size_t *num_local_step,
                                            size_t *num_global_step, size_t *global_step_offset,
                                            size_t *global_data_offset, size_t *last_batch_size);

size_t fgnn_dtype_bytes(int dtype);

/* ---------------------------------------------------------------- OpenMP CPU baseline -------
 * The reference's multi-threaded CPU path (khop2 + CPUHashTable2 + CPUExtract, cpu/cpu_loops.cc:55-227),
 * for bench.py's cpu_baseline timing only; see fgnn_oracle.c. */
typedef struct fgnn_omp_ctx fgnn_omp_ctx;
fgnn_omp_ctx *fgnn_omp_create(size_t num_node, size_t capacity, int threads);
void fgnn_omp_destroy(fgnn_omp_ctx *c);
size_t fgnn_omp_sample_batch(fgnn_omp_ctx *c, const uint32_t *indptr, uint32_t *indices, const uint32_t *seeds,
                             size_t num_seeds, const size_t *fanout, size_t num_layers, const float *feat,
                             size_t feat_dim, uint32_t feat_row_mask, float *feat_out, size_t *num_input_nodes);

#ifdef __cplusplus
}
#endif
#endif
