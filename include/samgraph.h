/*
 * samgraph.h -- the drop-in boundary: the `samgraph_*` C ABI of the reference
 * (samgraph/common/operation.h:27-109) as exported by this repo's engine library
 * fgnn-artifacts_amd/samgraph/torch/c_lib.so, which the reference's Python layer binds with
 * ctypes.CDLL (samgraph/common/__init__.py:268-341).  Same names, argument meaning and error
 * behaviour: no return codes; a violated check prints file:line and abort()s
 * (logging.h:32-45, logging.cc:72); the parent notices through samgraph_wait_one_child.
 *
 * Behind it the hot path runs as HIP kernels on MI355X (include/fgnn_hip.h); there is no CPU
 * fallback: a sampler/trainer context that is not "cuda:N" aborts.
 *
 * Supported run architectures (RunArch, common.h:70-79): arch1 (one GPU samples and extracts,
 * cuda/cuda_loops_arch1.cc), arch2 / arch3 / arch4 (one process, sampler context + trainer context, optional
 * background threads via samgraph_start; arch3 is the default of the reference's single-process scripts;
 * cuda/cuda_loops_arch{2,3,4}.cc -- including arch4's dynamic-cache prototype, DoGPUSampleDyCache +
 * DoDynamicCacheFeatureCopy, selected by _cache_policy = dynamic_cache), arch5 (FGNN: sampler processes +
 * trainer processes linked by the pinned host queue, dist/dist_engine.cc, dist/dist_loops_arch5.cc), arch6 (the
 * reference's SGNN baseline: samgraph_data_init in the parent, samgraph_sample_init + samgraph_train_init in every
 * worker process, each worker samples and extracts its equal share of the train set, dist/dist_loops_arch6.cc,
 * dist/dist_shuffler_aligned.cc) and arch7 (a sample-only engine per worker: samgraph_config with worker_id /
 * num_worker + samgraph_init in every worker, cuda/cuda_loops_arch7.cc).  Sample types: all seven (khop0,
 * khop1, khop2, weighted_khop, weighted_khop_hash_dedup, weighted_khop_prefix, random_walk).  Cache policies: every
 * _cache_policy value of the reference -- pre_sample (computed at sample_init, dist/pre_sampler.cc), presample_static
 * (whole-neighbourhood frequencies, cuda/pre_sampler.cc:69-71; refused by the multi-process engine like
 * dist/pre_sampler.cc:87-88), dynamic_cache (arch4) and the file-backed rankings (cache_by_*.bin, engine.cc:216-256).
 * SAMGRAPH_SANITY_CHECK=1 checks every batch of seeds on the GPU (cuda/cuda_sanity_check.cu:28-88).
 */
#ifndef SAMGRAPH_H
#define SAMGRAPH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* operation.h:29-31, operation.cc:45-169.  Required keys: dataset_path, _arch, _sample_type,
 * batch_size, num_epoch, _cache_policy, cache_percentage, max_sampling_jobs, max_copying_jobs,
 * omp_thread_num; arch1-4: sampler_ctx, trainer_ctx; arch5: num_sample_worker, num_train_worker,
 * [have_switcher]; arch6: num_worker; arch7: worker_id, num_worker, sampler_ctx, trainer_ctx; k-hop: num_fanout, fanout; random walk: random_walk_length,
 * random_walk_restart_prob, num_random_walk, num_neighbor, num_layer; optional barriered_epoch,
 * presample_epoch.  Unknown keys are ignored.  Extension keys of this build (ignored by the
 * reference): seed (Philox seed, default 0x5A4D47). */
void samgraph_config(const char **config_keys, const char **config_values, const size_t num_config_items);

void samgraph_init(void);     /* operation.h:32, single-process archs (arch1-4) and every arch7 worker */
void samgraph_start(void);    /* operation.h:34 */
void samgraph_shutdown(void); /* operation.h:36 */

size_t samgraph_num_epoch(void);       /* operation.h:38 */
size_t samgraph_steps_per_epoch(void); /* operation.h:40 */
size_t samgraph_num_class(void);       /* operation.h:42 */
size_t samgraph_feat_dim(void);        /* operation.h:44 */

uint64_t samgraph_get_next_batch(void); /* operation.h:46: blocks on the GraphPool, returns the batch key */
void samgraph_sample_once(void);        /* operation.h:48 */

size_t samgraph_get_graph_num_src(uint64_t key, int graph_id);  /* operation.h:50 */
size_t samgraph_get_graph_num_dst(uint64_t key, int graph_id);  /* operation.h:52 */
size_t samgraph_get_graph_num_edge(uint64_t key, int graph_id); /* operation.h:54 */

void samgraph_log_step(uint64_t epoch, uint64_t step, int item, double val);     /* operation.h:56 */
void samgraph_log_step_add(uint64_t epoch, uint64_t step, int item, double val); /* operation.h:58 */
void samgraph_log_epoch_add(uint64_t epoch, int item, double val);               /* operation.h:60 */
double samgraph_get_log_init_value(int item);                                    /* operation.cc:267 */
double samgraph_get_log_step_value(uint64_t epoch, uint64_t step, int item);     /* operation.h:62 */
double samgraph_get_log_epoch_value(uint64_t epoch, int item);                   /* operation.h:64 */

void samgraph_report_init(void);                                   /* operation.h:66 */
void samgraph_report_step(uint64_t epoch, uint64_t step);          /* operation.h:68 */
void samgraph_report_step_average(uint64_t epoch, uint64_t step);  /* operation.h:70 */
void samgraph_report_epoch(uint64_t epoch);                        /* operation.h:72 */
void samgraph_report_epoch_average(uint64_t epoch);                /* operation.h:74 */
void samgraph_report_node_access(void);                            /* operation.h:76 */

void samgraph_trace_step_begin(uint64_t key, int item, uint64_t ts); /* operation.h:78 */
void samgraph_trace_step_end(uint64_t key, int item, uint64_t ts);   /* operation.h:80 */
void samgraph_trace_step_begin_now(uint64_t key, int item);          /* operation.h:82 */
void samgraph_trace_step_end_now(uint64_t key, int item);            /* operation.h:84 */
void samgraph_dump_trace(void);                                      /* operation.h:86 */

void samgraph_forward_barrier(void); /* operation.h:88 */

/* multi-process (arch5) */
void samgraph_data_init(void);                               /* operation.h:91: before fork */
void samgraph_sample_init(int worker_id, const char *ctx);   /* operation.h:93 */
void samgraph_train_init(int worker_id, const char *ctx);    /* operation.h:95 */
void samgraph_extract_start(int count);                      /* operation.h:101 */
void samgraph_switch_init(int worker_id, const char *ctx, double cache_percentage); /* operation.h:104 */
size_t samgraph_num_local_step(void);                        /* operation.h:106 */
int samgraph_wait_one_child(void);                           /* operation.h:108 */

/* ---- tensor getters ---------------------------------------------------------------------------
 * The reference exposes these through pybind11 as torch tensors built with torch::from_blob
 * (samgraph/torch/adapter.h:29-42, adapter.cc:48-192).  Here they are plain C: a device (or host)
 * pointer plus element count, which samgraph/torch/adapter.py wraps without copying.  All check
 * key == current batch key (adapter.cc:52) and abort on mismatch.
 * The library is ALSO the Python extension module the reference's adapter.py imports (`from samgraph.torch import c_lib`,
 * adapter.py:26): PyInit_c_lib (csrc/engine/eng_pymodule.cc) registers samgraph_torch_get_graph_feat(key), ..._label,
 * ..._row(key, layer), ..._col, ..._data, ..._get_dataset_feat(), ..._get_dataset_label(), ..._graph_input_nodes(key),
 * ..._graph_output_nodes(key) -- the names of adapter.cc:177-189 -- as zero-copy wrappers over the getters below,
 * without linking libpython or libtorch (the entry points are resolved with dlsym when an interpreter imports it). */
const void *samgraph_torch_get_graph_feat_ptr(uint64_t key, size_t *num_rows, size_t *dim, int *dtype, int *device);
const void *samgraph_torch_get_graph_label_ptr(uint64_t key, size_t *num, int *dtype, int *device);
const uint32_t *samgraph_torch_get_graph_row_ptr(uint64_t key, int layer, size_t *num, int *device);
const uint32_t *samgraph_torch_get_graph_col_ptr(uint64_t key, int layer, size_t *num, int *device);
const uint32_t *samgraph_torch_get_graph_data_ptr(uint64_t key, int layer, size_t *num, int *device);
const uint32_t *samgraph_torch_get_graph_input_nodes_ptr(uint64_t key, size_t *num, int *device);
const uint32_t *samgraph_torch_get_graph_output_nodes_ptr(uint64_t key, size_t *num, int *device);
const void *samgraph_torch_get_dataset_feat_ptr(size_t *num_rows, size_t *dim, int *dtype);   /* host */
const void *samgraph_torch_get_dataset_label_ptr(size_t *num, int *dtype);                    /* host */

#ifdef __cplusplus
}
#endif
#endif
