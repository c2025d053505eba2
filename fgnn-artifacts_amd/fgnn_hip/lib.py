"""ctypes binding of fgnn-artifacts_amd/lib/libfgnn_hip.so (include/fgnn_hip.h).

PyTorch is used only as the owner of device memory and streams: every call below takes torch CUDA
(ROCm) tensors, passes their raw device pointers through the C ABI and enqueues HIP kernels on the
current torch stream.  There is NO fallback: if the library is missing, or no GPU is present, the
calls raise.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product library.  Nothing in the environment changes which file is loaded: the measurement tools under tools/ that
# want the profiling build (lib/libfgnn_hip_prof.so, `make -C csrc prof`: the only build whose kernels read A/B switches
# and ablation masks) say so in their own code, with use_library(path), before anything calls load()
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libfgnn_hip.so")
PROF_LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libfgnn_hip_prof.so")


def use_library(path):
    """tools/ only: bind another build of the kernel library (before the first call of load())."""
    global LIB_PATH
    if _lib is not None:
        raise FgnnError("use_library() after the kernel library has been loaded")
    LIB_PATH = os.path.abspath(path)


SRC_GLOBAL, SRC_LOCAL = 0, 1
LINK_WGS_SHARED, LINK_WGS_DEDICATED = 16, 16  # FGNN_LINK_WGS_* (fgnn_hip.h)
EMPTY = 0xFFFFFFFF
F32, F64, F16, U8, I32, I8, I64 = range(7)
_T2DT = {torch.float32: F32, torch.float64: F64, torch.float16: F16, torch.uint8: U8, torch.int32: I32,
         torch.int8: I8, torch.int64: I64}

EXPORTS = [
    "fgnn_version", "fgnn_last_error", "fgnn_device_count", "fgnn_debug_phase_log_bytes", "fgnn_debug_phase_log", "fgnn_debug_occupy", "fgnn_debug_scan_helps", "fgnn_debug_set_scan_help_after", "fgnn_debug_set_partition_lds_limit", "fgnn_debug_sort_pairs", "fgnn_debug_random_reads", "fgnn_scratch_bytes", "fgnn_sanity_map_bytes", "fgnn_sanity_check_batch", "fgnn_sample_khop0", "fgnn_sample_khop2",
    "fgnn_weighted_scratch_bytes", "fgnn_sample_weighted_khop_prefix", "fgnn_random_walk_scratch_bytes",
    "fgnn_sample_random_walk", "fgnn_sample_khop1", "fgnn_sample_weighted_khop",
    "fgnn_hash_dedup_scratch_bytes", "fgnn_sample_weighted_khop_hash_dedup",
    "fgnn_hashtable_create", "fgnn_hashtable_create_ex", "fgnn_hashtable_destroy", "fgnn_hashtable_capacity", "fgnn_hashtable_reset",
    "fgnn_hashtable_fill_unique", "fgnn_hashtable_fill_duplicates", "fgnn_hashtable_map", "fgnn_hashtable_n2o",
    "fgnn_hashtable_d_num_items", "fgnn_hashtable_set_n2o", "fgnn_hashtable_start_batch",
    "fgnn_extract_neighbour_scratch_bytes", "fgnn_extract_neighbour", "fgnn_neighbourhood_expand",
    "fgnn_presample_count", "fgnn_presample_rank_scratch_bytes", "fgnn_presample_rank", "fgnn_cache_table_build",
    "fgnn_cache_table_replace", "fgnn_get_miss_cache_index", "fgnn_gather_rows", "fgnn_gather_rows_masked", "fgnn_gather_rows_shared", "fgnn_extract_fused", "fgnn_extract_fused_grid", "fgnn_extract_fused_link_grid", "fgnn_block_aggregate", "fgnn_block_aggregate_ex", "fgnn_batch_set_feat_row_mask",
    "fgnn_sage_finish_z", "fgnn_sage_grad_prep", "fgnn_relu_dropout", "fgnn_relu_dropout_backward",
    "fgnn_softmax_xent_scratch_bytes", "fgnn_softmax_xent", "fgnn_adam_step",
]

_lib = None


class FgnnError(RuntimeError):
    pass


def load():
    """Loads the HIP library; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FgnnError(f"{LIB_PATH} is missing: build it with `make -C fgnn-artifacts_amd/csrc` "
                            "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.fgnn_version.restype = C.c_char_p
        L.fgnn_last_error.restype = C.c_char_p
        L.fgnn_debug_scan_helps.restype = C.c_ulonglong
        L.fgnn_debug_phase_log_bytes.restype = C.c_size_t
        L.fgnn_debug_phase_log.restype = None
        L.fgnn_debug_phase_log.argtypes = [C.c_void_p]
        L.fgnn_scratch_bytes.restype = C.c_size_t
        L.fgnn_scratch_bytes.argtypes = [C.c_size_t]
        L.fgnn_extract_neighbour_scratch_bytes.restype = C.c_size_t
        L.fgnn_extract_neighbour_scratch_bytes.argtypes = [C.c_size_t]
        L.fgnn_hashtable_create.restype = C.c_void_p
        L.fgnn_hashtable_create.argtypes = [C.c_size_t, C.POINTER(C.c_int)]
        L.fgnn_hashtable_destroy.argtypes = [C.c_void_p]
        L.fgnn_hashtable_capacity.restype = C.c_size_t
        L.fgnn_hashtable_capacity.argtypes = [C.c_void_p]
        L.fgnn_hashtable_n2o.restype = C.c_void_p
        L.fgnn_hashtable_n2o.argtypes = [C.c_void_p]
        L.fgnn_hashtable_d_num_items.restype = C.c_void_p
        L.fgnn_hashtable_d_num_items.argtypes = [C.c_void_p]
        L.fgnn_debug_set_scan_help_after.restype = None
        L.fgnn_debug_set_scan_help_after.argtypes = [C.c_int]
        # (neither the library nor this binding reads a switch from the environment: tests that force the helping path of
        # the single-pass kernels call fgnn_debug_set_scan_help_after themselves, tests/conftest.py)
        _lib = L
    return _lib


def _check(code, what):
    if code != 0:
        detail = load().fgnn_last_error().decode() if code == -3 else ""
        raise FgnnError(f"{what} failed with code {code} {detail}")


class DevicePointer:
    """a raw device-visible address standing in for a tensor argument (registered host memory whose device address is
    not its host address); `keep` holds whatever owns the memory"""

    def __init__(self, address, keep=None):
        self.address, self.keep = int(address), keep

    def data_ptr(self):
        return self.address

    is_cuda = True


def _ptr(t):
    return C.c_void_p(t.data_ptr() if t is not None else 0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise FgnnError("fgnn_hip kernels need device tensors (no CPU fallback)")


def _as_u32(t):
    """uint32 storage viewed through torch.int32 (torch has no native uint32 arithmetic)."""
    assert t.dtype in (torch.int32, torch.uint32) and t.is_contiguous()
    return t


def scratch(n_cap, device):
    return torch.empty(load().fgnn_scratch_bytes(int(n_cap)), dtype=torch.uint8, device=device)


def sample_khop(kind, indptr, indices, inp, fanout, seed, batch_key, layer, src_mode=SRC_GLOBAL, d_num_input=None,
                ws=None):
    """kind in {'khop0','khop2'}.  Returns (out_src, out_dst, d_num_out) -- device tensors, sized for the
    worst case num_input*fanout; d_num_out is an int64[1] device scalar.  Asynchronous."""
    L = load()
    _need_gpu(indptr, indices, inp)
    n = inp.numel()
    dev = inp.device
    out_src = torch.empty(max(n * fanout, 1), dtype=torch.int32, device=dev)
    out_dst = torch.empty(max(n * fanout, 1), dtype=torch.int32, device=dev)
    d_num_out = torch.zeros(1, dtype=torch.int64, device=dev)
    if ws is None:
        ws = scratch(n, dev)
        if kind == "khop0":  # room for the hub tables: rows beyond 16 K entries are then drawn by the whole chip
            ws = torch.empty(ws.numel() + 4 * (n + 16 + 1024 * (fanout + 3)), dtype=torch.uint8, device=dev)
    fn = L.fgnn_sample_khop0 if kind == "khop0" else L.fgnn_sample_khop2
    code = fn(_ptr(indptr), _ptr(indices), _ptr(inp), C.c_size_t(n), _ptr(d_num_input), C.c_size_t(n),
              C.c_size_t(fanout), _ptr(out_src), _ptr(out_dst), _ptr(d_num_out), C.c_int(src_mode), C.c_uint64(seed),
              C.c_uint64(batch_key), C.c_uint32(layer), _ptr(ws), C.c_size_t(ws.numel()), _stream())
    _check(code, "fgnn_sample_" + kind)
    return out_src, out_dst, d_num_out


def debug_sort_pairs(keys, vals):
    """fgnn_debug_sort_pairs: (keys, vals) int32 CUDA tensors sorted in place by key (as uint32), stable."""
    _need_gpu(keys, vals)
    assert keys.numel() == vals.numel()
    _check(load().fgnn_debug_sort_pairs(_ptr(keys), _ptr(vals), C.c_size_t(keys.numel()), _stream()),
           "fgnn_debug_sort_pairs")


def sample_weighted_khop_prefix(indptr, indices, prefix, inp, fanout, seed, batch_key, layer, src_mode=SRC_GLOBAL,
                                d_num_input=None):
    L = load()
    _need_gpu(indptr, indices, prefix, inp)
    L.fgnn_weighted_scratch_bytes.restype = C.c_size_t
    n, dev = inp.numel(), inp.device
    out_src = torch.empty(max(n * fanout, 1), dtype=torch.int32, device=dev)
    out_dst = torch.empty(max(n * fanout, 1), dtype=torch.int32, device=dev)
    d_num_out = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = torch.empty(L.fgnn_weighted_scratch_bytes(C.c_size_t(max(n, 1)), C.c_size_t(fanout)), dtype=torch.uint8,
                     device=dev)
    _check(L.fgnn_sample_weighted_khop_prefix(_ptr(indptr), _ptr(indices), _ptr(prefix), _ptr(inp), C.c_size_t(n),
                                              _ptr(d_num_input), C.c_size_t(n), C.c_size_t(fanout), _ptr(out_src),
                                              _ptr(out_dst), _ptr(d_num_out), C.c_int(src_mode), C.c_uint64(seed),
                                              C.c_uint64(batch_key), C.c_uint32(layer), _ptr(ws),
                                              C.c_size_t(ws.numel()), _stream()), "fgnn_sample_weighted_khop_prefix")
    return out_src, out_dst, d_num_out


def sample_with_replacement(kind, indptr, indices, inp, fanout, seed, batch_key, layer, prob=None, alias=None,
                            src_mode=SRC_GLOBAL):
    """kind in {'khop1', 'weighted_khop' (alias method), 'weighted_khop_hash_dedup'}."""
    L = load()
    _need_gpu(indptr, indices, inp)
    L.fgnn_weighted_scratch_bytes.restype = C.c_size_t
    n, dev = inp.numel(), inp.device
    out_src = torch.empty(max(n * fanout, 1), dtype=torch.int32, device=dev)
    out_dst = torch.empty(max(n * fanout, 1), dtype=torch.int32, device=dev)
    d_num_out = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = torch.empty(L.fgnn_weighted_scratch_bytes(C.c_size_t(max(n, 1)), C.c_size_t(fanout)), dtype=torch.uint8,
                     device=dev)
    tail = (_ptr(inp), C.c_size_t(n), C.c_void_p(0), C.c_size_t(n), C.c_size_t(fanout), _ptr(out_src), _ptr(out_dst),
            _ptr(d_num_out), C.c_int(src_mode), C.c_uint64(seed), C.c_uint64(batch_key), C.c_uint32(layer), _ptr(ws),
            C.c_size_t(ws.numel()), _stream())
    if kind == "khop1":
        _check(L.fgnn_sample_khop1(_ptr(indptr), _ptr(indices), *tail), "fgnn_sample_khop1")
    elif kind == "weighted_khop_hash_dedup":
        _check(L.fgnn_sample_weighted_khop_hash_dedup(_ptr(indptr), _ptr(indices), _ptr(prob), _ptr(alias), *tail),
               "fgnn_sample_weighted_khop_hash_dedup")
    else:
        _check(L.fgnn_sample_weighted_khop(_ptr(indptr), _ptr(indices), _ptr(prob), _ptr(alias), *tail),
               "fgnn_sample_weighted_khop")
    return out_src, out_dst, d_num_out


def sample_random_walk(indptr, indices, inp, walk_len, restart_prob, num_walks, K, seed, batch_key, layer,
                       src_mode=SRC_GLOBAL, d_num_input=None):
    L = load()
    _need_gpu(indptr, indices, inp)
    L.fgnn_random_walk_scratch_bytes.restype = C.c_size_t
    n, dev = inp.numel(), inp.device
    outs = [torch.empty(max(n * K, 1), dtype=torch.int32, device=dev) for _ in range(3)]
    d_num_out = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = torch.empty(L.fgnn_random_walk_scratch_bytes(C.c_size_t(max(n, 1)), C.c_size_t(K)), dtype=torch.uint8,
                     device=dev)
    _check(L.fgnn_sample_random_walk(_ptr(indptr), _ptr(indices), _ptr(inp), C.c_size_t(n), _ptr(d_num_input),
                                     C.c_size_t(n), C.c_size_t(walk_len), C.c_double(restart_prob),
                                     C.c_size_t(num_walks), C.c_size_t(K), _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]),
                                     _ptr(d_num_out), C.c_int(src_mode), C.c_uint64(seed), C.c_uint64(batch_key),
                                     C.c_uint32(layer), _ptr(ws), C.c_size_t(ws.numel()), _stream()),
           "fgnn_sample_random_walk")
    return outs[0], outs[1], outs[2], d_num_out


class HashTable:
    """OrderedHashTable (cuda_hashtable.h:99-149) on the GPU."""

    def __init__(self, max_items, device="cuda:0", max_fill_items=None):
        """max_fill_items: largest fill_duplicates call; when given, reset() is a generation bump (see fgnn_hip.h)."""
        L = load()
        torch.cuda.set_device(device)
        err = C.c_int(0)
        L.fgnn_hashtable_create_ex.restype = C.c_void_p
        if max_fill_items is None:
            self.h = C.c_void_p(L.fgnn_hashtable_create(C.c_size_t(max_items), C.byref(err)))
        else:
            self.h = C.c_void_p(L.fgnn_hashtable_create_ex(C.c_size_t(max_items), C.c_size_t(max_fill_items),
                                                           C.byref(err)))
        if not self.h:
            raise FgnnError(f"fgnn_hashtable_create failed with code {err.value}")
        self.max_items = max_items
        self.device = torch.device(device)

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.fgnn_hashtable_destroy(self.h)
            self.h = None

    def reset(self):
        _check(load().fgnn_hashtable_reset(self.h, _stream()), "fgnn_hashtable_reset")

    def fill_unique(self, items):
        _need_gpu(items)
        _check(load().fgnn_hashtable_fill_unique(self.h, _ptr(items), C.c_size_t(items.numel()), _stream()),
               "fgnn_hashtable_fill_unique")

    def fill_duplicates(self, items, num_items=None, d_num_items=None, want_mapped=True, ws=None):
        """Returns mapped (int32 device tensor, local id per item) or None."""
        _need_gpu(items)
        cap = items.numel()
        n = cap if num_items is None else num_items
        mapped = torch.empty(max(cap, 1), dtype=torch.int32, device=items.device) if want_mapped else None
        if ws is None:
            ws = scratch(cap, items.device)
        _check(load().fgnn_hashtable_fill_duplicates(self.h, _ptr(items), C.c_size_t(n), _ptr(d_num_items),
                                                     C.c_size_t(cap), _ptr(mapped), _ptr(ws), C.c_size_t(ws.numel()),
                                                     _stream()), "fgnn_hashtable_fill_duplicates")
        return mapped

    def map(self, items):
        _need_gpu(items)
        mapped = torch.empty(max(items.numel(), 1), dtype=torch.int32, device=items.device)
        _check(load().fgnn_hashtable_map(self.h, _ptr(items), C.c_size_t(items.numel()), C.c_void_p(0),
                                         C.c_size_t(items.numel()), _ptr(mapped), _stream()), "fgnn_hashtable_map")
        return mapped[:items.numel()]

    def num_items(self):
        """Synchronises."""
        p = load().fgnn_hashtable_d_num_items(self.h)
        t = _wrap_device_u32(p, 1, self.device)
        return int(t.cpu()[0])

    def d_num_items_ptr(self):
        return load().fgnn_hashtable_d_num_items(self.h)

    def unique(self, n=None):
        """Copy of the N2O list (first n entries; default all current items).  Synchronises if n is None."""
        if n is None:
            n = self.num_items()
        p = load().fgnn_hashtable_n2o(self.h)
        return _wrap_device_u32(p, n, self.device).clone()


class _DevArray:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def _wrap_device_u32(ptr, n, device):
    if n == 0:
        return torch.empty(0, dtype=torch.int32, device=device)
    return torch.as_tensor(_DevArray(ptr, n, "<i4"), device=device)


def extract_neighbour(indptr, indices, inp, out_cap, num_input=None, d_num_input=None):
    """GPUExtractNeighbour: returns (out[out_cap], d_num_out int64[1]) device tensors.  Asynchronous."""
    L = load()
    _need_gpu(indptr, indices, inp)
    cap = inp.numel()
    n = cap if num_input is None else num_input
    dev = inp.device
    out = torch.empty(max(out_cap, 1), dtype=torch.int32, device=dev)
    d_num_out = torch.full((1,), -1, dtype=torch.int64, device=dev)
    ws = torch.empty(L.fgnn_extract_neighbour_scratch_bytes(cap), dtype=torch.uint8, device=dev)
    _check(L.fgnn_extract_neighbour(_ptr(indptr), _ptr(indices), _ptr(inp), C.c_size_t(n), _ptr(d_num_input),
                                    C.c_size_t(cap), _ptr(out), C.c_size_t(out_cap), _ptr(d_num_out), _ptr(ws),
                                    C.c_size_t(ws.numel()), _stream()), "fgnn_extract_neighbour")
    return out, d_num_out


def neighbourhood_expand(indptr, indices, frontier, stamp, mark, freq, nxt, d_num_next, mark_frontier=False,
                         num_frontier=None, d_num_frontier=None):
    """One level of the closed-neighbourhood search (fgnn_neighbourhood_expand); d_num_next (int32[1]) is added to."""
    _need_gpu(indptr, indices, frontier, stamp, nxt, d_num_next)
    cap = frontier.numel()
    n = cap if num_frontier is None else num_frontier
    _check(load().fgnn_neighbourhood_expand(_ptr(indptr), _ptr(indices), _ptr(frontier), C.c_size_t(n),
                                            _ptr(d_num_frontier), C.c_size_t(cap), _ptr(stamp), C.c_uint32(mark),
                                            _ptr(freq), _ptr(nxt), C.c_size_t(nxt.numel()), _ptr(d_num_next),
                                            C.c_int(1 if mark_frontier else 0), _stream()), "fgnn_neighbourhood_expand")


class SanityChecker:
    """SAMGRAPH_SANITY_CHECK's per-epoch net (GPUSanityCheckList + GPUBatchSanityCheck, cuda_sanity_check.cu:28-88):
    check(batch) returns the flag word -- 0 = fine, 1 = an invalid id, 2 = an id seen before in this epoch (or twice in
    the batch), 4 = an id >= num_node.  new_epoch() clears the seen-bitmap."""

    def __init__(self, num_node, device, invalid_val=EMPTY):
        L = load()
        L.fgnn_sanity_map_bytes.restype = C.c_size_t
        self.num_node, self.invalid = num_node, invalid_val
        self.bits = torch.zeros(L.fgnn_sanity_map_bytes(C.c_size_t(num_node)) // 4, dtype=torch.int32, device=device)
        self.flags = torch.zeros(1, dtype=torch.int32, device=device)

    def new_epoch(self):
        self.bits.zero_()

    def check(self, batch, no_duplicates=True):
        _need_gpu(batch)
        self.flags.zero_()
        _check(load().fgnn_sanity_check_batch(_ptr(self.bits) if no_duplicates else None, C.c_size_t(self.num_node),
                                              _ptr(batch), C.c_size_t(batch.numel()), C.c_uint32(self.invalid),
                                              _ptr(self.flags), _stream()), "fgnn_sanity_check_batch")
        return int(self.flags.item())


def presample_count(freq, nodes, num_nodes=None, d_num_nodes=None):
    """PreSampler's counting step: freq[nodes[i]] += 1 (dist/pre_sampler.cc:117-131).  Asynchronous."""
    _need_gpu(freq, nodes)
    cap = nodes.numel()
    n = cap if num_nodes is None else num_nodes
    _check(load().fgnn_presample_count(_ptr(freq), _ptr(nodes), C.c_size_t(n), _ptr(d_num_nodes), C.c_size_t(cap),
                                       _stream()), "fgnn_presample_count")


def presample_rank(freq):
    """rank list (int32 device tensor, u32 storage): nodes by (frequency desc, id desc) (dist/pre_sampler.cc:140-160)."""
    L = load()
    _need_gpu(freq)
    L.fgnn_presample_rank_scratch_bytes.restype = C.c_size_t
    n = freq.numel()
    ws = torch.empty(L.fgnn_presample_rank_scratch_bytes(C.c_size_t(n)), dtype=torch.uint8, device=freq.device)
    rank = torch.empty(n, dtype=torch.int32, device=freq.device)
    _check(L.fgnn_presample_rank(_ptr(freq), C.c_size_t(n), _ptr(rank), _ptr(ws), C.c_size_t(ws.numel()), _stream()),
           "fgnn_presample_rank")
    return rank


def cache_table_build(rank, num_cached, num_node=None):
    """direct-map table node -> cache slot (SampleCacheTableInit, dist_engine.cc:193-229)"""
    _need_gpu(rank)
    n = rank.numel() if num_node is None else num_node
    table = torch.empty(n, dtype=torch.int32, device=rank.device)
    _check(load().fgnn_cache_table_build(_ptr(table), C.c_size_t(n), _ptr(rank), C.c_size_t(num_cached), _stream()),
           "fgnn_cache_table_build")
    return table


def cache_table_replace(table, old_nodes, new_nodes):
    """ReplaceCacheGPU's index update: old nodes -> EMPTY, new node i -> i"""
    _need_gpu(table, old_nodes, new_nodes)
    n_old = 0 if old_nodes is None else old_nodes.numel()
    _check(load().fgnn_cache_table_replace(_ptr(table), _ptr(old_nodes) if n_old else C.c_void_p(0), C.c_size_t(n_old),
                                           _ptr(new_nodes), C.c_size_t(new_nodes.numel()), _stream()),
           "fgnn_cache_table_replace")


def get_miss_cache_index(table, nodes, num_nodes=None, d_num_nodes=None, ws=None):
    """Returns (miss_src, miss_dst, cache_src, cache_dst, d_counts[2]) device tensors sized len(nodes)."""
    _need_gpu(table, nodes)
    cap = nodes.numel()
    n = cap if num_nodes is None else num_nodes
    dev = nodes.device
    outs = [torch.empty(max(cap, 1), dtype=torch.int32, device=dev) for _ in range(4)]
    d_counts = torch.zeros(2, dtype=torch.int32, device=dev)
    if ws is None:
        ws = scratch(cap, dev)
    _check(load().fgnn_get_miss_cache_index(_ptr(table), _ptr(nodes), C.c_size_t(n), _ptr(d_num_nodes),
                                            C.c_size_t(cap), _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]),
                                            _ptr(outs[3]), _ptr(d_counts), _ptr(ws), C.c_size_t(ws.numel()),
                                            _stream()), "fgnn_get_miss_cache_index")
    return outs[0], outs[1], outs[2], outs[3], d_counts


def gather_rows(out, src, src_index=None, dst_index=None, n=None, d_n=None, src_row_mask=None, shared_gpu=None):
    """out[dst_index[i] or i] = src[(src_index[i] or i) & src_row_mask]; rows are the trailing dims.
    shared_gpu (not None): fgnn_gather_rows_shared -- a host-source launch stays small on a GPU that also samples."""
    _need_gpu(out)
    if n is None:
        n = (src_index if src_index is not None else dst_index if dst_index is not None else src).shape[0]
    dim = 1
    for s in out.shape[1:]:
        dim *= s
    assert out.dtype == src.dtype and out.is_contiguous() and src.is_contiguous()
    if shared_gpu is not None:
        mask = 0xFFFFFFFF if src_row_mask is None else src_row_mask
        _check(load().fgnn_gather_rows_shared(_ptr(out), _ptr(src), _ptr(src_index), _ptr(dst_index), C.c_size_t(n),
                                              _ptr(d_n), C.c_size_t(n), C.c_size_t(dim), C.c_int(_T2DT[out.dtype]),
                                              C.c_uint32(mask), C.c_int(int(shared_gpu)), _stream()),
               "fgnn_gather_rows_shared")
        return out
    if src_row_mask is not None:
        _check(load().fgnn_gather_rows_masked(_ptr(out), _ptr(src), _ptr(src_index), _ptr(dst_index), C.c_size_t(n),
                                              _ptr(d_n), C.c_size_t(n), C.c_size_t(dim), C.c_int(_T2DT[out.dtype]),
                                              C.c_uint32(src_row_mask), _stream()), "fgnn_gather_rows_masked")
        return out
    _check(load().fgnn_gather_rows(_ptr(out), _ptr(src), _ptr(src_index), _ptr(dst_index), C.c_size_t(n), _ptr(d_n),
                                   C.c_size_t(n), C.c_size_t(dim), C.c_int(_T2DT[out.dtype]), _stream()),
           "fgnn_gather_rows")
    return out


def random_read_rate(array, num_items=4_000_000, repeats=24):
    """independent random 4-byte reads per second the GPU sustains from `array` (an int32 device tensor far larger than
    the caches) right now: fgnn_debug_random_reads, `repeats` launches of num_items reads between two events"""
    _need_gpu(array)
    sink = torch.zeros(1, dtype=torch.int32, device=array.device)
    L = load()

    def launch(salt):
        _check(L.fgnn_debug_random_reads(_ptr(array), C.c_size_t(array.numel()), C.c_size_t(num_items), C.c_uint64(salt),
                                         _ptr(sink), _stream()), "fgnn_debug_random_reads")
    launch(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(repeats):
        launch(2 + r * num_items)
    e1.record()
    e1.synchronize()
    return repeats * num_items / (e0.elapsed_time(e1) * 1e-3)


class CopySegment(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("words", C.c_size_t)]


class ExtractJob(C.Structure):
    """fgnn_extract_job (include/fgnn_hip.h)"""
    _fields_ = [("out", C.c_void_p), ("miss_rows", C.c_void_p), ("cache_rows", C.c_void_p),
                ("miss_src", C.c_void_p), ("miss_dst", C.c_void_p), ("cache_src", C.c_void_p), ("cache_dst", C.c_void_p),
                ("num_miss", C.c_size_t), ("num_cache", C.c_size_t), ("d_counts", C.c_void_p), ("cap", C.c_size_t),
                ("dim", C.c_size_t), ("dtype", C.c_int), ("miss_row_mask", C.c_uint32),
                ("label_out", C.c_void_p), ("label_src", C.c_void_p), ("label_index", C.c_void_p),
                ("num_label", C.c_size_t), ("label_dtype", C.c_int),
                ("segs", C.POINTER(CopySegment)), ("num_segs", C.c_int), ("link_workgroups", C.c_int),
                ("stamps", C.c_void_p)]


def _raw(t):
    """device-visible address of a tensor, a DevicePointer or None (C.c_void_p field value)"""
    p = _ptr(t)
    return p.value if isinstance(p, C.c_void_p) else p


def extract_fused(out, miss_rows, cache_rows, miss_src, miss_dst, cache_src, cache_dst, num_miss=None, num_cache=None,
                  d_counts=None, miss_row_mask=0xFFFFFFFF, label_out=None, label_src=None, label_index=None,
                  copies=(), link_workgroups=0, stamps=None):
    """fgnn_extract_fused: out[miss_dst] = miss_rows[miss_src & mask], out[cache_dst] = cache_rows[cache_src], the label
    rows and the word copies `copies` = [(dst, src, words)] in ONE launch.  Counts: host values (default: the index
    lists' lengths) or d_counts = device {num_miss, num_cache}.  Returns (workgroups, link-band workgroups)."""
    _need_gpu(out)
    dim = 1
    for s in out.shape[1:]:
        dim *= s
    j = ExtractJob()
    j.out, j.miss_rows, j.cache_rows = _raw(out), _raw(miss_rows), _raw(cache_rows)
    j.miss_src, j.miss_dst, j.cache_src, j.cache_dst = _raw(miss_src), _raw(miss_dst), _raw(cache_src), _raw(cache_dst)
    j.num_miss = (miss_src.numel() if miss_src is not None else 0) if num_miss is None else num_miss
    j.num_cache = (cache_src.numel() if cache_src is not None else 0) if num_cache is None else num_cache
    if d_counts is not None:
        j.d_counts = _raw(d_counts)
        j.cap = max(miss_src.numel() if miss_src is not None else 0, cache_src.numel() if cache_src is not None else 0)
    j.dim, j.dtype, j.miss_row_mask = dim, _T2DT[out.dtype], miss_row_mask
    if label_out is not None:
        j.label_out, j.label_src, j.label_index = _raw(label_out), _raw(label_src), _raw(label_index)
        j.num_label, j.label_dtype = label_index.numel(), _T2DT[label_out.dtype]
    segs = (CopySegment * max(len(copies), 1))()
    for k, (d, s, words) in enumerate(copies):
        segs[k].dst, segs[k].src, segs[k].words = _raw(d), _raw(s), words
    j.segs, j.num_segs = segs, len(copies)
    j.link_workgroups = link_workgroups
    j.stamps = _raw(stamps)
    L = load()
    L.fgnn_extract_fused_grid.restype = C.c_size_t
    L.fgnn_extract_fused_link_grid.restype = C.c_size_t
    _check(L.fgnn_extract_fused(C.byref(j), _stream()), "fgnn_extract_fused")
    return int(L.fgnn_extract_fused_grid(C.byref(j))), int(L.fgnn_extract_fused_link_grid(C.byref(j)))


# ---------------------------------------------------------------------------------------------------
# batch driver (fgnn_sampler / fgnn_batch)

MAX_LAYERS = 8
KHOP0, KHOP1, WEIGHTED_KHOP, RANDOM_WALK, WEIGHTED_KHOP_PREFIX, KHOP2, WEIGHTED_KHOP_HASH_DEDUP = 0, 1, 2, 3, 4, 5, 6

EXPORTS += [
    "fgnn_sampler_create", "fgnn_sampler_destroy", "fgnn_sampler_max_nodes", "fgnn_sampler_max_edges",
    "fgnn_batch_create", "fgnn_batch_destroy", "fgnn_sampler_sample", "fgnn_sampler_sample_ordered",
    "fgnn_sampler_run_batch", "fgnn_batch_enable_timing", "fgnn_batch_gather_ms", "fgnn_batch_cache_index", "fgnn_batch_extract",
    "fgnn_batch_extract_cached", "fgnn_sampler_run_batch_cached", "fgnn_batch_extract_cached_ms", "fgnn_batch_extract_launch_ms", "fgnn_batch_gather_kernel_ms", "fgnn_batch_finish", "fgnn_batch_wait", "fgnn_batch_row", "fgnn_batch_col",
    "fgnn_batch_data", "fgnn_batch_input_nodes", "fgnn_batch_output_nodes", "fgnn_batch_feat", "fgnn_batch_label",
    "fgnn_batch_cache_index_ptr", "fgnn_batch_device_meta", "fgnn_batch_host_meta", "fgnn_batch_meta_copied",
    "fgnn_sampler_run_range", "fgnn_sampler_sample_indexed", "fgnn_sampler_sample_begin", "fgnn_sampler_sample_begin_ordered", "fgnn_sampler_sample_end", "fgnn_sampler_prefix_tree_stats",
]


class SamplerConfig(C.Structure):
    _fields_ = [("indptr", C.c_void_p), ("indices", C.c_void_p), ("prob_prefix", C.c_void_p),
                ("num_node", C.c_size_t), ("sample_type", C.c_int), ("num_layers", C.c_size_t),
                ("fanout", C.c_size_t * MAX_LAYERS), ("max_batch_size", C.c_size_t), ("seed", C.c_uint64),
                ("walk_len", C.c_size_t), ("num_walks", C.c_size_t), ("restart_prob", C.c_double),
                ("prob_table", C.c_void_p), ("alias_table", C.c_void_p)]


class RunPlan(C.Structure):
    _fields_ = [("d_train", C.c_void_p), ("num_train", C.c_size_t), ("batch_size", C.c_size_t),
                ("batches", C.c_void_p), ("num_batches", C.c_size_t), ("streams", C.c_void_p),
                ("num_streams", C.c_size_t), ("cache_table", C.c_void_p), ("feat", C.c_void_p), ("label", C.c_void_p),
                ("cached", C.c_int), ("cache_rows", C.c_void_p), ("full_feat", C.c_void_p)]


class BatchMeta(C.Structure):
    _fields_ = [("key", C.c_uint64), ("num_edge", C.c_uint64 * MAX_LAYERS), ("num_src", C.c_uint32 * MAX_LAYERS),
                ("num_dst", C.c_uint32 * MAX_LAYERS), ("num_layers", C.c_uint32), ("num_input", C.c_uint32),
                ("num_output", C.c_uint32), ("num_miss", C.c_uint32), ("num_cache", C.c_uint32),
                ("overflow", C.c_uint32), ("t_start", C.c_uint64), ("t_sampled", C.c_uint64), ("t_closed", C.c_uint64)]


_TORCH_OF = {F32: torch.float32, F64: torch.float64, F16: torch.float16, U8: torch.uint8, I32: torch.int32,
             I8: torch.int8, I64: torch.int64}
_TYPESTR = {F32: "<f4", F64: "<f8", F16: "<f2", U8: "|u1", I32: "<i4", I8: "|i1", I64: "<i8"}


def _wrap_device(ptr, shape, dtype_code, device):
    n = 1
    for s in shape:
        n *= s
    if n == 0:
        return torch.empty(shape, dtype=_TORCH_OF[dtype_code], device=device)

    class _A:
        __cuda_array_interface__ = {"shape": tuple(shape), "typestr": _TYPESTR[dtype_code], "data": (ptr, False),
                                    "version": 2}
    return torch.as_tensor(_A(), device=device)


class Sampler:
    """DoGPUSample / DoGetCacheMissIndex / DoGPUFeatureExtract on one GPU, no host round trips."""

    def __init__(self, indptr, indices, fanout, max_batch_size, sample_type=KHOP2, seed=0x5A4D47, prob_prefix=None,
                 walk_len=0, num_walks=0, restart_prob=0.0, prob_table=None, alias_table=None):
        L = load()
        _need_gpu(indptr, indices)
        L.fgnn_sampler_create.restype = C.c_void_p
        L.fgnn_sampler_max_nodes.restype = C.c_size_t
        L.fgnn_sampler_max_edges.restype = C.c_size_t
        L.fgnn_batch_create.restype = C.c_void_p
        for name in ("fgnn_batch_row", "fgnn_batch_col", "fgnn_batch_data", "fgnn_batch_input_nodes",
                     "fgnn_batch_output_nodes", "fgnn_batch_feat", "fgnn_batch_label", "fgnn_batch_cache_index_ptr",
                     "fgnn_batch_device_meta"):
            getattr(L, name).restype = C.c_void_p
        self.device = indptr.device
        torch.cuda.set_device(self.device)
        self._keep = (indptr, indices, prob_prefix, prob_table, alias_table)
        cfg = SamplerConfig()
        cfg.indptr, cfg.indices = indptr.data_ptr(), indices.data_ptr()
        cfg.prob_prefix = prob_prefix.data_ptr() if prob_prefix is not None else 0
        cfg.num_node = indptr.numel() - 1
        cfg.sample_type = sample_type
        cfg.num_layers = len(fanout)
        for i, f in enumerate(fanout):
            cfg.fanout[i] = f
        cfg.max_batch_size, cfg.seed = max_batch_size, seed
        cfg.walk_len, cfg.num_walks, cfg.restart_prob = walk_len, num_walks, restart_prob
        cfg.prob_table = prob_table.data_ptr() if prob_table is not None else 0
        cfg.alias_table = alias_table.data_ptr() if alias_table is not None else 0
        if prob_prefix is not None:
            torch.cuda.synchronize(self.device)  # the sampler reads the table while it is created (search trees)
        err = C.c_int(0)
        self.h = C.c_void_p(L.fgnn_sampler_create(C.byref(cfg), C.byref(err)))
        if not self.h:
            raise FgnnError(f"fgnn_sampler_create failed with code {err.value} {L.fgnn_last_error().decode()}")
        self.fanout = list(fanout)
        self.max_batch_size = max_batch_size
        self.max_nodes = L.fgnn_sampler_max_nodes(self.h)

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.fgnn_sampler_destroy(self.h)
            self.h = None

    def prefix_tree_stats(self):
        """(rows with a search tree, long rows refused as not non-decreasing, bytes) -- weighted_khop_prefix only"""
        out = (C.c_size_t * 3)()
        _check(load().fgnn_sampler_prefix_tree_stats(self.h, out), "fgnn_sampler_prefix_tree_stats")
        return int(out[0]), int(out[1]), int(out[2])

    def max_edges(self, layer):
        return load().fgnn_sampler_max_edges(self.h, C.c_int(layer))

    def new_batch(self, feat_dim=0, feat_dtype=F32, label_dtype=I64, feat_rows_cap=0):
        return Batch(self, feat_dim, feat_dtype, label_dtype, feat_rows_cap)

    def sample(self, seeds, batch_key, batch, seq=None):
        """seq=None: internal call counter (single-threaded use); else the explicit batch order (thread-safe)."""
        _need_gpu(seeds)
        if seq is None:
            _check(load().fgnn_sampler_sample(self.h, _ptr(seeds), C.c_size_t(seeds.numel()), C.c_uint64(batch_key),
                                              batch.h, _stream()), "fgnn_sampler_sample")
        else:
            _check(load().fgnn_sampler_sample_ordered(self.h, C.c_uint64(seq), _ptr(seeds), C.c_size_t(seeds.numel()),
                                                      C.c_uint64(batch_key), batch.h, _stream()),
                   "fgnn_sampler_sample_ordered")

    def sample_indexed(self, seeds, batch_key, batch, cache_table=None):
        """fgnn_sampler_sample_indexed: sample + cache index in one call (an arch5 sampler's batch), internal counter"""
        _need_gpu(seeds)
        _check(load().fgnn_sampler_sample_indexed(self.h, _ptr(seeds), C.c_size_t(seeds.numel()), C.c_uint64(batch_key),
                                                  batch.h, _ptr(cache_table), _stream()), "fgnn_sampler_sample_indexed")

    def sample_begin(self, seq, seeds, batch_key, batch, stream=None):
        """fgnn_sampler_sample_begin_ordered: the batch's sampling CHAIN (everything up to its last sampler launch).
        Enqueue begin(k + 1) before end(k) to keep khop2's cross-batch chain busy; `stream` of end = that of begin."""
        _need_gpu(seeds)
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        _check(load().fgnn_sampler_sample_begin_ordered(self.h, C.c_uint64(seq), _ptr(seeds), C.c_size_t(seeds.numel()),
                                                        C.c_uint64(batch_key), batch.h, st),
               "fgnn_sampler_sample_begin_ordered")

    def sample_end(self, seq, batch, cache_table=None, stream=None):
        """fgnn_sampler_sample_end: the batch's TAIL (last dedup fill, fix-ups, table reset) [+ the cache-index split];
        extraction / finish are the caller's, on the same stream"""
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        _check(load().fgnn_sampler_sample_end(self.h, C.c_uint64(seq), batch.h, _ptr(cache_table), st),
               "fgnn_sampler_sample_end")

    def run_batch(self, seq, seeds, batch_key, batch, cache_table=None, feat=None, label=None, stream=None):
        """sample + cache index + extract + finish in ONE C call on `stream` (a torch stream; default current).
        Safe to call from several Python threads (ctypes releases the GIL)."""
        _need_gpu(seeds)
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        _check(load().fgnn_sampler_run_batch(self.h, C.c_uint64(seq), _ptr(seeds), C.c_size_t(seeds.numel()),
                                             C.c_uint64(batch_key), batch.h, _ptr(cache_table), _ptr(feat),
                                             _ptr(label), st), "fgnn_sampler_run_batch")


    def run_range(self, first_seq, count, train, batch_size, batches, streams, cache_table=None, feat=None, label=None,
                  cache_rows=None, full_feat=None, cached=False):
        """fgnn_sampler_run_range: batches first_seq .. first_seq+count-1 enqueued and collected by ONE native call (the
        reference's C++ loop thread, cuda_loops_arch1.cc:38-84).  Returns (metas, times, host_enqueue_seconds); times[i]
        = (ms, ms) HIP-event times of batch i, -1 where not timed.  Raises on a flagged batch."""
        call = self.range_call(first_seq, count, train, batch_size, batches, streams, cache_table, feat, label,
                               cache_rows, full_feat, cached)
        call.run()
        return call.results()

    def range_call(self, first_seq, count, train, batch_size, batches, streams, cache_table=None, feat=None,
                   label=None, cache_rows=None, full_feat=None, cached=False):
        """run_range in three steps for callers that time the native call alone: the argument marshalling here,
        .run() = the one C call (returns when every batch of the range has been collected), .results() afterwards."""
        _need_gpu(train)
        return _RangeCall(self, first_seq, count, train, batch_size, batches, streams, cache_table, feat, label,
                          cache_rows, full_feat, cached)

    def run_batch_cached(self, seq, seeds, batch_key, batch, cache_table, cache_rows, full_feat, label=None,
                         stream=None):
        """sample + cache index + CombineMissData (rows read from `full_feat`: device or pinned host tensor) +
        CombineCacheData + finish in one C call."""
        _need_gpu(seeds, cache_table, cache_rows)
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        _check(load().fgnn_sampler_run_batch_cached(self.h, C.c_uint64(seq), _ptr(seeds), C.c_size_t(seeds.numel()),
                                                    C.c_uint64(batch_key), batch.h, _ptr(cache_table),
                                                    _ptr(cache_rows), _ptr(full_feat), _ptr(label), st),
               "fgnn_sampler_run_batch_cached")


class _RangeCall:
    def __init__(self, sampler, first_seq, count, train, batch_size, batches, streams, cache_table, feat, label,
                 cache_rows, full_feat, cached):
        self.L = load()
        self.sampler, self.first_seq, self.count, self.batches = sampler, first_seq, count, batches
        plan = self.plan = RunPlan()
        plan.d_train, plan.num_train, plan.batch_size = train.data_ptr(), train.numel(), batch_size
        self._keep = (train, cache_table, feat, label, cache_rows, full_feat, list(streams),
                      (C.c_void_p * len(batches))(*[b.h for b in batches]),
                      (C.c_void_p * len(streams))(*[st.cuda_stream for st in streams]))
        plan.batches, plan.num_batches = C.cast(self._keep[-2], C.c_void_p), len(batches)
        plan.streams, plan.num_streams = C.cast(self._keep[-1], C.c_void_p), len(streams)
        plan.cache_table = cache_table.data_ptr() if cache_table is not None else None
        plan.feat = feat.data_ptr() if feat is not None else None
        plan.label = label.data_ptr() if label is not None else None
        plan.cached = 1 if cached else 0
        plan.cache_rows = cache_rows.data_ptr() if cache_rows is not None else None
        plan.full_feat = full_feat.data_ptr() if full_feat is not None else None
        self.metas = (BatchMeta * max(count, 1))()
        self.times = (C.c_float * (2 * max(count, 1)))()
        self.busy = C.c_double(0.0)
        self.rc = None
        self._args = (sampler.h, C.byref(plan), C.c_uint64(first_seq), C.c_size_t(count), self.metas, self.times,
                      C.byref(self.busy))

    def run(self):
        self.rc = self.L.fgnn_sampler_run_range(*self._args)

    def results(self):
        _check(self.rc, "fgnn_sampler_run_range")
        count, batches = self.count, self.batches
        out = [self.metas[i] for i in range(count)]
        for m in out:
            if m.overflow:
                raise FgnnError(f"batch {m.key} is invalid (overflow flag {m.overflow})")
        for i in range(max(0, count - len(batches)), count):  # the buffers hold the last batches: views stay usable
            batches[(self.first_seq + i) % len(batches)].meta = out[i]
        return out, [(self.times[2 * i], self.times[2 * i + 1]) for i in range(count)], self.busy.value


class Batch:
    def __init__(self, sampler, feat_dim, feat_dtype, label_dtype, feat_rows_cap):
        L = load()
        err = C.c_int(0)
        self.sampler = sampler
        self.h = C.c_void_p(L.fgnn_batch_create(sampler.h, C.c_size_t(feat_dim), C.c_int(feat_dtype),
                                                C.c_int(label_dtype), C.c_size_t(feat_rows_cap), C.byref(err)))
        if not self.h:
            raise FgnnError(f"fgnn_batch_create failed with code {err.value} {L.fgnn_last_error().decode()}")
        self.feat_dim, self.feat_dtype, self.label_dtype = feat_dim, feat_dtype, label_dtype
        self.feat_rows_cap = min(feat_rows_cap, sampler.max_nodes) if feat_rows_cap else sampler.max_nodes
        self.meta = None
        self._views = {}

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.fgnn_batch_destroy(self.h)
            self.h = None

    def cache_index(self, table):
        _check(load().fgnn_batch_cache_index(self.h, _ptr(table), _stream()), "fgnn_batch_cache_index")

    def extract(self, feat=None, label=None):
        _check(load().fgnn_batch_extract(self.h, _ptr(feat), _ptr(label), _stream()), "fgnn_batch_extract")

    def extract_cached(self, cache_rows, full_feat, label=None):
        _check(load().fgnn_batch_extract_cached(self.h, _ptr(cache_rows), _ptr(full_feat), _ptr(label), _stream()),
               "fgnn_batch_extract_cached")

    def finish(self):
        _check(load().fgnn_batch_finish(self.h, _stream()), "fgnn_batch_finish")

    def enable_timing(self, on=True):
        _check(load().fgnn_batch_enable_timing(self.h, C.c_int(1 if on else 0)), "fgnn_batch_enable_timing")

    def gather_ms(self):
        load().fgnn_batch_gather_ms.restype = C.c_float
        return float(load().fgnn_batch_gather_ms(self.h))

    def extract_cached_ms(self):
        """(miss-row gather ms, cached-row gather ms) of the last extract_cached, -1 where not timed"""
        out = (C.c_float * 2)()
        _check(load().fgnn_batch_extract_cached_ms(self.h, out), "fgnn_batch_extract_cached_ms")
        return float(out[0]), float(out[1])

    def extract_launch_ms(self):
        """HIP-event time around the whole one-launch cached extraction (-1 when not timed)"""
        load().fgnn_batch_extract_launch_ms.restype = C.c_float
        return float(load().fgnn_batch_extract_launch_ms(self.h))

    def d_num_input(self):
        """device view (int32[1]) of the batch summary's num_input"""
        p = load().fgnn_batch_device_meta(self.h)
        return _wrap_device_u32(p + BatchMeta.num_input.offset, 1, self.sampler.device)

    def input_nodes_buffer(self):
        """the whole input_nodes buffer (capacity max_nodes); valid entries = num_input"""
        return _wrap_device_u32(load().fgnn_batch_input_nodes(self.h), self.sampler.max_nodes, self.sampler.device)

    def wait(self):
        m = BatchMeta()
        _check(load().fgnn_batch_wait(self.h, C.byref(m)), "fgnn_batch_wait")
        self.meta = m
        if m.overflow:
            raise FgnnError(f"batch {m.key} is invalid (overflow flag {m.overflow}): a capacity was exceeded or a "
                            "cross-workgroup wait timed out")
        return m

    # views of the device buffers, sized by the (waited-for) summary.  A buffer of the batch never moves: it is wrapped
    # ONCE at its full capacity (torch.as_tensor on a __cuda_array_interface__ object costs ~25 us, eight of them per
    # batch were a fifth of a training step's host time) and every call returns a slice of that tensor.
    def _whole(self, key, ptr_fn, shape, dtype_code):
        t = self._views.get(key)
        if t is None:
            t = self._views[key] = _wrap_device(ptr_fn(), shape, dtype_code, self.sampler.device)
        return t

    def graph(self, layer):
        m, L = self.meta, load()
        ne, cap = int(m.num_edge[layer]), self.sampler.max_edges(layer)
        row = self._whole(("row", layer), lambda: L.fgnn_batch_row(self.h, layer), (cap,), I32)[:ne]
        col = self._whole(("col", layer), lambda: L.fgnn_batch_col(self.h, layer), (cap,), I32)[:ne]
        return row, col, int(m.num_src[layer]), int(m.num_dst[layer])

    def graph_buffers(self, layer):
        """(row, col) of a layer at their full capacity (the buffers never move): for callers that need stable addresses
        and sizes, e.g. a captured graph (examples/graphed_step.py); valid entries = meta.num_edge[layer]"""
        L, cap = load(), self.sampler.max_edges(layer)
        return (self._whole(("row", layer), lambda: L.fgnn_batch_row(self.h, layer), (cap,), I32),
                self._whole(("col", layer), lambda: L.fgnn_batch_col(self.h, layer), (cap,), I32))

    def feat_buffer(self):
        """the feature buffer at its full capacity [feat_rows_cap, feat_dim]; valid rows = meta.num_input"""
        return self._whole("feat", lambda: load().fgnn_batch_feat(self.h), (self.feat_rows_cap, self.feat_dim),
                           self.feat_dtype)

    def graph_data(self, layer):
        L = load()
        if not L.fgnn_batch_data(self.h, layer):
            return None
        return self._whole(("data", layer), lambda: L.fgnn_batch_data(self.h, layer), (self.sampler.max_edges(layer),),
                           I32)[:int(self.meta.num_edge[layer])]

    def input_nodes(self):
        return self._whole("input", lambda: load().fgnn_batch_input_nodes(self.h), (self.sampler.max_nodes,),
                           I32)[:int(self.meta.num_input)]

    def output_nodes(self):
        return self._whole("output", lambda: load().fgnn_batch_output_nodes(self.h), (self.sampler.max_batch_size,),
                           I32)[:int(self.meta.num_output)]

    def feat(self):
        # a batch larger than a caller-chosen feat_rows_cap carries meta.overflow and only the first cap rows
        return self._whole("feat", lambda: load().fgnn_batch_feat(self.h), (self.feat_rows_cap, self.feat_dim),
                           self.feat_dtype)[:min(int(self.meta.num_input), self.feat_rows_cap)]

    def label(self):
        return self._whole("label", lambda: load().fgnn_batch_label(self.h), (self.sampler.max_batch_size,),
                           self.label_dtype)[:int(self.meta.num_output)]

    def cache_index_arrays(self):
        m, L = self.meta, load()
        n = [int(m.num_miss), int(m.num_miss), int(m.num_cache), int(m.num_cache)]
        return [self._whole(("cidx", k), lambda k=k: L.fgnn_batch_cache_index_ptr(self.h, k), (self.sampler.max_nodes,),
                            I32)[:n[k]] for k in range(4)]
