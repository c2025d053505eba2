"""ctypes binding of oracle/libfgnn_oracle.so for tests/, smoke() and bench.py's cpu_baseline leg.

TEST INFRASTRUCTURE ONLY: nothing under fgnn-artifacts_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KHOP0, KHOP1, WEIGHTED_KHOP, RANDOM_WALK, WEIGHTED_KHOP_PREFIX, KHOP2, WEIGHTED_KHOP_HASH_DEDUP = range(7)
RNG_MT_CPU_TWIN, RNG_PHILOX = 0, 1
F32, F64, F16, U8, I32, I8, I64 = range(7)
EMPTY = 0xFFFFFFFF

_NP2DT = {np.dtype(np.float32): F32, np.dtype(np.float64): F64, np.dtype(np.float16): F16,
          np.dtype(np.uint8): U8, np.dtype(np.int32): I32, np.dtype(np.int8): I8, np.dtype(np.int64): I64}


class _MT(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("idx", C.c_int)]


class Rng(C.Structure):
    _fields_ = [("mode", C.c_int), ("seed", C.c_uint64), ("mt", _MT)]


class _Graph(C.Structure):
    _fields_ = [("row", C.POINTER(C.c_uint32)), ("col", C.POINTER(C.c_uint32)), ("data", C.POINTER(C.c_uint32)),
                ("num_src", C.c_size_t), ("num_dst", C.c_size_t), ("num_edge", C.c_size_t)]


class _Task(C.Structure):
    _fields_ = [("num_layers", C.c_size_t), ("graphs", C.POINTER(_Graph)), ("input_nodes", C.POINTER(C.c_uint32)),
                ("num_input_nodes", C.c_size_t), ("total_edges", C.c_size_t)]


class _Cfg(C.Structure):
    _fields_ = [("sample_type", C.c_int), ("num_layers", C.c_size_t), ("fanout", C.POINTER(C.c_size_t)),
                ("walk_len", C.c_size_t), ("num_walks", C.c_size_t), ("num_neighbor", C.c_size_t),
                ("restart_prob", C.c_double), ("alias_table", C.POINTER(C.c_uint32))]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libfgnn_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libfgnn_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.fgnn_predict_num_nodes.restype = C.c_size_t
        L.fgnn_table_size.restype = C.c_size_t
        L.fgnn_table_size.argtypes = [C.c_size_t, C.c_size_t]
        L.fgnn_oracle_ht_create.restype = C.c_void_p
        L.fgnn_oracle_ht_create.argtypes = [C.c_size_t, C.c_size_t]
        L.fgnn_oracle_do_sample.restype = C.POINTER(_Task)
        L.fgnn_mt19937_uniform_int.restype = C.c_uint32
        L.fgnn_mt19937_next.restype = C.c_uint32
        _LIB = L
    return _LIB


def _u32(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint32))


def make_rng(mode, seed=0):
    r = Rng()
    lib().fgnn_rng_init(C.byref(r), C.c_int(mode), C.c_uint64(seed))
    return r


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().fgnn_philox4x32_10(c, k, o)
    return list(o)


def philox_draw(seed, batch_key, tag, item, block):
    o = (C.c_uint32 * 4)()
    lib().fgnn_philox_draw(C.c_uint64(seed), C.c_uint64(batch_key), C.c_uint32(tag), C.c_uint32(item),
                           C.c_uint32(block), o)
    return list(o)


def predict_num_nodes(batch_size, fanout, n=None):
    f = (C.c_size_t * len(fanout))(*fanout)
    return lib().fgnn_predict_num_nodes(C.c_size_t(batch_size), f, C.c_size_t(len(fanout) if n is None else n))


def table_size(num, scale=2):
    return lib().fgnn_table_size(num, scale)


def _sample(fn, indptr, indices, inp, fanout, rng, batch_key, layer, extra=()):
    indptr, p_indptr = _u32(indptr)
    inp, p_in = _u32(inp)
    n = len(inp)
    src = np.empty(n * fanout + 1, dtype=np.uint32)
    dst = np.empty(n * fanout + 1, dtype=np.uint32)
    num_out = C.c_size_t(0)
    p_idx = indices.ctypes.data_as(C.POINTER(C.c_uint32))
    fn(p_indptr, p_idx, *extra, p_in, C.c_size_t(n), C.c_size_t(fanout), src.ctypes.data_as(C.POINTER(C.c_uint32)),
       dst.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(num_out), C.byref(rng), C.c_uint64(batch_key),
       C.c_uint32(layer))
    return src[:num_out.value].copy(), dst[:num_out.value].copy()


def sample_khop0(indptr, indices, inp, fanout, rng, batch_key=0, layer=0):
    indices = np.ascontiguousarray(indices, dtype=np.uint32)
    return _sample(lib().fgnn_oracle_sample_khop0, indptr, indices, inp, fanout, rng, batch_key, layer)


def sample_khop2(indptr, indices, inp, fanout, rng, batch_key=0, layer=0):
    """indices must be a C-contiguous uint32 array; it is MUTATED in place."""
    assert indices.dtype == np.uint32 and indices.flags.c_contiguous
    return _sample(lib().fgnn_oracle_sample_khop2, indptr, indices, inp, fanout, rng, batch_key, layer)


def sample_weighted_khop_prefix(indptr, indices, prefix, inp, fanout, rng, batch_key=0, layer=0):
    indices = np.ascontiguousarray(indices, dtype=np.uint32)
    prefix = np.ascontiguousarray(prefix, dtype=np.float32)
    return _sample(lib().fgnn_oracle_sample_weighted_khop_prefix, indptr, indices, inp, fanout, rng, batch_key,
                   layer, extra=(prefix.ctypes.data_as(C.POINTER(C.c_float)),))


def sample_khop1(indptr, indices, inp, fanout, rng, batch_key=0, layer=0):
    indices = np.ascontiguousarray(indices, dtype=np.uint32)
    return _sample(lib().fgnn_oracle_sample_khop1, indptr, indices, inp, fanout, rng, batch_key, layer)


def sample_weighted_khop(indptr, indices, prob, alias, inp, fanout, rng, batch_key=0, layer=0):
    indices = np.ascontiguousarray(indices, dtype=np.uint32)
    prob = np.ascontiguousarray(prob, dtype=np.float32)
    alias = np.ascontiguousarray(alias, dtype=np.uint32)
    return _sample(lib().fgnn_oracle_sample_weighted_khop, indptr, indices, inp, fanout, rng, batch_key, layer,
                   extra=(prob.ctypes.data_as(C.POINTER(C.c_float)), alias.ctypes.data_as(C.POINTER(C.c_uint32))))


def sample_weighted_khop_hash_dedup(indptr, indices, prob, alias, inp, fanout, rng, batch_key=0, layer=0):
    indices = np.ascontiguousarray(indices, dtype=np.uint32)
    prob = np.ascontiguousarray(prob, dtype=np.float32)
    alias = np.ascontiguousarray(alias, dtype=np.uint32)
    return _sample(lib().fgnn_oracle_sample_weighted_khop_hash_dedup, indptr, indices, inp, fanout, rng, batch_key,
                   layer, extra=(prob.ctypes.data_as(C.POINTER(C.c_float)), alias.ctypes.data_as(C.POINTER(C.c_uint32))))


def sample_random_walk(indptr, indices, inp, walk_len, restart_prob, num_walks, K, rng, batch_key=0, layer=0):
    indptr, p_indptr = _u32(indptr)
    indices, p_idx = _u32(indices)
    inp, p_in = _u32(inp)
    n = len(inp)
    src = np.empty(n * K + 1, dtype=np.uint32)
    dst = np.empty(n * K + 1, dtype=np.uint32)
    dat = np.empty(n * K + 1, dtype=np.uint32)
    num_out = C.c_size_t(0)
    P = C.POINTER(C.c_uint32)
    lib().fgnn_oracle_sample_random_walk(p_indptr, p_idx, p_in, C.c_size_t(n), C.c_size_t(walk_len),
                                         C.c_double(restart_prob), C.c_size_t(num_walks), C.c_size_t(K),
                                         src.ctypes.data_as(P), dst.ctypes.data_as(P), dat.ctypes.data_as(P),
                                         C.byref(num_out), C.byref(rng), C.c_uint64(batch_key), C.c_uint32(layer))
    k = num_out.value
    return src[:k].copy(), dst[:k].copy(), dat[:k].copy()


class HashTable:
    def __init__(self, num_node, capacity):
        self.h = C.c_void_p(lib().fgnn_oracle_ht_create(num_node, capacity))
        self.capacity = capacity

    def __del__(self):
        if getattr(self, "h", None):
            lib().fgnn_oracle_ht_destroy(self.h)
            self.h = None

    def reset(self):
        lib().fgnn_oracle_ht_reset(self.h)

    def fill_unique(self, items):
        items, p = _u32(items)
        return lib().fgnn_oracle_ht_fill_unique(self.h, p, C.c_size_t(len(items)))

    def fill_duplicates(self, items):
        items, p = _u32(items)
        uniq = np.empty(self.capacity + 1, dtype=np.uint32)
        nu = C.c_size_t(0)
        lib().fgnn_oracle_ht_fill_duplicates(self.h, p, C.c_size_t(len(items)),
                                             uniq.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(nu))
        return uniq[:nu.value].copy()

    def map_edges(self, src, dst):
        src, ps = _u32(src)
        dst, pd = _u32(dst)
        ns = np.empty(len(src), dtype=np.uint32)
        nd = np.empty(len(src), dtype=np.uint32)
        P = C.POINTER(C.c_uint32)
        lib().fgnn_oracle_map_edges(self.h, ps, pd, C.c_size_t(len(src)), ns.ctypes.data_as(P), nd.ctypes.data_as(P))
        return ns, nd


def do_sample(indptr, indices, seeds, fanout, sample_type, rng, batch_key, ht, prob_prefix=None, walk_len=0,
              num_walks=0, num_neighbor=0, restart_prob=0.0, alias_table=None):
    """DoGPUSample restatement.  `indices` (uint32, contiguous) is mutated when sample_type == KHOP2.
    Returns dict(graphs=[dict(row,col,data,num_src,num_dst,num_edge)], input_nodes, total_edges)."""
    assert indices.dtype == np.uint32 and indices.flags.c_contiguous
    indptr, p_indptr = _u32(indptr)
    seeds, p_seeds = _u32(seeds)
    fo = (C.c_size_t * len(fanout))(*fanout)
    al = None
    if alias_table is not None:
        alias_table = np.ascontiguousarray(alias_table, dtype=np.uint32)
        al = alias_table.ctypes.data_as(C.POINTER(C.c_uint32))
    cfg = _Cfg(sample_type, len(fanout), fo, walk_len, num_walks, num_neighbor, restart_prob, al)
    pp = None
    if prob_prefix is not None:
        prob_prefix = np.ascontiguousarray(prob_prefix, dtype=np.float32)
        pp = prob_prefix.ctypes.data_as(C.POINTER(C.c_float))
    t = lib().fgnn_oracle_do_sample(p_indptr, indices.ctypes.data_as(C.POINTER(C.c_uint32)), pp, C.byref(cfg), ht.h,
                                    p_seeds, C.c_size_t(len(seeds)), C.byref(rng), C.c_uint64(batch_key))
    tc = t.contents
    graphs = []
    for i in range(tc.num_layers):
        g = tc.graphs[i]
        ne = g.num_edge
        graphs.append(dict(row=np.ctypeslib.as_array(g.row, (ne,)).copy() if ne else np.empty(0, np.uint32),
                           col=np.ctypeslib.as_array(g.col, (ne,)).copy() if ne else np.empty(0, np.uint32),
                           data=(np.ctypeslib.as_array(g.data, (ne,)).copy() if ne else np.empty(0, np.uint32))
                           if g.data else None,
                           num_src=g.num_src, num_dst=g.num_dst, num_edge=ne))
    n_in = tc.num_input_nodes
    out = dict(graphs=graphs,
               input_nodes=np.ctypeslib.as_array(tc.input_nodes, (n_in,)).copy() if n_in else np.empty(0, np.uint32),
               total_edges=tc.total_edges)
    lib().fgnn_oracle_task_free(t)
    return out


def do_sample_dycache(indptr, indices, seeds, fanout, sample_type, rng, batch_key, ht, prob=None, alias=None):
    """DoGPUSampleDyCache restatement (cuda_loops.cc:269-498), the sampler of the arch4 dynamic-cache prototype: after
    layer 1's dedup the table takes ALL neighbours of every node seen so far (:400-421) and that list is the batch's
    input_nodes; layer 0 is sampled without inserting (:395-399); edges are mapped at the end (:461-476).
    `ht` must hold num_node items.  Same result layout as do_sample."""
    L = len(fanout)
    assert L >= 2 and sample_type in (KHOP0, KHOP1, WEIGHTED_KHOP)  # :333-357: everything else is CHECK(0)
    seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
    ht.reset()
    ht.fill_unique(seeds)  # :286-291
    unique, all_nodes, raw = seeds, None, [None] * L
    for i in range(L - 1, -1, -1):
        if sample_type == KHOP0:
            src, dst = sample_khop0(indptr, indices, unique, fanout[i], rng, batch_key, i)
        elif sample_type == KHOP1:
            src, dst = sample_khop1(indptr, indices, unique, fanout[i], rng, batch_key, i)
        else:
            src, dst = sample_weighted_khop(indptr, indices, prob, alias, unique, fanout[i], rng, batch_key, i)
        num_input = len(unique)
        if i == 0:
            num_unique = len(all_nodes)  # RefUnique of the table as it stands, :399
        else:
            unique = ht.fill_duplicates(dst)  # FillWithDupRevised + RefUnique, :402-404 / :423-425
            num_unique = len(unique)
            if i == 1:
                all_nodes = ht.fill_duplicates(extract_neighbour(indptr, indices, unique))  # :408-416
        raw[i] = (src, dst, num_unique, num_input)
    graphs, total = [], 0
    for src, dst, num_src, num_dst in raw:
        col, row = ht.map_edges(src, dst)  # GPUMapEdges over every layer, :461-476
        graphs.append(dict(row=row, col=col, data=None, num_src=num_src, num_dst=num_dst, num_edge=len(src)))
        total += len(src)
    return dict(graphs=graphs, input_nodes=all_nodes, total_edges=total)


def cache_table_build(ranking_nodes, num_cached, num_node):
    r, p = _u32(ranking_nodes)
    table = np.empty(num_node, dtype=np.uint32)
    lib().fgnn_oracle_cache_table_build(p, C.c_size_t(num_cached), C.c_size_t(num_node),
                                        table.ctypes.data_as(C.POINTER(C.c_uint32)))
    return table


def get_miss_cache_index(table, nodes):
    table, pt = _u32(table)
    nodes, pn = _u32(nodes)
    n = len(nodes)
    outs = [np.empty(n + 1, dtype=np.uint32) for _ in range(4)]
    nm, nc = C.c_size_t(0), C.c_size_t(0)
    P = C.POINTER(C.c_uint32)
    lib().fgnn_oracle_get_miss_cache_index(pt, pn, C.c_size_t(n), outs[0].ctypes.data_as(P), outs[1].ctypes.data_as(P),
                                           C.byref(nm), outs[2].ctypes.data_as(P), outs[3].ctypes.data_as(P),
                                           C.byref(nc))
    return (outs[0][:nm.value].copy(), outs[1][:nm.value].copy(), outs[2][:nc.value].copy(),
            outs[3][:nc.value].copy())


def extract(src, index):
    src = np.ascontiguousarray(src)
    index, pi = _u32(index)
    dim = 1 if src.ndim == 1 else src.shape[1]
    out = np.empty((len(index),) + src.shape[1:], dtype=src.dtype)
    lib().fgnn_oracle_extract(out.ctypes.data_as(C.c_void_p), src.ctypes.data_as(C.c_void_p), pi,
                              C.c_size_t(len(index)), C.c_size_t(dim), C.c_int(_NP2DT[src.dtype]))
    return out


def mock_extract(src, index, empty_feat_bits):
    """CPUMockExtract: rows of a 2^empty_feat_bits-row table, ids masked"""
    src = np.ascontiguousarray(src)
    index, pi = _u32(index)
    dim = 1 if src.ndim == 1 else src.shape[1]
    out = np.empty((len(index),) + src.shape[1:], dtype=src.dtype)
    lib().fgnn_oracle_mock_extract(out.ctypes.data_as(C.c_void_p), src.ctypes.data_as(C.c_void_p), pi,
                                   C.c_size_t(len(index)), C.c_size_t(dim), C.c_int(_NP2DT[src.dtype]),
                                   C.c_uint(empty_feat_bits))
    return out


def combine(out, rows, src_index, dst_index):
    assert out.flags.c_contiguous
    rows = np.ascontiguousarray(rows)
    dst_index, pd = _u32(dst_index)
    ps = None
    if src_index is not None:
        src_index, ps = _u32(src_index)
    dim = 1 if out.ndim == 1 else out.shape[1]
    lib().fgnn_oracle_combine(out.ctypes.data_as(C.c_void_p), rows.ctypes.data_as(C.c_void_p), ps, pd,
                              C.c_size_t(len(dst_index)), C.c_size_t(dim), C.c_int(_NP2DT[out.dtype]))
    return out


def presample_rank(freq):
    freq, pf = _u32(freq)
    out = np.empty(len(freq), dtype=np.uint32)
    lib().fgnn_oracle_presample_rank(pf, C.c_size_t(len(freq)), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def extract_neighbour(indptr, indices, inp):
    indptr, pp = _u32(indptr)
    indices, pi = _u32(indices)
    inp, pn = _u32(inp)
    fn = lib().fgnn_oracle_extract_neighbour
    fn.restype = C.c_size_t
    n = fn(pp, pi, pn, C.c_size_t(len(inp)), None)
    out = np.empty(n, dtype=np.uint32)
    fn(pp, pi, pn, C.c_size_t(len(inp)), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def sample_all_neighbour(indptr, indices, seeds, num_layers):
    """input_nodes of DoGPUSampleAllNeighbour (first-occurrence order)"""
    indptr, pp = _u32(indptr)
    indices, pi = _u32(indices)
    seeds, ps = _u32(seeds)
    num_node = len(indptr) - 1
    out = np.empty(num_node, dtype=np.uint32)
    fn = lib().fgnn_oracle_sample_all_neighbour
    fn.restype = C.c_size_t
    n = fn(pp, pi, ps, C.c_size_t(len(seeds)), C.c_size_t(num_layers), C.c_size_t(num_node),
           out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out[:n].copy()


def shuffle_minstd0(data, seed):
    data = np.ascontiguousarray(data, dtype=np.uint32).copy()
    lib().fgnn_oracle_shuffle_minstd0(data.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_size_t(len(data)),
                                      C.c_uint64(seed))
    return data


def dist_shuffler_partition(num_data, batch_size, sampler_id, num_sampler):
    v = [C.c_size_t(0) for _ in range(5)]
    lib().fgnn_oracle_dist_shuffler_partition(C.c_size_t(num_data), C.c_size_t(batch_size), C.c_int(sampler_id),
                                              C.c_int(num_sampler), *[C.byref(x) for x in v])
    return dict(dataset_offset=v[0].value, num_local_step=v[1].value, local_data_size=v[2].value,
                last_batch_size=v[3].value, epoch_step=v[4].value)


def aligned_shuffler_partition(num_data, batch_size, worker_id, num_worker):
    """DistAlignedShuffler's split (dist/dist_shuffler_aligned.cc:45-71)."""
    v = [C.c_size_t(0) for _ in range(7)]
    lib().fgnn_oracle_aligned_shuffler_partition(C.c_size_t(num_data), C.c_size_t(batch_size), C.c_size_t(worker_id),
                                                 C.c_size_t(num_worker), *[C.byref(x) for x in v])
    return dict(padded_size=v[0].value, local_data_size=v[1].value, num_local_step=v[2].value,
                epoch_step=v[3].value, step_offset=v[4].value, dataset_offset=v[5].value,
                last_batch_size=v[6].value)


def aligned_shuffler_batches(train_set, batch_size, worker_id, num_worker, num_epoch):
    """The batches one worker's DistAlignedShuffler hands out, in order: (epoch, global step, ids)
    (dist_shuffler_aligned.cc:80-146: cumulative Fisher-Yates over the padded set, seed = epoch)."""
    train_set = np.ascontiguousarray(train_set, dtype=np.uint32)
    p = aligned_shuffler_partition(len(train_set), batch_size, worker_id, num_worker)
    data = np.concatenate([train_set, train_set[:p["padded_size"] - len(train_set)]])
    for epoch in range(num_epoch):
        data = shuffle_minstd0(data, epoch)
        mine = data[p["dataset_offset"]:p["dataset_offset"] + p["local_data_size"]]
        for ls in range(p["num_local_step"]):
            yield epoch, p["step_offset"] + ls, mine[ls * batch_size:(ls + 1) * batch_size]


class OmpBaseline:
    """The reference's OpenMP CPU path (see fgnn_oracle.c, 'OpenMP CPU baseline'); timing only."""

    def __init__(self, num_node, capacity, threads):
        L = lib()
        L.fgnn_omp_create.restype = C.c_void_p
        L.fgnn_omp_create.argtypes = [C.c_size_t, C.c_size_t, C.c_int]
        L.fgnn_omp_sample_batch.restype = C.c_size_t
        self.h = C.c_void_p(L.fgnn_omp_create(num_node, capacity, threads))
        self.threads = threads

    def __del__(self):
        if getattr(self, "h", None):
            lib().fgnn_omp_destroy(self.h)
            self.h = None

    def sample_batch(self, indptr, indices, seeds, fanout, feat=None, feat_row_mask=0xFFFFFFFF, feat_out=None):
        fo = (C.c_size_t * len(fanout))(*fanout)
        n_in = C.c_size_t(0)
        P = C.POINTER(C.c_uint32)
        fp = feat.ctypes.data_as(C.c_void_p) if feat is not None else None
        fo_p = feat_out.ctypes.data_as(C.c_void_p) if feat_out is not None else None
        dim = feat.shape[1] if feat is not None else 0
        edges = lib().fgnn_omp_sample_batch(self.h, indptr.ctypes.data_as(P), indices.ctypes.data_as(P),
                                            seeds.ctypes.data_as(P), C.c_size_t(len(seeds)), fo,
                                            C.c_size_t(len(fanout)), fp, C.c_size_t(dim), C.c_uint32(feat_row_mask),
                                            fo_p, C.byref(n_in))
        return edges, n_in.value


class RefBaseline:
    """The REFERENCE's own CPU sampling path (oracle/_ref/libref_driver.so, compiled from /root/reference's unmodified
    cpu/*.cc; ref_driver.cc:ref_bench_*), timing only.  Present only where `make -C oracle _ref` has run (the build
    container; the built files travel to the GPU box with the snapshot)."""
    _DIR = os.path.join(_HERE, "_ref")

    @classmethod
    def available(cls):
        return all(os.path.exists(os.path.join(cls._DIR, f)) for f in ("libref_driver.so", "libref_loader.so"))

    def __init__(self, num_node, max_edges_per_layer, max_unique, threads):
        loader = C.CDLL(os.path.join(self._DIR, "libref_loader.so"))
        loader.ref_loader_open.restype = C.c_void_p
        loader.ref_loader_sym.restype = C.c_void_p
        loader.ref_loader_sym.argtypes = [C.c_void_p, C.c_char_p]
        drv = loader.ref_loader_open(os.path.join(self._DIR, "libref_driver.so").encode())
        if not drv:
            raise OSError("cannot open libref_driver.so")

        def fn(name, restype, *argtypes):
            addr = loader.ref_loader_sym(C.c_void_p(drv), name.encode())
            if not addr:
                raise OSError(name + " not found in libref_driver.so")
            return C.CFUNCTYPE(restype, *argtypes)(addr)

        self._keep = loader
        self._create = fn("ref_bench_create", C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int)
        self._destroy = fn("ref_bench_destroy", None, C.c_void_p)
        self._set_threads = fn("ref_bench_set_threads", None, C.c_int)
        P = C.POINTER(C.c_uint32)
        self._batch = fn("ref_bench_batch", C.c_int, C.c_void_p, P, P, P, C.c_size_t, C.POINTER(C.c_size_t), C.c_size_t,
                         C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t),
                         C.POINTER(C.c_size_t))
        self.h = self._create(num_node, max_edges_per_layer, max_unique, threads)
        if not self.h:
            raise MemoryError("ref_bench_create failed")
        self.threads = threads

    def close(self):
        if getattr(self, "h", None):
            self._destroy(self.h)
            self.h = None

    __del__ = close

    def set_threads(self, threads):
        self._set_threads(threads)
        self.threads = threads

    def sample_batch(self, indptr, indices, seeds, fanout, sample_type=KHOP2, feat=None, empty_feat_bits=0,
                     feat_out=None):
        """indices (uint32, contiguous) is mutated for KHOP2.  Returns (edges, input_nodes)."""
        fo = (C.c_size_t * len(fanout))(*fanout)
        P = C.POINTER(C.c_uint32)
        e, n = C.c_size_t(0), C.c_size_t(0)
        fp = feat.ctypes.data_as(C.c_void_p) if feat is not None else None
        fo_p = feat_out.ctypes.data_as(C.c_void_p) if feat_out is not None else None
        rc = self._batch(self.h, indptr.ctypes.data_as(P), indices.ctypes.data_as(P), seeds.ctypes.data_as(P), len(seeds),
                         fo, len(fanout), sample_type, fp, feat.shape[1] if feat is not None else 0, empty_feat_bits,
                         fo_p, C.byref(e), C.byref(n))
        if rc != 0:
            raise RuntimeError("ref_bench_batch: buffers too small")
        return e.value, n.value
