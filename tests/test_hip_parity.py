"""GPU parity: the HIP path (through the C ABI of libfgnn_hip.so) against the oracle (Philox mode) on the
same seeded inputs.  Bar: bit-exact -- all of this is integer / index / byte work."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SEED = 0x5A4D47


def dev(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(a).cuda()


def host_u32(t, n=None):
    a = t.cpu().numpy()
    if n is not None:
        a = a[:n]
    return a.view(np.uint32) if a.dtype == np.int32 else a


@pytest.fixture(scope="module")
def hip():
    from fgnn_hip import lib
    lib.load()
    assert torch.cuda.is_available()
    return lib


@pytest.fixture(scope="module")
def graph():
    from fgnn_hip import synth
    return synth.powerlaw_csr(3000, 60000, seed=21)


def _seeds(n, num_node, seed=1):
    return np.random.default_rng(seed).permutation(num_node)[:n].astype(np.uint32)


@pytest.mark.parametrize("kind", ["khop0", "khop2"])
@pytest.mark.parametrize("fanout", [1, 3, 10, 25, 45, 90])
def test_sampler_matches_oracle(hip, oracle, graph, kind, fanout):
    indptr, indices = graph
    d_indptr, d_indices = dev(indptr), dev(indices.copy())
    o_indices = indices.copy()
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    ofn = oracle.sample_khop0 if kind == "khop0" else oracle.sample_khop2
    # three calls in a row: khop2's CSR mutation must carry identically on both sides
    for call, (n, layer) in enumerate([(700, 1), (1300, 0), (5, 1)]):
        inp = _seeds(n, len(indptr) - 1, seed=call)
        for src_mode in (hip.SRC_GLOBAL, hip.SRC_LOCAL) if call == 0 else (hip.SRC_GLOBAL,):
            if src_mode == hip.SRC_LOCAL and kind == "khop2":
                continue  # a second khop2 call would mutate the CSR again
            out_src, out_dst, d_num = hip.sample_khop(kind, d_indptr, d_indices, dev(inp), fanout, SEED, 77 + call,
                                                      layer, src_mode)
            if src_mode == hip.SRC_GLOBAL:
                o_src, o_dst = ofn(indptr, o_indices, inp, fanout, rng, 77 + call, layer)
            ne = int(d_num.cpu()[0])
            assert ne == len(o_dst)
            np.testing.assert_array_equal(host_u32(out_dst, ne), o_dst)
            if src_mode == hip.SRC_GLOBAL:
                np.testing.assert_array_equal(host_u32(out_src, ne), o_src)
            else:
                pos = {int(v): i for i, v in enumerate(inp)}
                np.testing.assert_array_equal(host_u32(out_src, ne), [pos[int(v)] for v in o_src])
        np.testing.assert_array_equal(host_u32(d_indices), o_indices)
    if kind == "khop2" and fanout < 45:
        assert (o_indices != indices).any()


@pytest.mark.parametrize("kind", ["khop0", "khop2"])
@pytest.mark.parametrize("fanout", [5, 25])
def test_sampler_hub_rows(hip, oracle, kind, fanout):
    """Rows of tens of thousands of entries next to short ones (R-MAT hubs): khop0 draws once per row element
    (cuda_sampling_khop0.cu:41-90) -- rows beyond 4096 entries are walked by the whole workgroup, shorter ones by a wave."""
    rng_np = np.random.default_rng(11)
    num_node = 3000
    deg = rng_np.integers(0, 40, size=num_node)
    deg[[7, 8, 1500]] = [30000, 4097, 4096]  # two rows on the workgroup path, one exactly at the wave path's limit
    deg[2999] = 100000
    indptr = np.zeros(num_node + 1, dtype=np.uint32)
    indptr[1:] = np.cumsum(deg)
    indices = rng_np.integers(0, num_node, size=int(indptr[-1])).astype(np.uint32)
    d_indptr, d_indices = dev(indptr), dev(indices.copy())
    o_indices = indices.copy()
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    ofn = oracle.sample_khop0 if kind == "khop0" else oracle.sample_khop2
    for call in range(2):
        inp = np.concatenate([np.array([7, 2999, 8, 1500], dtype=np.uint32), _seeds(300, num_node - 1, seed=call)])
        inp = np.unique(inp)[rng_np.permutation(len(np.unique(inp)))].astype(np.uint32)
        out_src, out_dst, d_num = hip.sample_khop(kind, d_indptr, d_indices, dev(inp), fanout, SEED, 5 + call, 0,
                                                  hip.SRC_GLOBAL)
        o_src, o_dst = ofn(indptr, o_indices, inp, fanout, rng, 5 + call, 0)
        ne = int(d_num.cpu()[0])
        assert ne == len(o_dst)
        np.testing.assert_array_equal(host_u32(out_dst, ne), o_dst)
        np.testing.assert_array_equal(host_u32(out_src, ne), o_src)
        np.testing.assert_array_equal(host_u32(d_indices), o_indices)


@pytest.mark.parametrize("kind", ["khop0", "khop2"])
def test_sampler_edge_cases(hip, oracle, kind):
    # rows: 0,3,0,1,5,0,6 neighbours -- isolated nodes, rows shorter / equal / longer than the fanout
    indptr = np.array([0, 0, 3, 3, 4, 9, 9, 15], dtype=np.uint32)
    indices = np.array([4, 6, 1, 0, 1, 2, 3, 5, 6, 0, 1, 2, 3, 4, 5], dtype=np.uint32)
    ofn = oracle.sample_khop0 if kind == "khop0" else oracle.sample_khop2
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    d_indptr, d_indices = dev(indptr), dev(indices.copy())
    o_indices = indices.copy()
    for inp in (np.array([], dtype=np.uint32), np.arange(7, dtype=np.uint32), np.array([6, 5, 4], dtype=np.uint32),
                np.array([0, 2, 5], dtype=np.uint32)):
        for fanout in (1, 3, 5, 6, 8):
            o_src, o_dst = ofn(indptr, o_indices, inp, fanout, rng, 5, 0)
            d_inp = dev(inp) if len(inp) else torch.empty(0, dtype=torch.int32, device="cuda")
            out_src, out_dst, d_num = hip.sample_khop(kind, d_indptr, d_indices, d_inp, fanout, SEED, 5, 0)
            ne = int(d_num.cpu()[0])
            assert ne == len(o_dst)
            np.testing.assert_array_equal(host_u32(out_dst, ne), o_dst)
            np.testing.assert_array_equal(host_u32(out_src, ne), o_src)
            np.testing.assert_array_equal(host_u32(d_indices), o_indices)


def test_sampler_device_side_count(hip, oracle, graph):
    """d_num_input overrides the host count: grids are sized by the cap, surplus workgroups exit."""
    indptr, indices = graph
    inp = _seeds(1000, len(indptr) - 1, seed=9)
    d_n = torch.tensor([333], dtype=torch.int32, device="cuda")
    out_src, out_dst, d_num = hip.sample_khop("khop0", dev(indptr), dev(indices), dev(inp), 7, SEED, 1, 0,
                                              d_num_input=d_n)
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    o_src, o_dst = oracle.sample_khop0(indptr, indices, inp[:333], 7, rng, 1, 0)
    ne = int(d_num.cpu()[0])
    assert ne == len(o_dst)
    np.testing.assert_array_equal(host_u32(out_dst, ne), o_dst)


def test_hashtable_matches_oracle(hip, oracle):
    rs = np.random.default_rng(3)
    num_node = 50000
    seeds = rs.permutation(num_node)[:2000].astype(np.uint32)
    ht = hip.HashTable(40000)
    oht = oracle.HashTable(num_node, 40000)
    for rep in range(2):  # second round after a reset
        ht.reset()
        oht.reset()
        ht.fill_unique(dev(seeds))
        assert oht.fill_unique(seeds) == 0
        for n in (0, 1, 7000, 30000):
            # skewed ids: many duplicates inside a call and against earlier calls
            items = np.minimum((num_node * rs.random(n) ** 3).astype(np.uint32), num_node - 1)
            d_items = dev(items) if n else torch.empty(0, dtype=torch.int32, device="cuda")
            mapped = ht.fill_duplicates(d_items)
            uniq = oht.fill_duplicates(items)
            _, o_map = oht.map_edges(items, items)
            assert ht.num_items() == len(uniq)
            np.testing.assert_array_equal(host_u32(ht.unique()), uniq)
            np.testing.assert_array_equal(host_u32(mapped, n), o_map)
            if n:
                np.testing.assert_array_equal(host_u32(ht.map(d_items)), o_map)
        np.testing.assert_array_equal(host_u32(ht.unique())[:len(seeds)], seeds)


@pytest.mark.parametrize("max_fill", [40000, (1 << 27) - 1])
def test_hashtable_generations(hip, oracle, max_fill):
    """Reset as a generation bump: buckets of earlier generations must read as empty, also across the physical wipe
    when the generation counter wraps (max_fill 2^27 leaves 3 generation bits: the 7th reset wipes) and when the
    same keys come back generation after generation."""
    rs = np.random.default_rng(13)
    num_node = 20000
    ht = hip.HashTable(30000, max_fill_items=max_fill)
    oht = oracle.HashTable(num_node, 30000)
    for rep in range(24):
        ht.reset()
        oht.reset()
        seeds = rs.permutation(num_node)[:500 + 37 * rep].astype(np.uint32)
        ht.fill_unique(dev(seeds))
        assert oht.fill_unique(seeds) == 0
        for n in (9000, 25000):
            items = np.minimum((num_node * rs.random(n) ** 2).astype(np.uint32), num_node - 1)
            mapped = ht.fill_duplicates(dev(items))
            uniq = oht.fill_duplicates(items)
            _, o_map = oht.map_edges(items, items)
            assert ht.num_items() == len(uniq)
            np.testing.assert_array_equal(host_u32(ht.unique()), uniq)
            np.testing.assert_array_equal(host_u32(mapped, n), o_map)
            np.testing.assert_array_equal(host_u32(ht.map(dev(items))), o_map)
    with pytest.raises(hip.FgnnError):  # a fill larger than the table was created for
        hip.HashTable(1000, max_fill_items=5000).fill_duplicates(dev(np.zeros(6000, dtype=np.uint32)))


def test_hashtable_device_side_count(hip, oracle):
    rs = np.random.default_rng(4)
    items = rs.integers(0, 5000, size=10000).astype(np.uint32)
    ht = hip.HashTable(6000)
    d_n = torch.tensor([4321], dtype=torch.int64, device="cuda")
    mapped = ht.fill_duplicates(dev(items), d_num_items=d_n)
    oht = oracle.HashTable(5000, 6000)
    uniq = oht.fill_duplicates(items[:4321])
    np.testing.assert_array_equal(host_u32(ht.unique()), uniq)
    np.testing.assert_array_equal(host_u32(mapped, 4321), oht.map_edges(items[:4321], items[:4321])[1])


@pytest.mark.parametrize("n,frac", [(0, 0.5), (1, 1.0), (5000, 0.0), (5000, 1.0), (70001, 0.3)])
def test_cache_split_matches_oracle(hip, oracle, n, frac):
    rs = np.random.default_rng(8)
    num_node = 200000
    rank = rs.permutation(num_node).astype(np.uint32)
    table = oracle.cache_table_build(rank, int(num_node * frac), num_node)
    nodes = rs.permutation(num_node)[:n].astype(np.uint32)
    d_nodes = dev(nodes) if n else torch.empty(0, dtype=torch.int32, device="cuda")
    ms, md, cs, cd, d_counts = hip.get_miss_cache_index(dev(table), d_nodes)
    o = oracle.get_miss_cache_index(table, nodes)
    nm, nc = [int(x) for x in d_counts.cpu()]
    assert (nm, nc) == (len(o[0]), len(o[2])) and nm + nc == n
    for got, want, k in ((ms, o[0], nm), (md, o[1], nm), (cs, o[2], nc), (cd, o[3], nc)):
        np.testing.assert_array_equal(host_u32(got, k), want)


@pytest.mark.parametrize("dtype,dim", [(np.float32, 100), (np.float32, 128), (np.float32, 256), (np.float32, 7),
                                       (np.int64, 1), (np.float16, 24), (np.float16, 5), (np.uint8, 3),
                                       (np.float64, 2)])
def test_gather_matches_oracle(hip, oracle, dtype, dim):
    rs = np.random.default_rng(2)
    n_src, n = 5000, 3777
    src = (rs.standard_normal((n_src, dim)) * 1000).astype(dtype)
    idx = rs.integers(0, n_src, size=n).astype(np.uint32)
    want = oracle.extract(src, idx)
    got = torch.empty((n, dim), dtype=torch.from_numpy(src).dtype, device="cuda")
    hip.gather_rows(got, dev(src), src_index=dev(idx))
    assert got.cpu().numpy().tobytes() == want.tobytes()
    # CombineMissData + CombineCacheData reassemble the batch feature tensor
    table = oracle.cache_table_build(rs.permutation(n_src).astype(np.uint32), n_src // 3, n_src)
    nodes = rs.permutation(n_src)[:n].astype(np.uint32)
    ms, md, cs, cd = oracle.get_miss_cache_index(table, nodes)
    cache_rows = np.zeros((n_src // 3, dim), dtype=dtype)
    cache_rows[table[table != oracle.EMPTY]] = src[np.nonzero(table != oracle.EMPTY)[0]]
    out = torch.zeros((n, dim), dtype=got.dtype, device="cuda")
    miss_rows = dev(oracle.extract(src, ms))
    if len(ms):
        hip.gather_rows(out, miss_rows, src_index=None, dst_index=dev(md), n=len(ms))
    if len(cs):
        hip.gather_rows(out, dev(cache_rows), src_index=dev(cs), dst_index=dev(cd), n=len(cs))
    assert out.cpu().numpy().tobytes() == src[nodes].tobytes()


@pytest.mark.parametrize("shared", [0, 1])
def test_gather_from_pinned_host_rows_with_the_shared_gpu_hint(hip, oracle, shared):
    """fgnn_gather_rows_shared: miss rows read from pinned host memory by the kernel itself (ExtractMissData +
    CombineMissData, cuda_cache_manager_host.cc:38-56); the hint only changes the launch shape of a host-source gather"""
    rs = np.random.default_rng(5)
    n_src, n, dim = 1 << 12, 30011, 128
    src = rs.standard_normal((n_src, dim)).astype(np.float32)
    idx = rs.integers(0, 1 << 20, size=n).astype(np.uint32)  # ids beyond the table: masked like SAMGRAPH_EMPTY_FEAT
    dst = rs.permutation(n).astype(np.uint32)
    want = np.zeros((n, dim), dtype=np.float32)
    want[dst] = oracle.extract(src, idx & (n_src - 1))
    for table in (torch.from_numpy(src).pin_memory(), dev(src)):  # host source, then an HBM source (hint ignored)
        got = torch.zeros((n, dim), dtype=torch.float32, device="cuda")
        hip.gather_rows(got, table, src_index=dev(idx), dst_index=dev(dst), src_row_mask=n_src - 1, shared_gpu=shared)
        assert got.cpu().numpy().tobytes() == want.tobytes()


@pytest.mark.parametrize("dim,dtype", [(128, np.float32), (100, np.float32), (256, np.float32), (24, np.float16),
                                       (4, np.float32), (2, np.float64)])
@pytest.mark.parametrize("host_source,link", [(True, 64), (True, 3), (False, 0), (True, 0)])
@pytest.mark.parametrize("device_counts", [False, True])
def test_fused_extraction_matches_oracle(hip, oracle, dim, dtype, host_source, link, device_counts):
    """fgnn_extract_fused: ExtractMissData's fetch + CombineMissData + CombineCacheData
    (cuda_cache_manager_device.cu:165-210,339-442) + the label rows + the word copies out of a message slot, ONE launch
    -- against the oracle's split and extraction: miss rows from pinned host memory through a link band of `link`
    workgroups (0: no band), or from HBM; host counts or device counts; masked miss ids (SAMGRAPH_EMPTY_FEAT)."""
    rs = np.random.default_rng(11)
    n_node, n_tab, n = 1 << 15, 1 << 12, 23117
    src = (rs.standard_normal((n_tab, dim)) * 100).astype(dtype)          # the (masked) feature table
    rank = rs.permutation(n_node).astype(np.uint32)
    n_cached = n_node // 5
    table = oracle.cache_table_build(rank, n_cached, n_node)
    nodes = rs.permutation(n_node)[:n].astype(np.uint32)
    ms, md, cs, cd = oracle.get_miss_cache_index(table, nodes)
    cache_rows = oracle.extract(src, rank[:n_cached] & (n_tab - 1))         # the trainer's cache: feat[rank[i] & mask]
    want = oracle.extract(src, nodes & (n_tab - 1))
    label = rs.integers(0, 1000, size=n_node).astype(np.int64)
    seeds = nodes[:997]
    # arrays of a "message slot" at odd word offsets, copied out by the same launch
    slot = rs.integers(0, 1 << 32, size=60000, dtype=np.uint64).astype(np.uint32)
    d_slot = dev(slot)
    pieces = [(3, 997), (1001, 1), (1003, 33333), (40001, 0), (40001, 7777)]
    keeps = [torch.zeros(max(w, 1), dtype=torch.int32, device="cuda") for _, w in pieces]
    copies = [(k, hip.DevicePointer(d_slot.data_ptr() + 4 * off), w) for k, (off, w) in zip(keeps, pieces)]
    t_dtype = torch.from_numpy(src).dtype
    miss_rows = torch.from_numpy(src).pin_memory() if host_source else dev(src)
    out = torch.zeros((n, dim), dtype=t_dtype, device="cuda")
    lab = torch.zeros(len(seeds), dtype=torch.int64, device="cuda")
    pad = 100  # lists longer than the counts under device counts: the tail must not be touched
    d = [dev(np.concatenate([a, np.full(pad, 0x7FFFFFF0, np.uint32)])) if device_counts else dev(a) for a in (ms, md, cs, cd)]
    kw = dict(d_counts=dev(np.array([len(ms), len(cs)], np.uint32))) if device_counts else \
        dict(num_miss=len(ms), num_cache=len(cs))
    grid, band = hip.extract_fused(out, miss_rows, dev(cache_rows), d[0], d[1], d[2], d[3], miss_row_mask=n_tab - 1,
                                   label_out=lab, label_src=dev(label), label_index=dev(seeds), copies=copies,
                                   link_workgroups=link, **kw)
    torch.cuda.synchronize()
    assert grid > band and (band > 0) == (link > 0)
    assert out.cpu().numpy().tobytes() == want.tobytes()
    np.testing.assert_array_equal(lab.cpu().numpy(), label[seeds])
    for k, (off, w) in zip(keeps, pieces):
        np.testing.assert_array_equal(host_u32(k, w), slot[off:off + w])


def test_fused_extraction_edge_cases(hip, oracle):
    """all hits / all misses / nothing at all / labels only; rows that are not whole 16-byte chunks are refused (the
    callers then take one launch per list)"""
    rs = np.random.default_rng(12)
    n_tab, dim = 4096, 64
    src = rs.standard_normal((n_tab, dim)).astype(np.float32)
    cache = rs.standard_normal((512, dim)).astype(np.float32)
    host = torch.from_numpy(src).pin_memory()
    for n_miss, n_cache in ((0, 3000), (3000, 0), (0, 0), (1, 1)):
        n = n_miss + n_cache
        dst = rs.permutation(max(n, 1))[:n].astype(np.uint32)
        msrc = rs.integers(0, n_tab, size=n_miss).astype(np.uint32)
        csrc = rs.integers(0, 512, size=n_cache).astype(np.uint32)
        want = np.zeros((max(n, 1), dim), np.float32)
        want[dst[:n_miss]] = src[msrc]
        want[dst[n_miss:]] = cache[csrc]
        out = torch.zeros((max(n, 1), dim), dtype=torch.float32, device="cuda")
        lab = torch.zeros(5, dtype=torch.int64, device="cuda")
        hip.extract_fused(out, host, dev(cache), dev(msrc) if n_miss else None, dev(dst[:n_miss]) if n_miss else None,
                          dev(csrc) if n_cache else None, dev(dst[n_miss:]) if n_cache else None, link_workgroups=64,
                          label_out=lab, label_src=dev(np.arange(100, dtype=np.int64) * 3),
                          label_index=dev(np.array([5, 0, 99, 7, 7], np.uint32)))
        torch.cuda.synchronize()
        assert out.cpu().numpy().tobytes() == want.tobytes()
        assert lab.cpu().tolist() == [15, 0, 297, 21, 21]
    odd = torch.zeros((10, 7), dtype=torch.float32, device="cuda")
    with pytest.raises(hip.FgnnError):
        hip.extract_fused(odd, dev(np.zeros((4, 7), np.float32)), dev(np.zeros((4, 7), np.float32)),
                          dev(np.zeros(2, np.uint32)), dev(np.zeros(2, np.uint32)), dev(np.zeros(2, np.uint32)),
                          dev(np.ones(2, np.uint32)))


@pytest.mark.parametrize("kind,fanouts,batch", [("khop2", [25, 10], 2000), ("khop0", [5, 10, 15], 300),
                                                ("khop2", [10, 5], 1)])
def test_layered_pipeline_matches_oracle(hip, oracle, kind, fanouts, batch):
    """DoGPUSample order (cuda_loops.cc:50-267) chained on the device with device-side counts only:
    Reset, FillWithUnique, then per layer sample -> FillWithDuplicates(+map)."""
    from fgnn_hip import synth
    num_node = 100000
    indptr, indices = synth.powerlaw_csr(num_node, 1500000, seed=5)
    d_indptr, d_indices = dev(indptr), dev(indices.copy())
    o_indices = indices.copy()
    st = oracle.KHOP0 if kind == "khop0" else oracle.KHOP2
    max_items = oracle.predict_num_nodes(batch, fanouts)
    ht = hip.HashTable(max_items)
    oht = oracle.HashTable(num_node, max_items)
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    for b in range(3):
        seeds = _seeds(batch, num_node, seed=100 + b)
        want = oracle.do_sample(indptr, o_indices, seeds, fanouts, st, rng, b, oht)
        ht.reset()
        d_seeds = dev(seeds)
        ht.fill_unique(d_seeds)
        cap = batch
        cur, d_cur_n = d_seeds, None
        got = {}
        for li in range(len(fanouts) - 1, -1, -1):
            col, dst, d_ne = hip.sample_khop(kind, d_indptr, d_indices, cur, fanouts[li], SEED, b, li, hip.SRC_LOCAL,
                                             d_num_input=d_cur_n)
            row = ht.fill_duplicates(dst, d_num_items=d_ne)
            got[li] = (row, col, d_ne)
            cap = cap * (1 + fanouts[li])
            # next layer's input = the table's N2O list; its length lives on the device
            n2o_ptr = hip.load().fgnn_hashtable_n2o(ht.h)
            cur = hip._wrap_device_u32(n2o_ptr, min(cap, max_items), ht.device)
            d_cur_n = hip._wrap_device_u32(ht.d_num_items_ptr(), 1, ht.device)
        torch.cuda.synchronize()
        for li in range(len(fanouts)):
            row, col, d_ne = got[li]
            ne = int(d_ne.cpu()[0])
            assert ne == want["graphs"][li]["num_edge"]
            np.testing.assert_array_equal(host_u32(row, ne), want["graphs"][li]["row"])
            np.testing.assert_array_equal(host_u32(col, ne), want["graphs"][li]["col"])
        np.testing.assert_array_equal(host_u32(ht.unique()), want["input_nodes"])
    np.testing.assert_array_equal(host_u32(d_indices), o_indices)


@pytest.mark.parametrize("kind,fanouts,batch,dim", [("khop2", [25, 10], 3000, 128), ("khop0", [5, 10, 15], 200, 100),
                                                    ("khop2", [3], 17, 4), ("khop2", [10, 5], 500, 25),
                                                    ("khop2", [4, 4], 300, 3)])
def test_batch_driver_matches_oracle(hip, oracle, kind, fanouts, batch, dim):
    """fgnn_sampler_sample + cache_index + extract (the C++ per-batch driver) against
    oracle do_sample + get_miss_cache_index + extract, three batches in a row, last one short.
    Row widths: multiples of 16 bytes (the label rows and the summary copy ride on the feature gather launch) and
    100 / 12 bytes (element gather: label gather and summary copy are launches of their own)."""
    from fgnn_hip import synth
    num_node = 120000
    indptr, indices = synth.powerlaw_csr(num_node, 2000000, seed=6)
    feat = synth.node_features(num_node, dim)
    label = np.random.default_rng(1).integers(0, 47, size=num_node).astype(np.int64)
    rank = np.random.default_rng(2).permutation(num_node).astype(np.uint32)
    table = oracle.cache_table_build(rank, num_node // 4, num_node)
    cache_rows = oracle.extract(feat, rank[:num_node // 4])
    d_indices = dev(indices.copy())
    st = hip.KHOP0 if kind == "khop0" else hip.KHOP2
    sampler = hip.Sampler(dev(indptr), d_indices, fanouts, batch, sample_type=st, seed=SEED)
    assert sampler.max_nodes == oracle.predict_num_nodes(batch, fanouts)
    batches = [sampler.new_batch(dim, hip.F32, hip.I64) for _ in range(2)]
    d_feat, d_label, d_table, d_cache = dev(feat), dev(label), dev(table), dev(cache_rows)
    o_indices = indices.copy()
    oht = oracle.HashTable(num_node, sampler.max_nodes)
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    ost = oracle.KHOP0 if kind == "khop0" else oracle.KHOP2
    for b, n in enumerate([batch, batch, max(1, batch // 3)]):
        seeds = _seeds(n, num_node, seed=200 + b)
        bt = batches[b % 2]
        sampler.sample(dev(seeds), 1000 + b, bt)
        bt.cache_index(d_table)
        if b == 1:
            bt.extract_cached(d_cache, d_feat, d_label)  # cache rows from the HBM cache + miss rows by GPU gather
        else:
            bt.extract(d_feat, d_label)
        bt.finish()
        m = bt.wait()
        want = oracle.do_sample(indptr, o_indices, seeds, fanouts, ost, rng, 1000 + b, oht)
        assert m.key == 1000 + b and m.num_layers == len(fanouts) and m.overflow == 0 and m.num_output == n
        for li in range(len(fanouts)):
            row, col, nsrc, ndst = bt.graph(li)
            g = want["graphs"][li]
            assert (len(row), nsrc, ndst) == (g["num_edge"], g["num_src"], g["num_dst"])
            np.testing.assert_array_equal(host_u32(row), g["row"])
            np.testing.assert_array_equal(host_u32(col), g["col"])
        nodes = host_u32(bt.input_nodes())
        np.testing.assert_array_equal(nodes, want["input_nodes"])
        np.testing.assert_array_equal(host_u32(bt.output_nodes()), seeds)
        o_split = oracle.get_miss_cache_index(table, nodes)
        for got, w in zip(bt.cache_index_arrays(), o_split):
            np.testing.assert_array_equal(host_u32(got), w)
        assert m.num_miss + m.num_cache == m.num_input
        assert bt.feat().cpu().numpy().tobytes() == oracle.extract(feat, nodes).tobytes()
        np.testing.assert_array_equal(bt.label().cpu().numpy(), label[seeds])
    np.testing.assert_array_equal(host_u32(d_indices), o_indices)


@pytest.mark.parametrize("kind,fanouts", [("khop2", [10, 5]), ("khop2", [6]), ("khop0", [4, 3, 2]),
                                          ("weighted_khop_prefix", [3, 4])])
def test_batch_in_two_halves_matches_oracle(hip, oracle, kind, fanouts):
    """fgnn_sampler_sample_begin / _end: the sampling chain of batch k + 1 (k + 2) enqueued BEFORE the tail of batch k,
    batches rotating over three streams -- every batch equals the oracle's sequential replay (khop2: the CSR swaps happen
    in chain order), the tails in any order; a tail without its chain, or on another stream, is refused."""
    from fgnn_hip import synth
    num_node, batch, dim = 60000, 700, 16
    indptr, indices = synth.powerlaw_csr(num_node, 900000, seed=16)
    feat = synth.node_features(num_node, dim)
    table = oracle.cache_table_build(np.random.default_rng(3).permutation(num_node).astype(np.uint32), num_node // 5, num_node)
    prefix = synth.prob_prefix_table(indptr, indices) if kind == "weighted_khop_prefix" else None
    d_indices, d_feat, d_table = dev(indices.copy()), dev(feat), dev(table)
    st = {"khop2": hip.KHOP2, "khop0": hip.KHOP0, "weighted_khop_prefix": hip.WEIGHTED_KHOP_PREFIX}[kind]
    sampler = hip.Sampler(dev(indptr), d_indices, fanouts, batch, sample_type=st, seed=SEED,
                          prob_prefix=dev(prefix) if prefix is not None else None)
    nb = 9
    batches = [sampler.new_batch(dim, hip.F32, hip.I64) for _ in range(6)]
    streams = [torch.cuda.Stream() for _ in range(3)]
    seeds = [_seeds(batch if b % 4 else 1 + b, num_node, seed=300 + b) for b in range(nb)]
    d_seeds = [dev(x) for x in seeds]

    def begin(b):
        sampler.sample_begin(b, d_seeds[b], 50 + b, batches[b % 6], stream=streams[b % 3])

    def end(b):
        with torch.cuda.stream(streams[b % 3]):
            sampler.sample_end(b, batches[b % 6], d_table, stream=streams[b % 3])
            batches[b % 6].extract(d_feat, None)
            batches[b % 6].finish()
    with pytest.raises(hip.FgnnError):
        sampler.sample_end(0, batches[0], d_table, stream=streams[0])  # no chain waits for this tail
    # chains run up to two batches ahead of the tails; the tails of a pair in reverse order
    begin(0)
    begin(1)
    with pytest.raises(hip.FgnnError):
        sampler.sample_end(0, batches[0], d_table, stream=streams[1])  # not the chain's stream
    order = [("b", 2), ("e", 1), ("e", 0), ("b", 3), ("e", 2), ("b", 4), ("b", 5), ("e", 4), ("e", 3), ("e", 5)]
    done = set()
    got = {}
    oht = oracle.HashTable(num_node, sampler.max_nodes)
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    o_indices = indices.copy()
    okind = {"khop2": oracle.KHOP2, "khop0": oracle.KHOP0, "weighted_khop_prefix": oracle.WEIGHTED_KHOP_PREFIX}[kind]
    want = [oracle.do_sample(indptr, o_indices, seeds[b], fanouts, okind, rng, 50 + b, oht, prob_prefix=prefix)
            for b in range(6)]

    def check(b):
        bt = batches[b % 6]
        m = bt.wait()
        assert m.key == 50 + b and m.overflow == 0
        for li in range(len(fanouts)):
            row, col, nsrc, ndst = bt.graph(li)
            g = want[b]["graphs"][li]
            assert (len(row), nsrc, ndst) == (g["num_edge"], g["num_src"], g["num_dst"])
            np.testing.assert_array_equal(host_u32(row), g["row"])
            np.testing.assert_array_equal(host_u32(col), g["col"])
        nodes = host_u32(bt.input_nodes())
        np.testing.assert_array_equal(nodes, want[b]["input_nodes"])
        for gotten, w in zip(bt.cache_index_arrays(), oracle.get_miss_cache_index(table, nodes)):
            np.testing.assert_array_equal(host_u32(gotten), w)
        assert bt.feat().cpu().numpy().tobytes() == oracle.extract(feat, nodes).tobytes()
    for what, b in order:
        (begin if what == "b" else end)(b)
    for b in range(6):
        check(b)
    if kind == "khop2":
        np.testing.assert_array_equal(host_u32(d_indices), o_indices)


@pytest.mark.parametrize("fanout", [1, 5, 15])
def test_weighted_prefix_matches_oracle(hip, oracle, fanout):
    from fgnn_hip import synth
    num_node = 4000
    indptr, indices = synth.powerlaw_csr(num_node, 70000, seed=31)
    prefix = synth.prob_prefix_table(indptr, indices)
    d = [dev(indptr), dev(indices), dev(prefix)]
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    for call, n in enumerate([1500, 1, 0, 777]):
        inp = _seeds(n, num_node, seed=40 + call)
        d_inp = dev(inp) if n else torch.empty(0, dtype=torch.int32, device="cuda")
        o_src, o_dst = oracle.sample_weighted_khop_prefix(indptr, indices, prefix, inp, fanout, rng, 9 + call, 2)
        src, dst, d_ne = hip.sample_weighted_khop_prefix(d[0], d[1], d[2], d_inp, fanout, SEED, 9 + call, 2)
        ne = int(d_ne.cpu()[0])
        assert ne == len(o_dst)
        np.testing.assert_array_equal(host_u32(src, ne), o_src)
        np.testing.assert_array_equal(host_u32(dst, ne), o_dst)
        if n > 1:
            assert (np.diff(o_src.astype(np.int64)) >= 0).all()  # ordered by seed id (stable radix sort by src)


@pytest.mark.parametrize("bits", [32, 9])
def test_pair_sort_is_stable_for_any_keys(hip, bits):
    """scan.hip's sort on its own (fgnn_debug_sort_pairs): random 32-bit keys (all four digit passes) and 9-bit keys
    (heavy duplicates: stability decides the order of the values), at sizes either side of the switch from counting to
    one-launch radix passes (8192) and to three-launch passes (65536), and of a tile -- against numpy's stable argsort."""
    rng = np.random.default_rng(bits)
    for n in [0, 1, 255, 257, 8192, 8193, 30000, 65536, 65537, 1 << 20, (1 << 20) + 77]:
        keys = rng.integers(0, 1 << bits, size=n, dtype=np.uint64).astype(np.uint32)
        if n > 3:
            keys[rng.integers(0, n, size=3)] = 0xFFFFFFFF  # the samplers' padding key
        vals = np.arange(n, dtype=np.uint32)
        d_k, d_v = (dev(keys), dev(vals)) if n else (torch.empty(0, dtype=torch.int32, device="cuda"),) * 2
        hip.debug_sort_pairs(d_k, d_v)
        order = np.argsort(keys, kind="stable")
        np.testing.assert_array_equal(host_u32(d_k), keys[order], err_msg="n=%d" % n)
        np.testing.assert_array_equal(host_u32(d_v), vals[order], err_msg="n=%d" % n)


def test_weighted_prefix_orders_many_seeds_of_unknown_range(hip, oracle):
    """The stateless entry point cannot know the id range, so it sorts the seeds (scan.hip, where the reference calls
    cub::DeviceRadixSort, cuda_sampling_weighted_khop_prefix.cu:200-215): by counting up to 8192 seeds, with one launch
    per radix pass up to 32 sort tiles (65536 seeds), with three per pass beyond.  Seed counts either side of each
    switch and of a tile, ids that use three of the four digits; empty rows share one key."""
    from fgnn_hip import synth
    num_node = 1 << 21
    indptr, indices = synth.powerlaw_csr(num_node, 6000000, seed=77)
    prefix = synth.prob_prefix_table(indptr, indices)
    d = [dev(indptr), dev(indices), dev(prefix)]
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    for call, n in enumerate([2048, 2049, 8192, 8193, 22500, 65536, 65537, 300001]):
        inp = _seeds(n, num_node, seed=78 + call)
        assert (np.diff(indptr.astype(np.int64))[inp] == 0).any() and inp.max() >= 1 << 16
        o_src, o_dst = oracle.sample_weighted_khop_prefix(indptr, indices, prefix, inp, 5, rng, 21 + call, 1)
        src, dst, d_ne = hip.sample_weighted_khop_prefix(d[0], d[1], d[2], dev(inp), 5, SEED, 21 + call, 1)
        ne = int(d_ne.cpu()[0])
        assert ne == len(o_dst), n
        np.testing.assert_array_equal(host_u32(src, ne), o_src)
        np.testing.assert_array_equal(host_u32(dst, ne), o_dst)
        assert (np.diff(o_src.astype(np.int64)) >= 0).all()


@pytest.mark.parametrize("walk_len,num_walks,K,restart", [(3, 4, 5, 0.5), (3, 25, 5, 0.5), (2, 70, 3, 0.0),
                                                          (4, 3, 20, 0.9), (3, 1, 2, 0.3), (5, 32, 8, 0.2),
                                                          (2, 33, 4, 0.5), (600, 2, 6, 0.01)])
def test_random_walk_matches_oracle(hip, oracle, walk_len, num_walks, K, restart):
    from fgnn_hip import synth
    num_node = 3000
    indptr, indices = synth.powerlaw_csr(num_node, 40000, seed=33)
    d_indptr, d_indices = dev(indptr), dev(indices)
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    # seed counts that do not fill the last wavefront's seed groups (16 seeds per wave at 4 walks, 2 at 25)
    for call, n in enumerate([1000, 3, 0, 37]):
        inp = _seeds(n, num_node, seed=50 + call)
        d_inp = dev(inp) if n else torch.empty(0, dtype=torch.int32, device="cuda")
        o = oracle.sample_random_walk(indptr, indices, inp, walk_len, restart, num_walks, K, rng, 3 + call, 1)
        src, dst, dat, d_ne = hip.sample_random_walk(d_indptr, d_indices, d_inp, walk_len, restart, num_walks, K, SEED,
                                                     3 + call, 1)
        ne = int(d_ne.cpu()[0])
        assert ne == len(o[0])
        np.testing.assert_array_equal(host_u32(src, ne), o[0])
        np.testing.assert_array_equal(host_u32(dst, ne), o[1])
        np.testing.assert_array_equal(host_u32(dat, ne), o[2])


@pytest.mark.parametrize("mode", ["weighted", "random_walk"])
def test_batch_driver_other_samplers(hip, oracle, mode):
    """BASELINE configs 4/5 in miniature: GCN [5,10,15] weighted-prefix; PinSAGE random walk, 3 layers, K=5."""
    from fgnn_hip import synth
    num_node = 30000
    indptr, indices = synth.powerlaw_csr(num_node, 500000, seed=35)
    batch = 150
    if mode == "weighted":
        prefix = synth.prob_prefix_table(indptr, indices)
        fanouts = [5, 10, 15]
        sampler = hip.Sampler(dev(indptr), dev(indices), fanouts, batch, sample_type=hip.WEIGHTED_KHOP_PREFIX,
                              seed=SEED, prob_prefix=dev(prefix))
        okw = dict(prob_prefix=prefix)
        ost = oracle.WEIGHTED_KHOP_PREFIX
    else:
        fanouts = [5, 5, 5]
        sampler = hip.Sampler(dev(indptr), dev(indices), fanouts, batch, sample_type=hip.RANDOM_WALK, seed=SEED,
                              walk_len=3, num_walks=25, restart_prob=0.5)
        okw = dict(walk_len=3, num_walks=25, num_neighbor=5, restart_prob=0.5)
        ost = oracle.RANDOM_WALK
    bt = sampler.new_batch()
    oht = oracle.HashTable(num_node, sampler.max_nodes)
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    o_indices = indices.copy()
    for b in range(2):
        seeds = _seeds(batch - 13 * b, num_node, seed=60 + b)
        sampler.sample(dev(seeds), b, bt)
        bt.finish()
        m = bt.wait()
        want = oracle.do_sample(indptr, o_indices, seeds, fanouts, ost, rng, b, oht, **okw)
        assert m.overflow == 0
        for li in range(3):
            row, col, nsrc, ndst = bt.graph(li)
            g = want["graphs"][li]
            assert (len(row), nsrc, ndst) == (g["num_edge"], g["num_src"], g["num_dst"])
            np.testing.assert_array_equal(host_u32(row), g["row"])
            np.testing.assert_array_equal(host_u32(col), g["col"])
            if mode == "random_walk":
                np.testing.assert_array_equal(host_u32(bt.graph_data(li)), g["data"])
        np.testing.assert_array_equal(host_u32(bt.input_nodes()), want["input_nodes"])


def test_weighted_prefix_search_trees_on_long_rows(hip, oracle):
    """Rows of the prefix table longer than 64 entries are searched through a 5-ary tree built when the sampler is
    created (prefix_tree.hip) instead of the reference's binary search (cuda_sampling_weighted_khop_prefix.cu:66-86):
    row lengths around the level boundaries (65, 125 / 126, 625 / 626, 3 125 / 3 126, 78 125 / 78 126, ...), rows of equal weights
    (long runs of ties would expose a wrong '<' / '<='), and ONE long row whose prefix sums are not non-decreasing -- it
    must be refused a tree (the answer of a binary search on such a row depends on the probe order) and still match.
    Bit-exact against the oracle's binary search, two layers so that hub rows are also reached as sampled neighbours."""
    lens = [257, 300, 4096, 4097, 65536, 65537, 300000, 256, 17, 1, 0, 1000, 5000, 70000, 64, 65, 125, 126, 625, 626,
            3125, 3126, 15625, 15626, 78125, 78126]
    rng = np.random.default_rng(77)
    num_node = 4000
    deg = np.concatenate([np.array(lens), rng.integers(0, 40, size=num_node - len(lens))]).astype(np.int64)
    indptr = np.zeros(num_node + 1, dtype=np.int64)
    np.cumsum(deg, out=indptr[1:])
    E = int(indptr[-1])
    # neighbours: mostly the special rows, so that layer 0's frontier is full of them
    indices = np.where(rng.random(E) < 0.5, rng.integers(0, len(lens), size=E), rng.integers(0, num_node, size=E))
    indices = indices.astype(np.uint32)
    w = np.where(rng.random(E) < 0.3, 100.0, 1.0).astype(np.float32)
    w[indptr[5]:indptr[6]] = 1.0  # a row of equal weights
    # a row whose sums stall: beyond 1e8 a float moves in steps of 8, so most of these weights change nothing and
    # neighbouring prefix values are EQUAL over long stretches (first-position-of-a-tie semantics)
    w[indptr[13]:indptr[14]] = rng.choice(np.array([1.0, 4.0, 8.0, 16.0], dtype=np.float32), size=lens[13])
    w[indptr[13]] = 1e8
    prefix = np.empty(E, dtype=np.float32)
    for r in range(num_node):
        a, b = indptr[r], indptr[r + 1]
        if b > a:
            prefix[a:b] = np.cumsum(w[a:b], dtype=np.float32)  # sequential f32 accumulation like the tool
    k = indptr[12] + 2500  # row 12 (5 000 entries): two neighbouring sums swapped -> not non-decreasing
    assert prefix[k] < prefix[k + 1]
    prefix[k], prefix[k + 1] = prefix[k + 1], prefix[k]
    indptr = indptr.astype(np.uint32)
    fanouts, batch = [6, 15], 96
    sampler = hip.Sampler(dev(indptr), dev(indices), fanouts, batch, sample_type=hip.WEIGHTED_KHOP_PREFIX, seed=SEED,
                          prob_prefix=dev(prefix))
    long_rows, refused, nbytes = sampler.prefix_tree_stats()
    assert long_rows == sum(1 for x in lens if x > 64) and refused == 1 and nbytes > 0
    bt = sampler.new_batch()
    oht = oracle.HashTable(num_node, sampler.max_nodes)
    rng_o = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    o_indices = indices.copy()
    for b in range(3):
        seeds = np.concatenate([np.arange(len(lens)), 100 + rng.permutation(1000)[:batch - len(lens)]]).astype(np.uint32)
        sampler.sample(dev(seeds), b, bt)
        bt.finish()
        m = bt.wait()
        want = oracle.do_sample(indptr, o_indices, seeds, fanouts, oracle.WEIGHTED_KHOP_PREFIX, rng_o, b, oht,
                                prob_prefix=prefix)
        assert m.overflow == 0
        for li in range(2):
            row, col, nsrc, ndst = bt.graph(li)
            g = want["graphs"][li]
            assert (len(row), nsrc, ndst) == (g["num_edge"], g["num_src"], g["num_dst"])
            np.testing.assert_array_equal(host_u32(row), g["row"])
            np.testing.assert_array_equal(host_u32(col), g["col"])
        np.testing.assert_array_equal(host_u32(bt.input_nodes()), want["input_nodes"])


@pytest.mark.parametrize("batch", [12000, 44000])
def test_batch_driver_random_walk_large_capacities(hip, oracle, batch):
    """PinSAGE walks through the batch driver at worst-case frontier capacities of 432 K and 1.58 M seeds: the one-launch
    emit (rw_emit_sp_kernel) then gives a lane 4 / 16 consecutive seeds (150-seed batches elsewhere: one)."""
    from fgnn_hip import synth
    num_node = 60000
    indptr, indices = synth.powerlaw_csr(num_node, 600000, seed=37)
    fanouts = [5, 5, 5]
    sampler = hip.Sampler(dev(indptr), dev(indices), fanouts, batch, sample_type=hip.RANDOM_WALK, seed=SEED,
                          walk_len=3, num_walks=4, restart_prob=0.5)
    bt = sampler.new_batch()
    oht = oracle.HashTable(num_node, sampler.max_nodes)
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    seeds = _seeds(batch, num_node, seed=80)
    sampler.sample(dev(seeds), 3, bt)
    bt.finish()
    m = bt.wait()
    want = oracle.do_sample(indptr, indices.copy(), seeds, fanouts, oracle.RANDOM_WALK, rng, 3, oht, walk_len=3,
                            num_walks=4, num_neighbor=5, restart_prob=0.5)
    assert m.overflow == 0
    for li in range(3):
        row, col, nsrc, ndst = bt.graph(li)
        g = want["graphs"][li]
        assert (len(row), nsrc, ndst) == (g["num_edge"], g["num_src"], g["num_dst"])
        np.testing.assert_array_equal(host_u32(row), g["row"])
        np.testing.assert_array_equal(host_u32(col), g["col"])
        np.testing.assert_array_equal(host_u32(bt.graph_data(li)), g["data"])
    np.testing.assert_array_equal(host_u32(bt.input_nodes()), want["input_nodes"])


@pytest.mark.parametrize("kind", ["weighted_khop_prefix", "khop1"])
@pytest.mark.parametrize("batch", [150, 8000, 10000])
def test_batch_driver_bitmap_seed_ranking(hip, oracle, kind, batch):
    """The with-replacement samplers of the batch driver order their seeds by counting bits in a bitmap over the node ids
    (sample_weighted.hip; the reference radix-sorts all pairs, cuda_sampling_weighted_khop_prefix.cu:148-255).  Shapes that
    reach every variant: a graph of several bitmap tiles (rank_prefix_kernel's cross-workgroup prefix), and worst-case
    frontier capacities of 1.4 M and 1.76 M seeds (weighted_emit_sp_kernel with 4 and 16 ranks per lane; batch 150: 1)."""
    from fgnn_hip import synth
    num_node = 400000  # 12 500 bitmap words = 4 tiles of 4096
    indptr, indices = synth.powerlaw_csr(num_node, 4000000, seed=36)
    fanouts = [5, 10, 15]
    if kind == "khop1":
        sampler = hip.Sampler(dev(indptr), dev(indices), fanouts, batch, sample_type=hip.KHOP1, seed=SEED)
        okw, ost = {}, oracle.KHOP1
    else:
        prefix = synth.prob_prefix_table(indptr, indices)
        sampler = hip.Sampler(dev(indptr), dev(indices), fanouts, batch, sample_type=hip.WEIGHTED_KHOP_PREFIX,
                              seed=SEED, prob_prefix=dev(prefix))
        okw, ost = dict(prob_prefix=prefix), oracle.WEIGHTED_KHOP_PREFIX
    bt = sampler.new_batch()
    oht = oracle.HashTable(num_node, sampler.max_nodes)
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    o_indices = indices.copy()
    for b in range(2):  # the second batch finds the bitmap as the first one left it: all zero
        seeds = _seeds(batch - 13 * b, num_node, seed=70 + b)
        sampler.sample(dev(seeds), b, bt)
        bt.finish()
        m = bt.wait()
        want = oracle.do_sample(indptr, o_indices, seeds, fanouts, ost, rng, b, oht, **okw)
        assert m.overflow == 0
        for li in range(3):
            row, col, nsrc, ndst = bt.graph(li)
            g = want["graphs"][li]
            assert (len(row), nsrc, ndst) == (g["num_edge"], g["num_src"], g["num_dst"])
            np.testing.assert_array_equal(host_u32(row), g["row"])
            np.testing.assert_array_equal(host_u32(col), g["col"])
        np.testing.assert_array_equal(host_u32(bt.input_nodes()), want["input_nodes"])


@pytest.mark.parametrize("kind", ["khop1", "weighted_khop", "weighted_khop_hash_dedup"])
@pytest.mark.parametrize("fanout", [1, 6, 15, 50])
def test_with_replacement_samplers_match_oracle(hip, oracle, kind, fanout):
    """khop1 (uniform with replacement), weighted_khop (alias method) and weighted_khop_hash_dedup (alias draws,
    repeated values rejected): SURVEY 8(f) rank 4."""
    from fgnn_hip import synth
    num_node = 4000
    indptr, indices = synth.powerlaw_csr(num_node, 70000, seed=37)
    prob, alias = synth.alias_tables(indptr, indices)
    d_indptr, d_indices, d_prob, d_alias = dev(indptr), dev(indices), dev(prob), dev(alias)
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    for call, n in enumerate([1200, 1, 0, 501]):
        inp = _seeds(n, num_node, seed=70 + call)
        d_inp = dev(inp) if n else torch.empty(0, dtype=torch.int32, device="cuda")
        if kind == "khop1":
            o_src, o_dst = oracle.sample_khop1(indptr, indices, inp, fanout, rng, 4 + call, 1)
        elif kind == "weighted_khop_hash_dedup":
            o_src, o_dst = oracle.sample_weighted_khop_hash_dedup(indptr, indices, prob, alias, inp, fanout, rng,
                                                                  4 + call, 1)
        else:
            o_src, o_dst = oracle.sample_weighted_khop(indptr, indices, prob, alias, inp, fanout, rng, 4 + call, 1)
        src, dst, d_ne = hip.sample_with_replacement(kind, d_indptr, d_indices, d_inp, fanout, SEED, 4 + call, 1,
                                                     prob=d_prob, alias=d_alias)
        ne = int(d_ne.cpu()[0])
        assert ne == len(o_dst)
        np.testing.assert_array_equal(host_u32(src, ne), o_src)
        np.testing.assert_array_equal(host_u32(dst, ne), o_dst)


def test_hash_dedup_gives_up_on_rows_without_enough_distinct_values(hip, oracle):
    """Rows longer than the fanout whose neighbours (and aliases) hold fewer distinct values than the fanout: the
    reference never returns; the build stops after 64 * fanout attempts and emits what it found."""
    num_node, fanout = 64, 6
    deg = np.full(num_node, 12, dtype=np.uint32)
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.uint32)
    rows = np.arange(num_node, dtype=np.uint32)[:, None]
    indices = ((rows + (np.arange(12, dtype=np.uint32)[None, :] % 2) + 1) % num_node).astype(np.uint32).ravel()  # 2 values
    prob = np.full(indices.shape, 0.5, dtype=np.float32)
    alias = indices.copy()
    alias[::3] = (alias[::3] + 7) % num_node  # a third distinct value through the alias table for some rows
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    inp = np.arange(num_node, dtype=np.uint32)
    o_src, o_dst = oracle.sample_weighted_khop_hash_dedup(indptr, indices, prob, alias, inp, fanout, rng, 3, 0)
    src, dst, d_ne = hip.sample_with_replacement("weighted_khop_hash_dedup", dev(indptr), dev(indices), dev(inp), fanout,
                                                 SEED, 3, 0, prob=dev(prob), alias=dev(alias))
    ne = int(d_ne.cpu()[0])
    assert ne == len(o_dst) and 2 * num_node <= ne < fanout * num_node
    np.testing.assert_array_equal(host_u32(src, ne), o_src)
    np.testing.assert_array_equal(host_u32(dst, ne), o_dst)


@pytest.mark.parametrize("mode", ["khop1", "weighted_khop", "weighted_khop_hash_dedup"])
def test_batch_driver_with_replacement(hip, oracle, mode):
    from fgnn_hip import synth
    num_node = 20000
    indptr, indices = synth.powerlaw_csr(num_node, 300000, seed=39)
    prob, alias = synth.alias_tables(indptr, indices)
    fanouts, batch = [4, 6], 200
    if mode == "khop1":
        sampler = hip.Sampler(dev(indptr), dev(indices), fanouts, batch, sample_type=hip.KHOP1, seed=SEED)
        okw, ost = {}, oracle.KHOP1
    else:
        hst, ost = ((hip.WEIGHTED_KHOP, oracle.WEIGHTED_KHOP) if mode == "weighted_khop" else
                    (hip.WEIGHTED_KHOP_HASH_DEDUP, oracle.WEIGHTED_KHOP_HASH_DEDUP))
        sampler = hip.Sampler(dev(indptr), dev(indices), fanouts, batch, sample_type=hst, seed=SEED,
                              prob_table=dev(prob), alias_table=dev(alias))
        okw = dict(prob_prefix=prob, alias_table=alias)
    bt = sampler.new_batch()
    oht = oracle.HashTable(num_node, sampler.max_nodes)
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    o_indices = indices.copy()
    for b in range(2):
        seeds = _seeds(batch, num_node, seed=80 + b)
        sampler.sample(dev(seeds), b, bt)
        bt.finish()
        bt.wait()
        want = oracle.do_sample(indptr, o_indices, seeds, fanouts, ost, rng, b, oht, **okw)
        for li in range(2):
            row, col, nsrc, ndst = bt.graph(li)
            g = want["graphs"][li]
            assert (len(row), nsrc, ndst) == (g["num_edge"], g["num_src"], g["num_dst"])
            np.testing.assert_array_equal(host_u32(row), g["row"])
            np.testing.assert_array_equal(host_u32(col), g["col"])
        np.testing.assert_array_equal(host_u32(bt.input_nodes()), want["input_nodes"])


@pytest.mark.gpu
def test_batch_driver_large_frontier(hip, oracle):
    """A frontier of millions of edges / nodes (GCN-style fanout [5,10,15]): the single-launch dedup and cache-split
    kernels then run many rounds per workgroup and the worst-case capacity exceeds 4 M items."""
    from fgnn_hip import synth
    num_node, batch, fanouts, dim = 6_000_000, 12000, [6, 10, 15], 4
    indptr, indices = synth.powerlaw_csr(num_node, 90_000_000, seed=12)
    rank = np.random.default_rng(2).permutation(num_node).astype(np.uint32)
    table = oracle.cache_table_build(rank, num_node // 5, num_node)
    feat = synth.node_features(num_node, dim)
    label = np.random.default_rng(1).integers(0, 47, size=num_node).astype(np.int64)
    d_indices = dev(indices.copy())
    sampler = hip.Sampler(dev(indptr), d_indices, fanouts, batch, sample_type=hip.KHOP2, seed=SEED)
    assert sampler.max_nodes == oracle.predict_num_nodes(batch, fanouts) > (4 << 20)
    bt = sampler.new_batch(dim, hip.F32, hip.I64)
    d_feat, d_label, d_table = dev(feat), dev(label), dev(table)
    o_indices = indices.copy()
    oht = oracle.HashTable(num_node, sampler.max_nodes)
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    for b in range(2):
        seeds = _seeds(batch, num_node, seed=300 + b)
        if b == 0:
            sampler.sample(dev(seeds), 50 + b, bt)
            bt.cache_index(d_table)
        else:  # sample + split as one call (what an arch5 sampler does), > 4 rounds per workgroup in the split
            sampler.sample_indexed(dev(seeds), 50 + b, bt, d_table)
        bt.extract(d_feat, d_label)
        bt.finish()
        m = bt.wait()
        want = oracle.do_sample(indptr, o_indices, seeds, fanouts, oracle.KHOP2, rng, 50 + b, oht)
        assert m.overflow == 0 and m.num_input == len(want["input_nodes"]) > 1_600_000  # > 4 rounds of 1536 x 256
        for li in range(len(fanouts)):
            row, col, nsrc, ndst = bt.graph(li)
            g = want["graphs"][li]
            assert (len(row), nsrc, ndst) == (g["num_edge"], g["num_src"], g["num_dst"])
            np.testing.assert_array_equal(host_u32(row), g["row"])
            np.testing.assert_array_equal(host_u32(col), g["col"])
        nodes = host_u32(bt.input_nodes())
        np.testing.assert_array_equal(nodes, want["input_nodes"])
        for got, w in zip(bt.cache_index_arrays(), oracle.get_miss_cache_index(table, nodes)):
            np.testing.assert_array_equal(host_u32(got), w)
        assert bt.feat().cpu().numpy().tobytes() == oracle.extract(feat, nodes).tobytes()
    np.testing.assert_array_equal(host_u32(d_indices), o_indices)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["khop1", "khop2"])
def test_batch_driver_layer_beyond_the_one_launch_dedup(hip, oracle, kind):
    """fanout [15,15,15] x batch 4000: the last layer's worst case (15.4 M edges) is beyond the one-launch count+assign
    (~12.6 M items), so that fill takes the global-table passes.  khop1 never inserts into the table itself: its early,
    small layers must then NOT go through the partitioned table-free path either, or the last fill would dedup against a
    table that only knows the seeds (input_nodes with duplicates; the table-free decision is per batch, engine.hip)."""
    from fgnn_hip import synth
    num_node, batch, fanouts = 300_000, 4000, [15, 15, 15]
    indptr, indices = synth.powerlaw_csr(num_node, 4_000_000, seed=23)
    d_indices = dev(indices.copy())
    st, ost = (hip.KHOP1, oracle.KHOP1) if kind == "khop1" else (hip.KHOP2, oracle.KHOP2)
    sampler = hip.Sampler(dev(indptr), d_indices, fanouts, batch, sample_type=st, seed=SEED)
    assert sampler.max_edges(0) > 13_000_000
    bt = sampler.new_batch()
    oht = oracle.HashTable(num_node, sampler.max_nodes)
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    o_indices = indices.copy()
    for b in range(2):
        seeds = _seeds(batch, num_node, seed=500 + b)
        sampler.sample(dev(seeds), b, bt)
        bt.finish()
        m = bt.wait()
        want = oracle.do_sample(indptr, o_indices, seeds, fanouts, ost, rng, b, oht)
        assert m.overflow == 0
        nodes = host_u32(bt.input_nodes())
        assert len(np.unique(nodes)) == len(nodes)
        np.testing.assert_array_equal(nodes, want["input_nodes"])
        for li in range(3):
            row, col, nsrc, ndst = bt.graph(li)
            g = want["graphs"][li]
            assert (len(row), nsrc, ndst) == (g["num_edge"], g["num_src"], g["num_dst"])
            np.testing.assert_array_equal(host_u32(row), g["row"])
            np.testing.assert_array_equal(host_u32(col), g["col"])
    np.testing.assert_array_equal(host_u32(d_indices), o_indices)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["khop2", "khop0", "khop1", "random_walk"])
def test_run_batch_with_a_cache_table_then_a_second_split(hip, oracle, kind):
    """fgnn_sampler_run_batch with a cache table (sample + GetMissCacheIndex in one call, cuda_cache.cu:33-158), per
    sampler family: the fused k-hop inserts, the partitioned table-free fills (khop1, random walk); empty batches,
    batches of isolated seeds (no fill launches at all), a short batch after a long one on the same buffer, and a
    second split of the same batch against ANOTHER table (nothing of the first split may be reused).  (Written for a
    variant that looked the table up inside the dedup launches -- profiles/NOTES_rejected_experiments.md, round 5 --
    and kept: it is the only test that splits one batch twice.)"""
    from fgnn_hip import synth
    num_node = 60000
    indptr, indices = synth.powerlaw_csr(num_node, 900000, seed=44)
    deg = np.diff(indptr.astype(np.int64))
    isolated = np.nonzero(deg == 0)[0].astype(np.uint32)
    assert len(isolated) >= 20
    fanouts, batch = ([5, 5, 5], 400) if kind == "random_walk" else ([7, 5], 700)
    hst, ost = dict(khop2=(hip.KHOP2, oracle.KHOP2), khop0=(hip.KHOP0, oracle.KHOP0), khop1=(hip.KHOP1, oracle.KHOP1),
                    random_walk=(hip.RANDOM_WALK, oracle.RANDOM_WALK))[kind]
    skw = dict(walk_len=3, num_walks=8, restart_prob=0.5) if kind == "random_walk" else {}
    okw = dict(walk_len=3, num_walks=8, num_neighbor=5, restart_prob=0.5) if kind == "random_walk" else {}
    d_indices = dev(indices.copy())
    o_indices = indices.copy()
    sampler = hip.Sampler(dev(indptr), d_indices, fanouts, batch, sample_type=hst, seed=SEED, **skw)
    bt = sampler.new_batch()
    rs = np.random.default_rng(9)
    tables = [oracle.cache_table_build(rs.permutation(num_node).astype(np.uint32), num_node // k, num_node) for k in (4, 2)]
    d_tables = [dev(t) for t in tables]
    oht = oracle.HashTable(num_node, sampler.max_nodes)
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    plans = [_seeds(batch, num_node, seed=1), np.empty(0, dtype=np.uint32), isolated[:20], _seeds(33, num_node, seed=2),
             _seeds(batch, num_node, seed=3)]
    for b, seeds in enumerate(plans):
        d_seeds = dev(seeds) if len(seeds) else torch.empty(0, dtype=torch.int32, device="cuda")
        sampler.run_batch(b, d_seeds, 900 + b, bt, d_tables[0])
        m = bt.wait()
        assert m.overflow == 0 and m.num_output == len(seeds)
        if len(seeds) == 0:
            assert m.num_input == 0 and m.num_miss == 0 and m.num_cache == 0
            continue
        want = oracle.do_sample(indptr, o_indices, seeds, fanouts, ost, rng, 900 + b, oht, **okw)
        nodes = host_u32(bt.input_nodes())
        np.testing.assert_array_equal(nodes, want["input_nodes"])
        for got, w in zip(bt.cache_index_arrays(), oracle.get_miss_cache_index(tables[0], nodes)):
            np.testing.assert_array_equal(host_u32(got), w)
        assert m.num_miss + m.num_cache == m.num_input
        # the same batch against the other table
        bt.cache_index(d_tables[1])
        bt.finish()
        m2 = bt.wait()
        for got, w in zip(bt.cache_index_arrays(), oracle.get_miss_cache_index(tables[1], nodes)):
            np.testing.assert_array_equal(host_u32(got), w)
        assert m2.num_miss + m2.num_cache == m2.num_input == len(nodes)
    np.testing.assert_array_equal(host_u32(d_indices), o_indices)


@pytest.mark.gpu
def test_samplers_created_and_driven_from_two_threads(hip, oracle):
    """Two host threads, each with its own sampler, batches in flight at the same time (an engine process whose sampler
    and pre-sampler threads both use the kernel library; a second GPU's sampler in one process): per-device launch
    attributes and occupancy look-ups are cached behind the C ABI (hashtable_partition.hip: the dynamic-LDS attribute
    is asked once per DEVICE) and must not depend on which thread or device asked first.  Results stay the oracle's."""
    import threading
    errs = []

    def run(kind, fanouts, batch, dim):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(2):
                    test_batch_driver_matches_oracle(hip, oracle, kind, fanouts, batch, dim)
        except BaseException as e:  # noqa: BLE001 (reported by the main thread)
            errs.append((kind, repr(e)))
    ths = [threading.Thread(target=run, args=a) for a in (("khop2", [25, 10], 3000, 128), ("khop0", [5, 10, 15], 200, 100),
                                                          ("khop2", [3], 17, 4))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs


@pytest.mark.gpu
@pytest.mark.parametrize("limit", [0, 48])
def test_partitioned_last_fill_falls_back_per_bin(hip, oracle, limit):
    """The batch's last dedup fill is partitioned by hash and deduplicated per bin in an LDS table
    (hashtable_partition.hip; reference: the global-table insert of cuda_hashtable.cu:176-211).  A bin with more
    distinct keys than its LDS table may hold goes through the global table instead -- forced here for every bin
    (limit 0) and for the fuller bins only (limit 48: LDS and global bins side by side in one fill); the batch-driver
    parity tests (oracle, bit-exact: blocks, unique list, cache split, features) must hold unchanged.  The default
    limit is what every other batch-driver test in this file runs with."""
    L = hip.load()
    L.fgnn_debug_set_partition_lds_limit(limit)
    try:
        test_batch_driver_matches_oracle(hip, oracle, "khop2", [25, 10], 3000, 128)
        test_batch_driver_matches_oracle(hip, oracle, "khop2", [3], 17, 4)
        test_batch_driver_other_samplers(hip, oracle, "weighted")
        test_batch_driver_other_samplers(hip, oracle, "random_walk")
        test_batch_driver_empty_and_isolated_batches(hip, oracle, "khop2")
        test_batch_driver_hub_nodes_in_every_row(hip, oracle)
    finally:
        L.fgnn_debug_set_partition_lds_limit(-1)


@pytest.mark.gpu
def test_hash_dedup_large_frontier_two_pass(hip, oracle):
    """More than 2048 workgroups of seeds: the sampler takes the count / scan / emit route instead of the look-back."""
    from fgnn_hip import synth
    num_node, fanout = 700_000, 3
    indptr, indices = synth.powerlaw_csr(num_node, 6_000_000, seed=44)
    rng_np = np.random.default_rng(8)
    prob = rng_np.random(len(indices), dtype=np.float32)
    alias = np.roll(indices, 1).astype(np.uint32)  # any node id will do for parity
    inp = rng_np.permutation(num_node)[:600_000].astype(np.uint32)
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    o_src, o_dst = oracle.sample_weighted_khop_hash_dedup(indptr, indices, prob, alias, inp, fanout, rng, 7, 2)
    src, dst, d_ne = hip.sample_with_replacement("weighted_khop_hash_dedup", dev(indptr), dev(indices), dev(inp), fanout,
                                                 SEED, 7, 2, prob=dev(prob), alias=dev(alias))
    ne = int(d_ne.cpu()[0])
    assert ne == len(o_dst) > 1_000_000
    np.testing.assert_array_equal(host_u32(src, ne), o_src)
    np.testing.assert_array_equal(host_u32(dst, ne), o_dst)


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [128, 7])
def test_masked_gather_is_the_reference_mock_extraction(hip, oracle, dim):
    """SAMGRAPH_EMPTY_FEAT=k: the feature table holds 2^k rows and node ids are ANDed with 2^k - 1 before indexing it
    (cpu_extraction.cc:47-62); both gather kernels (16-byte chunks, element-wise)."""
    rs = np.random.default_rng(21)
    k = 10
    src = rs.standard_normal((1 << k, dim)).astype(np.float32)
    idx = rs.integers(0, 1 << 30, size=5003).astype(np.uint32)
    out = torch.empty((len(idx), dim), dtype=torch.float32, device="cuda")
    hip.gather_rows(out, dev(src), src_index=dev(idx), src_row_mask=(1 << k) - 1)
    assert out.cpu().numpy().tobytes() == src[idx & ((1 << k) - 1)].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("dim,weighted,sorted_col", [(128, False, True), (256, True, True), (100, False, True),
                                                    (7, True, False), (64, False, False)])
def test_block_aggregate_matches_torch_reference(hip, dim, weighted, sorted_col):
    """fgnn_block_aggregate (fp32, atomics at segment boundaries) against the plain PyTorch fp32 formulation
    out.index_add_(0, col, w * h[row]) and its autograd gradient; tolerance rtol 1e-4 / atol 1e-4 (the order of the
    float additions differs)."""
    from fgnn_hip.nn import block_aggregate
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    E, nsrc, ndst = 60001, 20000, 3000
    row = torch.randint(0, nsrc, (E,), device="cuda", generator=g, dtype=torch.int32)
    col = torch.randint(0, ndst, (E,), device="cuda", generator=g, dtype=torch.int32)
    if sorted_col:
        col = torch.sort(col)[0]
    w = torch.rand(E, device="cuda", generator=g) if weighted else None
    h1 = torch.randn(nsrc, dim, device="cuda", generator=g, requires_grad=True)
    h2 = h1.detach().clone().requires_grad_(True)
    out = block_aggregate(h1, row, col, ndst, w)
    msg = h2[row.long()] if w is None else h2[row.long()] * w.unsqueeze(1)
    ref = torch.zeros((ndst, dim), device="cuda").index_add_(0, col.long(), msg)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)
    gout = torch.randn(ndst, dim, device="cuda", generator=g)
    out.backward(gout)
    ref.backward(gout)
    torch.testing.assert_close(h1.grad, h2.grad, rtol=1e-4, atol=1e-4)
    # empty block
    e = torch.empty(0, dtype=torch.int32, device="cuda")
    assert float(block_aggregate(h1.detach(), e, e, 5).abs().sum()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("layer", ["sage", "gcn", "pinsage"])
def test_example_layers_same_with_fused_and_torch_aggregation(hip, layer):
    """examples/models.py: each layer gives the same output and input gradient whether its message passing runs in
    fgnn_block_aggregate or in torch's gather + index_add_ (rtol / atol 1e-4, fp32)."""
    import importlib
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import models
    importlib.reload(models)
    assert models._fused_aggregate is not None

    class Block:
        def __init__(self, row, col, ndst, w=None):
            self.row, self.col, self.ndst = row, col, ndst
            self.edata = {"weights": w} if w is not None else {}

        def number_of_dst_nodes(self):
            return self.ndst

    g = torch.Generator(device="cuda")
    g.manual_seed(9)
    E, nsrc, ndst, din, dout = 30000, 9000, 1200, 96, 64
    row = torch.randint(0, nsrc, (E,), device="cuda", generator=g, dtype=torch.int32)
    col = torch.sort(torch.randint(0, ndst, (E,), device="cuda", generator=g, dtype=torch.int32))[0]
    w = torch.randint(1, 9, (E,), device="cuda", generator=g, dtype=torch.int32)
    blk = Block(row, col, ndst, w if layer == "pinsage" else None)
    torch.manual_seed(0)
    mod = {"sage": lambda: models.SAGEConvMean(din, dout), "gcn": lambda: models.GraphConv(din, dout),
           "pinsage": lambda: models.WeightedSAGEConv(din, 48, dout, 0.0)}[layer]().cuda()
    outs = []
    for fused in (True, False):
        saved = models._fused_aggregate
        if not fused:
            models._fused_aggregate = None
        x = torch.randn(nsrc, din, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3),
                        requires_grad=True)
        y = mod(blk, x)
        y.square().sum().backward()
        outs.append((y.detach(), x.grad.detach()))
        models._fused_aggregate = saved
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(outs[0][1], outs[1][1], rtol=1e-3, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("din,dout,first_layer", [(128, 256, True), (256, 172, False), (100, 47, False)])
def test_fused_sage_layer_matches_the_op_by_op_layer(hip, din, dout, first_layer):
    """examples/models.py FusedSAGEConv (one autograd node, one GEMM over [h_dst | mean h_u]) against SAGEConvMean
    with the same weights and against the plain torch formulation without the HIP aggregation kernel: output,
    gradients of both weight halves, the bias and (where the input needs one) the input, fp32, rtol 1e-4 (the GEMMs
    group their additions differently).  Reference: dgl.nn.SAGEConv('mean') as train_graphsage.py:24-51 uses it."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "examples"))
    models = importlib.import_module("models")
    from samgraph.torch.adapter import CooBlock
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    E, nsrc, ndst = 50003, 30000, 2500
    row = torch.randint(0, nsrc, (E,), device="cuda", generator=g, dtype=torch.int32)
    col = torch.sort(torch.randint(0, ndst - 7, (E,), device="cuda", generator=g, dtype=torch.int32))[0]  # 7 dst without edges
    fused, plain = models.FusedSAGEConv(din, dout).cuda(), models.SAGEConvMean(din, dout).cuda()
    with torch.no_grad():
        plain.fc_self.weight.copy_(fused.weight[:, :din])
        plain.fc_neigh.weight.copy_(fused.weight[:, din:])
        plain.fc_neigh.bias.copy_(fused.bias)
    x = torch.randn(nsrc, din, device="cuda", generator=g)
    gy = torch.randn(ndst, dout, device="cuda", generator=g)
    outs = []
    for layer in (fused, plain):
        h = x.clone().requires_grad_(not first_layer)
        y = layer(CooBlock(row, col, nsrc, ndst), h)
        y.backward(gy)
        outs.append((y.detach(), None if first_layer else h.grad))
    tol = dict(rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(outs[0][0], outs[1][0], **tol)
    if not first_layer:
        torch.testing.assert_close(outs[0][1], outs[1][1], **tol)
    torch.testing.assert_close(fused.weight.grad[:, :din], plain.fc_self.weight.grad, rtol=1e-4, atol=2e-3)
    torch.testing.assert_close(fused.weight.grad[:, din:], plain.fc_neigh.weight.grad, rtol=1e-4, atol=2e-3)
    torch.testing.assert_close(fused.bias.grad, plain.fc_neigh.bias.grad, rtol=1e-4, atol=2e-3)
    # and against plain torch without any kernel of this repo
    h2 = x.clone()
    msg = torch.zeros((ndst, din), device="cuda").index_add_(0, col.long(), h2[row.long()])
    deg = torch.zeros(ndst, device="cuda").index_add_(0, col.long(), torch.ones(E, device="cuda")).clamp(min=1)
    ref = h2[:ndst] @ fused.weight[:, :din].t() + (msg / deg[:, None]) @ fused.weight[:, din:].t() + fused.bias
    torch.testing.assert_close(outs[0][0], ref.detach(), **tol)


@pytest.mark.gpu
@pytest.mark.parametrize("fused_adam,tune_gemms", [(False, False), (True, False), (True, True)])
def test_graphed_training_step_equals_the_eager_step(hip, fused_adam, tune_gemms):
    """examples/graphed_step.py: the GraphSAGE training step replayed as a captured HIP graph on the batch buffers'
    full-capacity tensors, sizes rounded up to buckets, padded edges pointed at a discarded row -- against the same
    steps taken op by op (the reference's loop, train_graphsage.py:300-330) from the same initial weights on the same
    batches, dropout 0: every parameter after three steps within rtol 1e-4 (fp32; the padded rows add exact zeros to
    the weight gradients but change how the GEMMs group their sums).  tune_gemms: every shape's forward + backward runs
    once more before its capture, with PyTorch's TunableOp choosing the GEMM kernels -- nothing of the training state
    may move in that pass, and the chosen kernels give the same step within the same tolerance."""
    import copy
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "examples"))
    from graphed_step import GraphedSageStep
    from models import SAGE
    from samgraph.torch.adapter import CooBlock
    from fgnn_hip import synth
    num_node, dim, ncls, batch, fanouts = 60000, 64, 13, 512, [8, 5]
    indptr, indices = synth.powerlaw_csr(num_node, 900000, seed=8)
    feat = np.random.default_rng(3).standard_normal((num_node, dim)).astype(np.float32)
    label = np.random.default_rng(4).integers(0, ncls, size=num_node).astype(np.int64)
    d_feat, d_label = dev(feat), dev(label)
    sampler = hip.Sampler(dev(indptr), dev(indices.copy()), fanouts, batch, sample_type=hip.KHOP2, seed=SEED)
    bufs = [sampler.new_batch(dim, hip.F32, hip.I64) for _ in range(2)]
    torch.manual_seed(0)
    model_e = SAGE(dim, 32, ncls, 2, 0.0).cuda()
    model_g = copy.deepcopy(model_e)
    loss_fcn = torch.nn.CrossEntropyLoss()
    opt_e = torch.optim.Adam(model_e.parameters(), lr=0.01, fused=True, capturable=True)
    opt_g = torch.optim.Adam(model_g.parameters(), lr=0.01, fused=True, capturable=True)
    if fused_adam:  # the one-launch Adam with its step count on the device (fgnn_hip.nn.Adam), as bench.py's train legs
        from fgnn_hip.nn import Adam
        opt_g = Adam(model_g.parameters(), lr=0.01)
        model_g.dropout_step = opt_g.step_count  # (dropout 0.0 here: the fused ReLU launch without a mask)
    stepper = GraphedSageStep(model_g, opt_g, loss_fcn, batch, edge_bucket=2048, node_bucket=1024, inner_bucket=256,
                              tune_gemms=tune_gemms)
    losses = []
    for b in range(4):
        bt = bufs[b % 2]
        sampler.run_batch(b, dev(_seeds(batch, num_node, seed=500 + b)), b, bt, None, d_feat, d_label)
        m = bt.wait()
        blocks = [CooBlock(*bt.graph(l)) for l in range(2)]
        loss_e = loss_fcn(model_e(blocks, bt.feat()), bt.label())
        opt_e.zero_grad()
        loss_e.backward()
        opt_e.step()
        loss_g = stepper.step(bt, CooBlock)
        losses.append((float(loss_e), float(loss_g)))
        assert int(m.num_output) == batch
    assert not torch.cuda.tunable.is_enabled()  # the stepper leaves the process-wide switch as it found it
    assert stepper.eager_steps == 1 and stepper.replays == 3 and 1 <= len(stepper.graphs) <= 3
    assert stepper.tuned_shapes == (len(stepper.graphs) if tune_gemms else 0)
    for le, lg in losses:
        assert abs(le - lg) <= 1e-4 * max(1.0, abs(le)), losses
    for pe, pg in zip(model_e.parameters(), model_g.parameters()):
        torch.testing.assert_close(pg, pe, rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_tall_linear_matches_nn_linear():
    """examples/models.py TallLinear (weight gradient as a batched GEMM over row slices) against nn.Linear, fp32,
    rtol 1e-4."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import models
    torch.manual_seed(1)
    a, b = models.TallLinear(96, 40).cuda(), torch.nn.Linear(96, 40).cuda()
    b.load_state_dict(a.state_dict())
    x1 = torch.randn(20011, 96, device="cuda", requires_grad=True)
    x2 = x1.detach().clone().requires_grad_(True)
    g = torch.randn(20011, 40, device="cuda")
    a(x1).backward(g)
    b(x2).backward(g)
    torch.testing.assert_close(a(x1), b(x2), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(x1.grad, x2.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(a.weight.grad, b.weight.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(a.bias.grad, b.bias.grad, rtol=1e-4, atol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 255, 256, 257, 3000])
def test_extract_neighbour_matches_oracle(hip, oracle, graph, n):
    """GPUExtractNeighbour (cuda_extract_neighbour.cu:111-169): all neighbours, input order, exact count on the device"""
    indptr, indices = graph
    d_indptr, d_indices = dev(indptr), dev(indices)
    inp = _seeds(n, len(indptr) - 1, seed=n)
    want = oracle.extract_neighbour(indptr, indices, inp)
    out, d_num = hip.extract_neighbour(d_indptr, d_indices, dev(inp) if n else torch.empty(0, dtype=torch.int32,
                                                                                         device="cuda"),
                                       max(len(want), 1))
    assert int(d_num.item()) == len(want)
    np.testing.assert_array_equal(host_u32(out, len(want)), want)
    if n >= 256:
        # count on the device (capacity > count), output clipped at out_cap, count-only call
        cap_inp = np.concatenate([inp, np.zeros(500, dtype=np.uint32)])
        d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
        clip = len(want) // 2
        out, d_num = hip.extract_neighbour(d_indptr, d_indices, dev(cap_inp), clip, num_input=0, d_num_input=d_n)
        assert int(d_num.item()) == len(want)
        np.testing.assert_array_equal(host_u32(out, clip), want[:clip])
        out, d_num = hip.extract_neighbour(d_indptr, d_indices, dev(inp), 0)
        assert int(d_num.item()) == len(want)


@pytest.mark.gpu
def test_extract_neighbour_long_and_empty_rows(hip, oracle):
    """a row far longer than a tile's worth of lanes between empty rows"""
    deg = np.zeros(600, dtype=np.uint32)
    deg[[3, 4, 300, 599]] = [70000, 1, 513, 2]
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.uint32)
    indices = np.random.default_rng(3).integers(0, 600, int(indptr[-1]), dtype=np.uint32)
    inp = np.array([0, 3, 1, 2, 599, 4, 5, 300, 3] + list(range(6, 290)), dtype=np.uint32)
    want = oracle.extract_neighbour(indptr, indices, inp)
    out, d_num = hip.extract_neighbour(dev(indptr), dev(indices), dev(inp), len(want))
    assert int(d_num.item()) == len(want)
    np.testing.assert_array_equal(host_u32(out, len(want)), want)


@pytest.mark.gpu
@pytest.mark.parametrize("layers,batch", [(1, 40), (2, 300), (3, 7)])
def test_neighbourhood_expand_matches_oracle(hip, oracle, graph, layers, batch):
    """DoGPUSampleAllNeighbour (cuda_loops.cc:500-571) as level-wise expansion over a stamp array: same closed
    neighbourhoods, same access counts over several batches"""
    indptr, indices = graph
    n = len(indptr) - 1
    d_indptr, d_indices = dev(indptr), dev(indices)
    stamp = torch.zeros(n, dtype=torch.int32, device="cuda")
    freq = torch.zeros(n, dtype=torch.int32, device="cuda")
    want_freq = np.zeros(n, dtype=np.uint32)
    fronts = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(2)]
    for b in range(4):
        seeds = _seeds(batch, n, seed=100 + b)
        want = oracle.sample_all_neighbour(indptr, indices, seeds, layers)
        want_freq[want] += 1
        counts = torch.zeros(layers + 1, dtype=torch.int32, device="cuda")
        seen = [seeds]
        for l in range(layers):
            hip.neighbourhood_expand(d_indptr, d_indices, dev(seeds) if l == 0 else fronts[(l - 1) & 1], stamp, b + 1,
                                     freq, fronts[l & 1], counts[l + 1:l + 2], mark_frontier=(l == 0),
                                     num_frontier=batch if l == 0 else 0,
                                     d_num_frontier=None if l == 0 else counts[l:l + 1])
            seen.append(host_u32(fronts[l & 1], int(counts[l + 1].item())).copy())
        got = np.concatenate(seen)
        assert len(got) == len(want) and len(np.unique(got)) == len(got)
        np.testing.assert_array_equal(np.sort(got), np.sort(want))
        # level structure: level l holds exactly the nodes first reached after l hops
        assert set(seen[0].tolist()) == set(want[:batch].tolist())
    np.testing.assert_array_equal(host_u32(freq), want_freq)
    # stamps: every node ever reached carries the mark of the last batch that reached it
    assert int((host_u32(stamp) != 0).sum()) == int((want_freq != 0).sum())


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["khop0", "khop2"])
def test_batch_driver_resolving_insert_across_generation_wrap(hip, oracle, graph, kind):
    """The last fill of a batch goes through the resolving insert (outcomes + take-over notes instead of bucket
    reads).  A dense little graph makes most edges duplicates (long take-over chains), and capacities of 2^23..2^24 edges
    per layer leave the bucket value 7 generation bits, so a slot's table AND its notes are wiped after 127 batches of
    that slot (the driver rotates over 4 slots: 508 batches in all): every batch before, at and after the wraps must
    match the oracle edge for edge."""
    indptr, indices = graph                     # 3000 nodes, 60000 edges
    num_node = len(indptr) - 1
    fanouts, max_batch, nseed = [45, 50], 4000, 40
    d_indices = dev(indices.copy())
    st, ost = (hip.KHOP0, oracle.KHOP0) if kind == "khop0" else (hip.KHOP2, oracle.KHOP2)
    sampler = hip.Sampler(dev(indptr), d_indices, fanouts, max_batch, sample_type=st, seed=SEED)
    assert (1 << 23) < sampler.max_edges(0) <= (1 << 24)
    bt = sampler.new_batch(0, hip.F32, hip.I64)
    o_indices = indices.copy()
    oht = oracle.HashTable(num_node, sampler.max_nodes)
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    dup_edges = 0
    for b in range(530):
        seeds = _seeds(nseed, num_node, seed=900 + b)
        sampler.sample(dev(seeds), b, bt)
        bt.finish()
        m = bt.wait()
        want = oracle.do_sample(indptr, o_indices, seeds, fanouts, ost, rng, b, oht)
        assert m.overflow == 0
        for li in range(2):
            row, col, nsrc, ndst = bt.graph(li)
            g = want["graphs"][li]
            assert (len(row), nsrc, ndst) == (g["num_edge"], g["num_src"], g["num_dst"]), (b, li)
            np.testing.assert_array_equal(host_u32(row), g["row"], err_msg="batch %d layer %d" % (b, li))
            np.testing.assert_array_equal(host_u32(col), g["col"], err_msg="batch %d layer %d" % (b, li))
        np.testing.assert_array_equal(host_u32(bt.input_nodes()), want["input_nodes"])
        g0 = want["graphs"][0]
        dup_edges += g0["num_edge"] - (g0["num_src"] - g0["num_dst"])
    assert dup_edges > 530 * 2000  # duplicates inside the last fill: the take-over notes were exercised
    np.testing.assert_array_equal(host_u32(d_indices), o_indices)


@pytest.mark.gpu
def test_batch_driver_hub_nodes_in_every_row(hip, oracle):
    """three hub nodes are neighbours of EVERY node: thousands of duplicates of one key inside a single fill, whatever
    order the hardware inserts them in (the resolving insert's take-over notes may form long chains)"""
    num_node, deg = 6000, 12
    rng_np = np.random.default_rng(31)
    rows = np.concatenate([np.tile(np.array([5, 77, 4242], dtype=np.uint32), (num_node, 1)),
                           rng_np.integers(0, num_node, (num_node, deg - 3), dtype=np.uint32)], axis=1)
    for r in rows:
        rng_np.shuffle(r)  # hubs at random positions of the row
    indptr = (np.arange(num_node + 1, dtype=np.uint32) * deg).astype(np.uint32)
    indices = rows.reshape(-1).copy()
    fanouts, batch = [12, 12], 1500
    for kind, st, ost in (("khop0", hip.KHOP0, oracle.KHOP0), ("khop2", hip.KHOP2, oracle.KHOP2)):
        d_indices = dev(indices.copy())
        o_indices = indices.copy()
        sampler = hip.Sampler(dev(indptr), d_indices, fanouts, batch, sample_type=st, seed=SEED)
        bt = sampler.new_batch(0, hip.F32, hip.I64)
        oht = oracle.HashTable(num_node, sampler.max_nodes)
        rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
        for b in range(6):
            seeds = _seeds(batch, num_node, seed=40 + b)
            sampler.sample(dev(seeds), b, bt)
            bt.finish()
            m = bt.wait()
            want = oracle.do_sample(indptr, o_indices, seeds, fanouts, ost, rng, b, oht)
            assert m.overflow == 0
            for li in range(2):
                row, col, nsrc, ndst = bt.graph(li)
                g = want["graphs"][li]
                assert (len(row), nsrc, ndst) == (g["num_edge"], g["num_src"], g["num_dst"]), (kind, b, li)
                np.testing.assert_array_equal(host_u32(row), g["row"], err_msg="%s batch %d layer %d" % (kind, b, li))
                np.testing.assert_array_equal(host_u32(col), g["col"])
            np.testing.assert_array_equal(host_u32(bt.input_nodes()), want["input_nodes"])


@pytest.mark.gpu
def test_batch_driver_random_configurations(hip, oracle):
    """Differential test over 120 seeded random configurations: sampler type, 1-3 layers, fanouts 1..30 (below and above
    the rows' lengths), graphs from 60 to 20 000 nodes from sparse to dense (few to almost-all duplicates), batch sizes
    1..700 with a short last batch -- three batches each, every block, node list and (khop2) the mutated CSR bit-exact."""
    from fgnn_hip import synth
    rs = np.random.default_rng(20240611)
    kinds = ["khop0", "khop2", "khop1", "weighted_khop", "weighted_khop_prefix", "weighted_khop_hash_dedup"]
    for case in range(120):
        kind = kinds[case % len(kinds)]
        num_node = int(rs.choice([60, 500, 4000, 20000]))
        avg_deg = float(rs.choice([1.5, 6, 25, 60]))
        num_edge = max(num_node, int(num_node * avg_deg))
        L = int(rs.integers(1, 4))
        fanouts = [int(rs.integers(1, 31)) for _ in range(L)]
        batch = int(min(num_node, rs.choice([1, 7, 64, 300, 700])))
        indptr, indices = synth.powerlaw_csr(num_node, num_edge, seed=1000 + case)
        what = "case %d: %s fanouts %s batch %d nodes %d edges %d" % (case, kind, fanouts, batch, num_node, len(indices))
        kw, okw = {}, {}
        if kind == "weighted_khop_prefix":
            prefix = synth.prob_prefix_table(indptr, indices)
            kw, okw = dict(prob_prefix=dev(prefix)), dict(prob_prefix=prefix)
        elif kind in ("weighted_khop", "weighted_khop_hash_dedup"):
            prob, alias = synth.alias_tables(indptr, indices)
            kw, okw = dict(prob_table=dev(prob), alias_table=dev(alias)), dict(prob_prefix=prob, alias_table=alias)
        hst = dict(khop0=hip.KHOP0, khop2=hip.KHOP2, khop1=hip.KHOP1, weighted_khop=hip.WEIGHTED_KHOP,
                   weighted_khop_prefix=hip.WEIGHTED_KHOP_PREFIX, weighted_khop_hash_dedup=hip.WEIGHTED_KHOP_HASH_DEDUP)[kind]
        ost = dict(khop0=oracle.KHOP0, khop2=oracle.KHOP2, khop1=oracle.KHOP1, weighted_khop=oracle.WEIGHTED_KHOP,
                   weighted_khop_prefix=oracle.WEIGHTED_KHOP_PREFIX,
                   weighted_khop_hash_dedup=oracle.WEIGHTED_KHOP_HASH_DEDUP)[kind]
        d_indices = dev(indices.copy())
        o_indices = indices.copy()
        sampler = hip.Sampler(dev(indptr), d_indices, fanouts, batch, sample_type=hst, seed=SEED + case, **kw)
        bt = sampler.new_batch()
        oht = oracle.HashTable(num_node, sampler.max_nodes)
        rng = oracle.make_rng(oracle.RNG_PHILOX, SEED + case)
        for b, n in enumerate([batch, batch, max(1, batch // 2)]):
            seeds = _seeds(n, num_node, seed=5000 + 10 * case + b)
            sampler.sample(dev(seeds), 77 + b, bt)
            bt.finish()
            m = bt.wait()
            want = oracle.do_sample(indptr, o_indices, seeds, fanouts, ost, rng, 77 + b, oht, **okw)
            assert m.overflow == 0 and m.num_layers == L, what
            for li in range(L):
                row, col, nsrc, ndst = bt.graph(li)
                g = want["graphs"][li]
                assert (len(row), nsrc, ndst) == (g["num_edge"], g["num_src"], g["num_dst"]), (what, b, li)
                np.testing.assert_array_equal(host_u32(row), g["row"], err_msg="%s batch %d layer %d" % (what, b, li))
                np.testing.assert_array_equal(host_u32(col), g["col"], err_msg="%s batch %d layer %d" % (what, b, li))
            np.testing.assert_array_equal(host_u32(bt.input_nodes()), want["input_nodes"], err_msg=what)
        np.testing.assert_array_equal(host_u32(d_indices), o_indices, err_msg=what)
        del sampler, bt


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["khop2", "khop0", "khop1"])
def test_batch_driver_empty_and_isolated_batches(hip, oracle, kind):
    """a batch without seeds, a batch whose seeds have no neighbours at all, then ordinary batches: the summaries are
    what the reference's loop would report (no edges, input nodes = seeds) and the chain of later batches is intact"""
    from fgnn_hip import synth
    num_node = 5000
    indptr, indices = synth.powerlaw_csr(num_node, 40000, seed=77)
    deg = np.diff(indptr.astype(np.int64))
    isolated = np.flatnonzero(deg == 0).astype(np.uint32)
    assert len(isolated) >= 20
    fanouts, batch = [6, 4], 300
    hst, ost = dict(khop2=(hip.KHOP2, oracle.KHOP2), khop0=(hip.KHOP0, oracle.KHOP0), khop1=(hip.KHOP1, oracle.KHOP1))[kind]
    d_indices = dev(indices.copy())
    o_indices = indices.copy()
    sampler = hip.Sampler(dev(indptr), d_indices, fanouts, batch, sample_type=hst, seed=SEED)
    bt = sampler.new_batch()
    oht = oracle.HashTable(num_node, sampler.max_nodes)
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    plans = [np.empty(0, dtype=np.uint32), isolated[:20], _seeds(batch, num_node, seed=1), np.empty(0, dtype=np.uint32),
             _seeds(41, num_node, seed=2)]
    for b, seeds in enumerate(plans):
        d_seeds = dev(seeds) if len(seeds) else torch.empty(0, dtype=torch.int32, device="cuda")
        sampler.sample(d_seeds, 300 + b, bt)
        bt.finish()
        m = bt.wait()
        assert m.overflow == 0 and m.num_output == len(seeds), (kind, b)
        if len(seeds) == 0:
            assert m.num_input == 0 and all(int(m.num_edge[l]) == 0 for l in range(2)), (kind, b)
            continue
        want = oracle.do_sample(indptr, o_indices, seeds, fanouts, ost, rng, 300 + b, oht)
        for li in range(2):
            row, col, nsrc, ndst = bt.graph(li)
            g = want["graphs"][li]
            assert (len(row), nsrc, ndst) == (g["num_edge"], g["num_src"], g["num_dst"]), (kind, b, li)
            np.testing.assert_array_equal(host_u32(row), g["row"])
            np.testing.assert_array_equal(host_u32(col), g["col"])
        np.testing.assert_array_equal(host_u32(bt.input_nodes()), want["input_nodes"])
        if b == 1:
            assert want["total_edges"] == 0 and len(want["input_nodes"]) == 20
    np.testing.assert_array_equal(host_u32(d_indices), o_indices)


@pytest.mark.gpu
def test_presample_rank_and_cache_table_match_oracle(hip, oracle):
    """fgnn_presample_count / fgnn_presample_rank / fgnn_cache_table_build (dist/pre_sampler.cc:75-162,
    dist_engine.cc:193-229) against the oracle: frequencies counted with device-side sizes, rank = (frequency desc,
    id desc), table[rank[i]] = i for the cached head."""
    num_node = 50000
    rs = np.random.default_rng(5)
    freq = torch.zeros(num_node, dtype=torch.int32, device="cuda")
    want = np.zeros(num_node, dtype=np.uint32)
    for k in range(6):
        cap = 9000
        n = int(rs.integers(0, cap + 1)) if k else cap
        nodes = (rs.zipf(1.3, size=cap) % num_node).astype(np.uint32)
        d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
        if k % 2:
            hip.presample_count(freq, dev(nodes), d_num_nodes=d_n)   # device-side count, capacity = len(nodes)
        else:
            hip.presample_count(freq, dev(nodes), num_nodes=n)
        np.add.at(want, nodes[:n], 1)
    np.testing.assert_array_equal(host_u32(freq), want)
    rank = hip.presample_rank(freq)
    o_rank = oracle.presample_rank(want)
    np.testing.assert_array_equal(host_u32(rank), o_rank)
    for n_cached in (0, 1, num_node // 5, num_node):
        table = hip.cache_table_build(rank, n_cached)
        np.testing.assert_array_equal(host_u32(table), oracle.cache_table_build(o_rank, n_cached, num_node))


@pytest.mark.gpu
def test_run_batch_cached_matches_oracle(hip, oracle):
    """fgnn_sampler_run_batch_cached: sample + cache split + CombineMissData (rows fetched from PINNED HOST memory by
    the gather kernel) + CombineCacheData in one call, with a masked 2^k-row feature table (SAMGRAPH_EMPTY_FEAT)."""
    from fgnn_hip import synth
    num_node, dim, batch, fanouts, bits = 30000, 128, 1500, [10, 5], 13
    mask = (1 << bits) - 1
    indptr, indices = synth.powerlaw_csr(num_node, 500000, seed=77)
    feat = synth.node_features(1 << bits, dim)
    label = np.random.default_rng(3).integers(0, 40, size=num_node).astype(np.int64)
    rank = np.random.default_rng(4).permutation(num_node).astype(np.uint32)
    n_cached = num_node // 4
    table = oracle.cache_table_build(rank, n_cached, num_node)
    cache_rows = feat[rank[:n_cached] & mask]
    d_indices = dev(indices.copy())
    sampler = hip.Sampler(dev(indptr), d_indices, fanouts, batch, sample_type=hip.KHOP2, seed=SEED)
    bt = sampler.new_batch(dim, hip.F32, hip.I64)
    bt.enable_timing(True)
    hip.load().fgnn_batch_set_feat_row_mask(bt.h, mask)
    host_feat = torch.from_numpy(feat).pin_memory()
    d_table, d_cache, d_label = dev(table), dev(cache_rows), dev(label)
    o_indices = indices.copy()
    oht = oracle.HashTable(num_node, sampler.max_nodes)
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    for b in range(3):
        seeds = _seeds(batch if b < 2 else 17, num_node, seed=300 + b)
        sampler.run_batch_cached(b, dev(seeds), 50 + b, bt, d_table, d_cache, host_feat, d_label)
        m = bt.wait()
        want = oracle.do_sample(indptr, o_indices, seeds, fanouts, oracle.KHOP2, rng, 50 + b, oht)
        nodes = host_u32(bt.input_nodes())
        np.testing.assert_array_equal(nodes, want["input_nodes"])
        for got, w in zip(bt.cache_index_arrays(), oracle.get_miss_cache_index(table, nodes)):
            np.testing.assert_array_equal(host_u32(got), w)
        assert m.num_miss + m.num_cache == m.num_input and m.num_miss > 0 and m.num_cache > 0
        assert bt.feat().cpu().numpy().tobytes() == feat[nodes & mask].tobytes()
        np.testing.assert_array_equal(bt.label().cpu().numpy(), label[seeds])
        ms = bt.extract_cached_ms()
        assert ms[0] >= 0 and ms[1] >= 0
        np.testing.assert_array_equal(host_u32(bt.d_num_input()), [m.num_input])


@pytest.mark.gpu
def test_sanity_check_kernel(hip):
    """GPUSanityCheckList + GPUBatchSanityCheck (cuda_sanity_check.cu:28-88) as one launch reporting through a flag
    word: clean batches pass, an id handed out twice in an epoch (in another batch or inside one batch), the invalid
    value and an id beyond the graph are each reported; a new epoch may hand every id out again."""
    num_node = 100_000
    rs = np.random.default_rng(3)
    perm = rs.permutation(num_node).astype(np.uint32)
    chk = hip.SanityChecker(num_node, torch.device("cuda", 0))
    for b in range(10):
        assert chk.check(dev(perm[b * 8000:(b + 1) * 8000])) == 0
    again = perm[80000:88000].copy()
    again[4321] = perm[17]                      # handed out in batch 0
    assert chk.check(dev(again)) == 2
    twice = perm[88000:96000].copy()
    twice[7000] = twice[12]                     # twice inside one batch
    assert chk.check(dev(twice)) == 2
    bad = perm[96000:99000].copy()
    bad[5] = 0xFFFFFFFF
    assert chk.check(dev(bad)) == 1
    assert chk.check(dev(np.array([num_node + 5], dtype=np.uint32))) == 4
    assert chk.check(dev(np.array([0xFFFFFFFF, perm[0]], dtype=np.uint32))) == 3
    assert chk.check(dev(perm[:100]), no_duplicates=False) == 0  # list check only: repeats are not its business
    chk.new_epoch()
    for b in range(12):
        assert chk.check(dev(perm[b * 8000:(b + 1) * 8000])) == 0
    assert chk.check(dev(np.zeros(0, dtype=np.uint32))) == 0
