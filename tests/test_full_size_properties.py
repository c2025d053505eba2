"""Parity at BASELINE.json's full size (papers100M shape: N=111 059 956, E=1 615 685 872, D=128, batch 8000,
fanout [25,10]) through size-independent properties -- the oracle cannot hold a 57 GB feature table in a test, so the
full-size run is checked by invariants that any correct run of the reference satisfies (tier T3),
plus exact checks that only need the outputs themselves."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def world():
    sys.path.insert(0, ROOT)
    import bench
    from fgnn_hip import lib
    lib.load()
    dev = torch.device("cuda:0")
    w = bench.WORKLOADS["papers100M"]
    indptr, indices, ne = bench.gen_graph_on_gpu(w["num_node"], w["num_edge"], 42, dev)
    feat = bench.gen_features_on_gpu(w["num_node"], w["feat_dim"], dev)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    label = torch.randint(0, w["num_class"], (w["num_node"],), generator=g, device=dev, dtype=torch.int64)
    train = torch.randperm(w["num_node"], generator=g, device=dev)[:40000].to(torch.int32)
    table = torch.full((w["num_node"],), -1, dtype=torch.int32, device=dev)
    cached = torch.randperm(w["num_node"], generator=g, device=dev)[:w["num_node"] // 5]
    table[cached] = torch.arange(cached.numel(), device=dev, dtype=torch.int32)
    return dict(lib=lib, w=w, indptr=indptr, indices=indices, feat=feat, label=label, train=train, table=table)


def _run(world, sample_type, nbatch=3):
    lib, w = world["lib"], world["w"]
    indices = world["indices"].clone() if sample_type == lib.KHOP2 else world["indices"]
    sampler = lib.Sampler(world["indptr"], indices, w["fanout"], w["batch_size"], sample_type=sample_type, seed=11)
    out = []
    for b in range(nbatch):
        bt = sampler.new_batch(w["feat_dim"], lib.F32, lib.I64)
        seeds = world["train"][b * 8000:(b + 1) * 8000]
        sampler.run_batch(b, seeds, 100 + b, bt, world["table"], world["feat"], world["label"])
        m = bt.wait()
        out.append((bt, m, seeds))
    return sampler, indices, out


@pytest.mark.parametrize("kind", ["khop2", "khop0"])
def test_full_size_invariants(world, kind):
    lib, w = world["lib"], world["w"]
    st = lib.KHOP2 if kind == "khop2" else lib.KHOP0
    sampler, indices, out = _run(world, st)
    ip = world["indptr"].long() & 0xFFFFFFFF
    for bt, m, seeds in out:
        assert m.overflow == 0 and m.num_layers == 2 and m.num_output == 8000
        nodes = bt.input_nodes().long() & 0xFFFFFFFF
        U = nodes.numel()
        # dedup: the unique list is duplicate-free and starts with the batch seeds (FillWithUnique)
        assert torch.unique(nodes).numel() == U
        assert torch.equal(nodes[:8000], seeds.long() & 0xFFFFFFFF)
        prev_src = None
        for l in (1, 0):  # sampled from the last fanout to the first (cuda_loops.cc:87)
            row, col, nsrc, ndst = bt.graph(l)
            row, col = row.long(), col.long()
            F = w["fanout"][l]
            assert nsrc >= ndst and int(row.max()) < nsrc and int(col.max()) < ndst
            assert ndst == (8000 if l == 1 else prev_src)          # next layer's seeds = everything seen so far
            # seed-major order, and every seed emits exactly min(degree, fanout) edges
            assert bool((col[1:] >= col[:-1]).all())
            deg = (ip[nodes[:ndst] + 1] - ip[nodes[:ndst]])
            want = torch.clamp(deg, max=F)
            got = torch.bincount(col, minlength=ndst)
            assert torch.equal(got, want)
            assert int(m.num_edge[l]) == int(want.sum())
            # every emitted edge is an edge of the CSR row of its seed (checked exactly on a sample of edges)
            pick = torch.randint(0, row.numel(), (2000,), device=row.device)
            s_ids, n_ids = nodes[col[pick]].cpu().numpy(), nodes[row[pick]].cpu().numpy()
            for s, nb in zip(s_ids[:300], n_ids[:300]):
                a, b = int(ip[s]), int(ip[s + 1])
                assert (indices[a:b].long() & 0xFFFFFFFF == int(nb)).any()
            # without replacement: a seed never emits the same CSR position twice => for rows without repeated
            # neighbour ids the emitted neighbours of a seed are distinct; checked on short rows (emit everything)
            prev_src = nsrc
        assert U == prev_src == int(m.num_input)
        # cache split: stable partition, sizes add up, hits point at the right slots
        ms, md, cs, cd = [t.long() & 0xFFFFFFFF for t in bt.cache_index_arrays()]
        assert int(m.num_miss) + int(m.num_cache) == U
        assert bool((md[1:] > md[:-1]).all()) and bool((cd[1:] > cd[:-1]).all())
        tab = world["table"].long()
        assert torch.equal(nodes[md], ms) and bool((tab[ms] == -1).all())
        assert torch.equal(tab[nodes[cd]] & 0xFFFFFFFF, cs)
        # gather: bit-exact against torch indexing of the same table
        assert torch.equal(bt.feat(), world["feat"][nodes])
        assert torch.equal(bt.label(), world["label"][seeds.long()])
    if kind == "khop2":
        # in-place Fisher-Yates only permutes rows: every row of the mutated CSR is a permutation of the original
        touched = torch.cat([bt.input_nodes().long() & 0xFFFFFFFF for bt, _, _ in out])[:5000].cpu().numpy()
        changed = 0
        for s in touched[:400]:
            a, b = int(ip[s]), int(ip[s + 1])
            o, n = world["indices"][a:b], indices[a:b]
            assert torch.equal(torch.sort(o).values, torch.sort(n).values)
            changed += int(not torch.equal(o, n))
        assert changed > 0
    else:
        assert torch.equal(indices, world["indices"])  # khop0 never writes the CSR


def test_full_size_determinism(world):
    """Same seed, same batches -> identical outputs (counter-based RNG; khop0 keeps the CSR intact)."""
    lib = world["lib"]
    _, _, a = _run(world, lib.KHOP0, nbatch=2)
    _, _, b = _run(world, lib.KHOP0, nbatch=2)
    for (b1, m1, _), (b2, m2, _) in zip(a, b):
        assert torch.equal(b1.input_nodes(), b2.input_nodes())
        for l in (0, 1):
            assert torch.equal(b1.graph(l)[0], b2.graph(l)[0]) and torch.equal(b1.graph(l)[1], b2.graph(l)[1])


def test_full_size_cached_extraction_invariants(world):
    """BASELINE config 3's trainer side at its full size: 0.2 of papers100M's rows in an HBM cache (22.2 M rows, 11.4 GB),
    every other row pulled out of a pinned HOST table by the same launch (2^24 rows, ids masked: SAMGRAPH_EMPTY_FEAT /
    the reference's CPUMockExtract), three batches of [25, 10] x 8000.  Checked with what any correct run satisfies:
    hit rows = the cache's rows = the feature table's rows, miss rows = the host table's rows of the masked ids, every
    output row written exactly once (CombineMissData + CombineCacheData, cuda_cache_manager_device.cu:165-210), labels
    of the seeds; and the one-launch path against the two gathers of the plain C ABI on the same lists."""
    lib, w = world["lib"], world["w"]
    dev = world["feat"].device
    bits = 24
    mask = (1 << bits) - 1
    host = torch.empty((1 << bits, w["feat_dim"]), dtype=torch.float32).pin_memory()
    host.copy_(world["feat"][:1 << bits])
    tab = world["table"].long()
    n_cached = int((tab >= 0).sum())
    slot_node = torch.empty(n_cached, dtype=torch.int64, device=dev)
    cached_nodes = torch.nonzero(tab >= 0).flatten()
    slot_node[tab[cached_nodes]] = cached_nodes
    cache_rows = torch.empty((n_cached, w["feat_dim"]), dtype=torch.float32, device=dev)
    lib.gather_rows(cache_rows, world["feat"], src_index=slot_node.to(torch.int32))
    del slot_node, cached_nodes
    sampler = lib.Sampler(world["indptr"], world["indices"], w["fanout"], w["batch_size"], sample_type=lib.KHOP0, seed=11)
    for b in range(3):
        bt = sampler.new_batch(w["feat_dim"], lib.F32, lib.I64)
        lib.load().fgnn_batch_set_feat_row_mask(bt.h, mask)
        seeds = world["train"][b * 8000:(b + 1) * 8000]
        bt.feat_buffer().fill_(float("nan"))  # a row nobody writes stays NaN
        sampler.run_batch_cached(b, seeds, 100 + b, bt, world["table"], cache_rows, host, world["label"])
        m = bt.wait()
        assert m.overflow == 0
        nodes = bt.input_nodes().long() & 0xFFFFFFFF
        U = nodes.numel()
        assert int(m.num_miss) + int(m.num_cache) == U and int(m.num_miss) > 0 and int(m.num_cache) > 0
        hit = tab[nodes] >= 0
        assert int(hit.sum()) == int(m.num_cache)
        feat = bt.feat()
        assert not bool(torch.isnan(feat).any())
        want = torch.where(hit[:, None], world["feat"][nodes], world["feat"][nodes & mask])
        assert torch.equal(feat, want)
        assert torch.equal(bt.label(), world["label"][seeds.long()])
        # the same lists through the two-call C ABI (fgnn_gather_rows_masked from the host table, fgnn_gather_rows
        # from the cache) into a second buffer
        ms, md, cs, cd = bt.cache_index_arrays()
        two = torch.full_like(feat, float("nan"))
        lib.gather_rows(two, host, src_index=ms, dst_index=md, src_row_mask=mask)
        lib.gather_rows(two, cache_rows, src_index=cs, dst_index=cd)
        assert torch.equal(two, feat)


@pytest.mark.gpu
def test_gather_larger_than_one_launch_is_sliced():
    from fgnn_hip import lib as hip
    """A host-sized gather of more than 2^32 16-byte chunks (a whole-table feature cache: 66 GiB of rows here) is cut
    into several launches; rows are checked at both ends and around the slice boundary."""
    dim = 256  # 64 chunks per row -> 2^25 rows per slice
    n = (1 << 26) + 12345
    src_rows = 1 << 10
    src = torch.arange(src_rows * dim, dtype=torch.float32, device="cuda").reshape(src_rows, dim)
    idx = (torch.arange(n, dtype=torch.int64, device="cuda") * 7919 % src_rows).to(torch.int32)
    out = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    hip.gather_rows(out, src, src_index=idx)
    for r in (0, 1, (1 << 25) - 1, 1 << 25, (1 << 25) + 1, (1 << 26) - 1, 1 << 26, n - 1):
        assert torch.equal(out[r], src[int(idx[r])]), r


@pytest.mark.gpu
def test_full_size_neighbourhood_kernels(world):
    """fgnn_extract_neighbour / fgnn_neighbourhood_expand on the papers100M-shaped CSR (1.6 G edges, ids beyond 2^31
    bytes of offset): properties that need no oracle -- the emitted list IS the concatenation of the rows, the count is
    the degree sum; the level-wise expansion reaches exactly the set of distinct neighbours, stamps each node once,
    counts each reached node once per batch and is idempotent within a batch."""
    lib, w = world["lib"], world["w"]
    indptr, indices = world["indptr"], world["indices"]
    ip = indptr.long() & 0xFFFFFFFF
    n = w["num_node"]
    seeds = world["train"][:8000]
    sl = seeds.long() & 0xFFFFFFFF
    deg = ip[sl + 1] - ip[sl]
    total = int(deg.sum())
    out, d_num = lib.extract_neighbour(indptr, indices, seeds, total)
    assert int(d_num.item()) == total
    # row by row: the segment of seed i starts at the exclusive degree prefix and equals indices[indptr[v] : indptr[v+1]]
    start = torch.cumsum(deg, 0) - deg
    pos = torch.arange(total, device=out.device)
    owner = torch.searchsorted(start + deg, pos, right=True)
    src = ip[sl][owner] + (pos - start[owner])
    assert torch.equal(out[:total], indices[src])
    # two levels of the closed neighbourhood from the same seeds
    stamp = torch.zeros(n, dtype=torch.int32, device=out.device)
    freq = torch.zeros(n, dtype=torch.int32, device=out.device)
    fronts = [torch.empty(n, dtype=torch.int32, device=out.device) for _ in range(2)]
    counts = torch.zeros(3, dtype=torch.int32, device=out.device)
    lib.neighbourhood_expand(indptr, indices, seeds, stamp, 7, freq, fronts[0], counts[1:2], mark_frontier=True)
    lib.neighbourhood_expand(indptr, indices, fronts[0], stamp, 7, freq, fronts[1], counts[2:3], num_frontier=0,
                             d_num_frontier=counts[1:2])
    c1, c2 = int(counts[1].item()), int(counts[2].item())
    level1 = fronts[0][:c1].long() & 0xFFFFFFFF
    want1 = torch.unique(out[:total].long() & 0xFFFFFFFF)
    want1 = want1[~torch.isin(want1, sl)]
    assert c1 == want1.numel() and torch.equal(torch.sort(level1).values, want1)
    level2 = fronts[1][:c2].long() & 0xFFFFFFFF
    assert torch.unique(level2).numel() == c2 and not torch.isin(level2, level1).any() and not torch.isin(level2, sl).any()
    reached = 8000 + c1 + c2
    assert int((stamp == 7).sum()) == reached and int(freq.sum()) == reached and int(freq.max()) == 1
    # every level-2 node is a neighbour of some level-1 node: check a sample of them through the reverse direction
    l1_deg = ip[level1 + 1] - ip[level1]
    assert c2 <= int(l1_deg.sum())
    # idempotence: the same batch mark again adds nothing
    counts.zero_()
    lib.neighbourhood_expand(indptr, indices, seeds, stamp, 7, freq, fronts[1], counts[1:2], mark_frontier=True)
    assert int(counts[1].item()) == 0 and int(freq.sum()) == reached
    # a new batch mark reaches the same level-1 set again and counts it a second time
    lib.neighbourhood_expand(indptr, indices, seeds, stamp, 8, freq, fronts[1], counts[1:2], mark_frontier=True)
    assert int(counts[1].item()) == c1 and int(freq.sum()) == reached + 8000 + c1 and int(freq.max()) == 2


def _shape_world(name, with_prefix):
    sys.path.insert(0, ROOT)
    import bench
    from fgnn_hip import lib
    lib.load()
    dev = torch.device("cuda:0")
    w = bench.WORKLOADS[name]
    indptr, indices, ne = bench.gen_graph_on_gpu(w["num_node"], w["num_edge"], 42, dev)
    prefix = bench.gen_prefix_on_gpu(indptr, ne, 11, dev) if with_prefix else None
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    train = torch.randperm(w["num_node"], generator=g, device=dev)[:24000].to(torch.int32)
    return lib, w, indptr, indices, prefix, train


def _edge_in_row(ip, indices, nodes, row, col, samples=300):
    pick = torch.randint(0, row.numel(), (samples,), device=row.device)
    s_ids, n_ids = nodes[col[pick]].cpu().numpy(), nodes[row[pick]].cpu().numpy()
    for s, nb in zip(s_ids, n_ids):
        a, b = int(ip[s]), int(ip[s + 1])
        assert (indices[a:b].long() & 0xFFFFFFFF == int(nb)).any()


def test_full_size_weighted_gcn_invariants():
    """BASELINE config 4's sampler side at full size (twitter shape: N=41 652 230, E=1 468 365 182, GCN fanout [5,10,15],
    weighted_khop_prefix, batch 8000): the properties any run of GPUSampleWeightedKHopPrefix + the dedup satisfies
    (SURVEY.md 8(c) T3): at most `fanout` edges per seed, output ordered by seed id, no adjacent duplicate pairs, every
    edge in the CSR row of its seed, layer sizes chained."""
    lib, w, indptr, indices, prefix, train = _shape_world("twitter", True)
    sampler = lib.Sampler(indptr, indices, w["fanout"], w["batch_size"], sample_type=lib.WEIGHTED_KHOP_PREFIX, seed=13,
                          prob_prefix=prefix)
    ip = indptr.long() & 0xFFFFFFFF
    for b in range(2):
        bt = sampler.new_batch()
        seeds = train[b * 8000:(b + 1) * 8000]
        sampler.sample(seeds, 40 + b, bt)
        bt.finish()
        m = bt.wait()
        assert m.overflow == 0 and m.num_layers == 3 and m.num_output == 8000
        nodes = bt.input_nodes().long() & 0xFFFFFFFF
        assert torch.unique(nodes).numel() == nodes.numel() == int(m.num_input)
        assert torch.equal(nodes[:8000], seeds.long() & 0xFFFFFFFF)
        prev_src = None
        for l in (2, 1, 0):
            row, col, nsrc, ndst = bt.graph(l)
            row, col = row.long(), col.long()
            F = w["fanout"][l]
            assert nsrc >= ndst and int(row.max()) < nsrc and int(col.max()) < ndst
            assert ndst == (8000 if l == 2 else prev_src)
            per_seed = torch.bincount(col, minlength=ndst)
            deg = ip[nodes[:ndst] + 1] - ip[nodes[:ndst]]
            assert int(per_seed.max()) <= F and bool((per_seed[deg == 0] == 0).all()) and bool((per_seed[deg > 0] >= 1).all())
            # ordered by the seed's node id (the reference's stable radix sort by src), draws of a seed together
            seed_ids = nodes[col]
            assert bool((seed_ids[1:] >= seed_ids[:-1]).all())
            # only a draw equal to the seed's NEXT draw is dropped: no adjacent duplicate (seed, neighbour) pairs
            assert not bool(((col[1:] == col[:-1]) & (row[1:] == row[:-1])).any())
            _edge_in_row(ip, indices, nodes, row, col)
            prev_src = nsrc
        assert prev_src == int(m.num_input)
    del sampler, prefix, indices
    torch.cuda.empty_cache()


def test_full_size_random_walk_invariants():
    """BASELINE config 5's sampler side at full size (uk-2006-05 shape: N=77 741 046, E=2 965 197 340, PinSAGE walks
    25 x 3, restart 0.5, top-5, 3 layers): per seed at most K edges with non-increasing visit counts that sum to at most
    walks x length, distinct destinations, seeds in input order, layer sizes chained."""
    lib, w, indptr, indices, _, train = _shape_world("uk-2006-05", False)
    K, W, Lw = w["fanout"][0], w["num_walks"], w["walk_len"]
    sampler = lib.Sampler(indptr, indices, w["fanout"], w["batch_size"], sample_type=lib.RANDOM_WALK, seed=13,
                          walk_len=Lw, num_walks=W, restart_prob=w["restart_prob"])
    for b in range(2):
        bt = sampler.new_batch()
        seeds = train[b * 8000:(b + 1) * 8000]
        sampler.sample(seeds, 50 + b, bt)
        bt.finish()
        m = bt.wait()
        assert m.overflow == 0 and m.num_layers == 3 and m.num_output == 8000
        nodes = bt.input_nodes().long() & 0xFFFFFFFF
        assert torch.unique(nodes).numel() == nodes.numel() == int(m.num_input)
        prev_src = None
        for l in (2, 1, 0):
            row, col, nsrc, ndst = bt.graph(l)
            cnt = bt.graph_data(l).long()
            row, col = row.long(), col.long()
            assert nsrc >= ndst and int(row.max()) < nsrc and int(col.max()) < ndst
            assert ndst == (8000 if l == 2 else prev_src)
            assert bool((col[1:] >= col[:-1]).all())                      # seeds in input order
            per_seed = torch.bincount(col, minlength=ndst)
            assert int(per_seed.max()) <= K
            visits = torch.zeros(ndst, dtype=torch.long, device=col.device).index_add_(0, col, cnt)
            assert int(visits.max()) <= W * Lw and int(cnt.min()) >= 1
            same = col[1:] == col[:-1]
            assert bool((cnt[1:][same] <= cnt[:-1][same]).all())            # top-K order: counts non-increasing
            key = col * (nsrc + 1) + row                                    # destinations of a seed are distinct
            assert torch.unique(key).numel() == key.numel()
            prev_src = nsrc
        assert prev_src == int(m.num_input)
    del sampler, indices
    torch.cuda.empty_cache()


@pytest.mark.parametrize("kind,fanout", [("khop2", [10, 5]), ("khop2", [25, 10]), ("khop0", [10, 5])])
def test_products_shape_bit_exact_and_invariants(oracle, kind, fanout):
    """BASELINE configs 1 and 2 at full size on the GPU box (products shape: N=2 449 029, E~123.7 M, D=100, C=47,
    batch 8000; fanout [10,5] = config 1's sampling work, [25,10] = config 2; datagen/products.py:92-98): this graph is
    small enough to bring to the host, so on top of the size-independent invariants the whole batch -- blocks, unique
    list, cache split, D=100 feature rows (25 sixteen-byte chunks: the gather's odd-width path), labels and khop2's
    mutated CSR -- is compared BIT-EXACTLY with the oracle's restatement of DoGPUSample (cuda_loops.cc:50-267) over
    three consecutive batches (the CSR mutation carries).  khop0: the R-MAT hubs of this graph (rows of 10^4..10^5
    entries in every frontier) go through the hub split of the FUSED sampler -- listed by a scan, drawn by all
    workgroups, winners copied by the owner -- and must still equal the oracle's one-draw-per-element reservoir."""
    sys.path.insert(0, ROOT)
    import bench
    from fgnn_hip import lib
    lib.load()
    dev = torch.device("cuda:0")
    w = bench.WORKLOADS["products"]
    N, D, B = w["num_node"], w["feat_dim"], w["batch_size"]
    assert (N, D, w["num_class"], w["num_train"]) == (2449029, 100, 47, 196615)
    indptr, indices, ne = bench.gen_graph_on_gpu(N, w["num_edge"], 42, dev)
    feat = bench.gen_features_on_gpu(N, D, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    label = torch.randint(0, w["num_class"], (N,), generator=g, device=dev, dtype=torch.int64)
    train = torch.randperm(N, generator=g, device=dev)[:w["num_train"]].to(torch.int32)
    rank = torch.randperm(N, generator=g, device=dev).to(torch.int32)
    table = lib.cache_table_build(rank, N // 5, N)
    h_indptr = indptr.cpu().numpy().view(np.uint32)
    h_indices = indices.cpu().numpy().view(np.uint32).copy()
    h_feat, h_label = feat.cpu().numpy(), label.cpu().numpy()
    h_table = table.cpu().numpy().view(np.uint32)
    np.testing.assert_array_equal(h_table, oracle.cache_table_build(rank.cpu().numpy().view(np.uint32), N // 5, N))
    d_indices = indices.clone()
    st, ost = (lib.KHOP2, oracle.KHOP2) if kind == "khop2" else (lib.KHOP0, oracle.KHOP0)
    sampler = lib.Sampler(indptr, d_indices, fanout, B, sample_type=st, seed=0x5A4D47)
    deg_max = int(((indptr[1:].long() - indptr[:-1].long()) & 0xFFFFFFFF).max())
    assert deg_max > 16384  # the graph does have rows beyond the split threshold
    rng = oracle.make_rng(oracle.RNG_PHILOX, 0x5A4D47)
    oht = oracle.HashTable(N, oracle.predict_num_nodes(B, fanout))
    ip = indptr.long() & 0xFFFFFFFF
    # the last batch of an epoch is short (196 615 = 24 * 8000 + 4615): take it as one of the three
    spans = [(0, B), (B, 2 * B), (24 * B, w["num_train"])]
    for b, (lo, hi) in enumerate(spans):
        seeds = train[lo:hi]
        bt = sampler.new_batch(D, lib.F32, lib.I64)
        sampler.run_batch(b, seeds, 7 + b, bt, table, feat, label)
        m = bt.wait()
        assert m.overflow == 0 and m.num_output == hi - lo
        h_seeds = seeds.cpu().numpy().view(np.uint32)
        want = oracle.do_sample(h_indptr, h_indices, h_seeds, fanout, ost, rng, 7 + b, oht)
        nodes = bt.input_nodes().cpu().numpy().view(np.uint32)
        np.testing.assert_array_equal(nodes, want["input_nodes"])
        prev_src = None
        for l in (1, 0):
            row, col, nsrc, ndst = bt.graph(l)
            gr = want["graphs"][l]
            assert (nsrc, ndst, int(m.num_edge[l])) == (gr["num_src"], gr["num_dst"], gr["num_edge"])
            np.testing.assert_array_equal(row.cpu().numpy().view(np.uint32), gr["row"])
            np.testing.assert_array_equal(col.cpu().numpy().view(np.uint32), gr["col"])
            # invariants that hold for any run of the reference
            assert ndst == (hi - lo if l == 1 else prev_src)
            dn = torch.from_numpy(nodes[:ndst].astype(np.int64)).to(dev)
            assert torch.equal(torch.bincount(col.long(), minlength=ndst), torch.clamp(ip[dn + 1] - ip[dn], max=fanout[l]))
            prev_src = nsrc
        o_split = oracle.get_miss_cache_index(h_table, nodes)
        assert int(m.num_miss) + int(m.num_cache) == len(nodes)
        for got, wv in zip(bt.cache_index_arrays(), o_split):
            np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), wv)
        assert bt.feat().cpu().numpy().tobytes() == oracle.extract(h_feat, nodes).tobytes()
        np.testing.assert_array_equal(bt.label().cpu().numpy(), h_label[h_seeds])
    np.testing.assert_array_equal(d_indices.cpu().numpy().view(np.uint32), h_indices)  # khop2's swaps, all three batches
    assert (h_indices != indices.cpu().numpy().view(np.uint32)).any() == (kind == "khop2")  # khop0 never writes the CSR


@pytest.mark.gpu
@pytest.mark.parametrize("help_after", [-1, 0])
def test_products_shape_overlapped_batches_bit_exact(help_after):
    """The loop bench.py times -- fgnn_sampler_run_range: whole batches in flight together over three streams, sample ->
    cache split -> feature / label gather -- on the products-shaped graph, three layers: 72 consecutive batches, every
    block, node list, cache index array, gathered row and label and khop2's mutated CSR identical to the oracle's replay
    of the same sequence (tools/soak_overlapped.py; thousands of batches there: profiles/r05_j_soak_overlapped.txt).
    help_after 0: the single-pass kernels' waits take the helping path."""
    import argparse
    import importlib.util
    from fgnn_hip import lib
    spec = importlib.util.spec_from_file_location("soak_overlapped", os.path.join(ROOT, "tools", "soak_overlapped.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    try:
        mod.run(argparse.Namespace(kind="khop2", fanout="10,5,5", rounds=3, per_round=24, streams=3, batch=8000,
                                   seed=0x5A4D47, help_after=help_after))
    finally:
        if help_after >= 0:  # back to what the rest of the test process runs with (tests/conftest.py)
            env = os.environ.get("FGNN_SCAN_HELP_AFTER")
            lib.load().fgnn_debug_set_scan_help_after(int(env) if env is not None else -1)  # -1: no override
