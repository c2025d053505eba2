"""Pins oracle/ (the CPU restatement) against outputs of the REFERENCE's own CPU code.

The fixtures under tests/golden/ were produced by tests/golden/make_golden.py, which ran the
reference's unmodified cpu_sampling_khop{0,2}.cc / cpu_random.cc / cpu_hashtable2.cc /
cpu_extraction.cc (compiled in place by `make -C oracle _ref`).  Bar: bit-exact.
"""
import json
import os

import numpy as np
import pytest


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("sample,fname", [("khop0", "khop0_pipeline.npz"), ("khop2", "khop2_pipeline.npz")])
def test_pipeline_matches_reference_cpu(oracle, golden_dir, sample, fname):
    """DoCPUSample order (cpu/cpu_loops.cc:55-191) over 3 batches in one process: the mt19937 stream and
    khop2's in-place CSR mutation carry across layers and batches."""
    g = _load(golden_dir, fname)
    indptr = g["indptr"]
    indices = g["indices"].copy()
    fanouts = [int(x) for x in g["fanouts"]]
    st = oracle.KHOP0 if sample == "khop0" else oracle.KHOP2
    rng = oracle.make_rng(oracle.RNG_MT_CPU_TWIN)
    ht = oracle.HashTable(len(indptr) - 1, len(indptr) - 1)
    b = 0
    while f"b{b}_seeds" in g:
        task = oracle.do_sample(indptr, indices, g[f"b{b}_seeds"], fanouts, st, rng, batch_key=b, ht=ht)
        for li in range(len(fanouts)):
            p = f"b{b}_l{li}"
            gr = task["graphs"][li]
            assert gr["num_edge"] == len(g[p + "_out_dst"])
            np.testing.assert_array_equal(gr["row"], g[p + "_row"])
            np.testing.assert_array_equal(gr["col"], g[p + "_col"])
            assert gr["num_src"] == len(g[p + "_unique"])
        np.testing.assert_array_equal(task["input_nodes"], g[f"b{b}_l0_unique"])
        b += 1
    assert b == 3
    np.testing.assert_array_equal(indices, g["indices_after"])
    if sample == "khop0":
        np.testing.assert_array_equal(indices, g["indices"])  # khop0 never mutates the CSR
    else:
        assert (indices != g["indices"]).any()                # khop2 does


@pytest.mark.parametrize("sample", ["khop0", "khop2"])
def test_per_call_outputs_match_reference_cpu(oracle, golden_dir, sample):
    """Same fixtures, checked stage by stage: sampler COO, Populate/MapNodes unique list, MapEdges."""
    g = _load(golden_dir, f"{sample}_pipeline.npz")
    indptr = g["indptr"]
    indices = g["indices"].copy()
    fanouts = [int(x) for x in g["fanouts"]]
    fn = oracle.sample_khop0 if sample == "khop0" else oracle.sample_khop2
    rng = oracle.make_rng(oracle.RNG_MT_CPU_TWIN)
    ht = oracle.HashTable(len(indptr) - 1, len(indptr) - 1)
    for b in range(3):
        ht.reset()
        assert ht.fill_unique(g[f"b{b}_seeds"]) == 0
        cur = g[f"b{b}_seeds"]
        for li in range(len(fanouts) - 1, -1, -1):
            p = f"b{b}_l{li}"
            src, dst = fn(indptr, indices, cur, fanouts[li], rng, b, li)
            np.testing.assert_array_equal(src, g[p + "_out_src"])
            np.testing.assert_array_equal(dst, g[p + "_out_dst"])
            uniq = ht.fill_duplicates(dst)
            np.testing.assert_array_equal(uniq, g[p + "_unique"])
            np.testing.assert_array_equal(uniq[:len(cur)], cur)  # next input starts with previous seeds
            ns, nd = ht.map_edges(src, dst)
            np.testing.assert_array_equal(ns, g[p + "_col"])
            np.testing.assert_array_equal(nd, g[p + "_row"])
            cur = uniq


def test_edge_cases_match_reference_cpu(oracle, golden_dir):
    g = _load(golden_dir, "edge_cases.npz")
    indptr = g["indptr"]
    indices = g["indices"].copy()
    rng = oracle.make_rng(oracle.RNG_MT_CPU_TWIN)
    for tag in g["order"]:
        tag = str(tag)
        sample, iname, f = tag.split("_")
        fanout = int(f[1:])
        fn = oracle.sample_khop0 if sample == "khop0" else oracle.sample_khop2
        src, dst = fn(indptr, indices, g["in_" + iname], fanout, rng)
        np.testing.assert_array_equal(src, g[tag + "_src"], err_msg=tag)
        np.testing.assert_array_equal(dst, g[tag + "_dst"], err_msg=tag)
    np.testing.assert_array_equal(indices, g["indices_after"])
    # CPUHashTable2::Populate with duplicates inside one call and across calls
    ht = oracle.HashTable(7, 16)
    np.testing.assert_array_equal(ht.fill_duplicates(g["dup"]), g["dup_unique"])
    np.testing.assert_array_equal(ht.fill_duplicates(g["more"]), g["more_unique"])
    ns, _ = ht.map_edges(g["dup"], g["dup"])
    np.testing.assert_array_equal(ns, g["dupmap"])


def test_extract_matches_reference_cpu(oracle, golden_dir):
    g = _load(golden_dir, "extract.npz")
    for name in ("f32_d7", "f32_d100", "i64_d1", "u8_d3", "f16_d5"):
        out = oracle.extract(g[name + "_src"], g["index"])
        assert out.tobytes() == g[name + "_out"].tobytes(), name


def test_mock_extract_matches_reference_cpu(oracle, golden_dir):
    """CPUMockExtract (cpu/cpu_extraction.cc:44-62, 92-116; the GPU twin gpu_mock_extract, cuda_extraction.cu:50-70):
    the reference's own masking of row ids to a 2^SAMGRAPH_EMPTY_FEAT-row table, ids up to 2^32 - 1"""
    g = _load(golden_dir, "mock_extract.npz")
    names = sorted(k[:-4] for k in g.files if k.endswith("_src"))
    assert len(names) == 5
    for name in names:
        bits = int(name.rsplit("_b", 1)[1])
        out = oracle.mock_extract(g[name + "_src"], g["index"], bits)
        assert out.tobytes() == g[name + "_out"].tobytes(), name
        # the mask is the whole difference to CPUExtract
        assert out.tobytes() == oracle.extract(g[name + "_src"], g["index"] & ((1 << bits) - 1)).tobytes(), name


def test_shuffle_matches_libstdcxx(oracle, golden_dir):
    """Shufflers' Fisher-Yates with std::default_random_engine(epoch) (dist/dist_shuffler.cc:112-131):
    cumulative over epochs, seed = epoch."""
    g = _load(golden_dir, "shuffle.npz")
    for key in g.files:
        n = int(key[1:])
        data = np.arange(n, dtype=np.uint32)
        for epoch, want in enumerate(g[key]):
            data = oracle.shuffle_minstd0(data, epoch)
            np.testing.assert_array_equal(data, want, err_msg=f"{key} epoch {epoch}")


def test_philox_known_answers(oracle):
    """Random123 kat_vectors for philox4x32-10."""
    assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_formulas(oracle):
    # PredictNumNodes (common.cc:330-339): SURVEY 8: bs 8000, [25,10] -> 2 288 000; GCN [5,10,15] -> 8 448 000
    assert oracle.predict_num_nodes(8000, [25, 10]) == 2288000
    assert oracle.predict_num_nodes(8000, [5, 10, 15]) == 8448000
    assert oracle.predict_num_nodes(8000, [25, 10], 1) == 8000 * 26
    # TableSize (cuda_hashtable.cu:125-128): 2 288 000 -> 2^21 << 2 = 8 388 608; TableSize(75,3) = 512
    assert oracle.table_size(2288000, 2) == 8388608
    assert oracle.table_size(75, 3) == 512
    assert oracle.table_size(8448000, 2) == 32 * 1024 * 1024
    # DistShuffler partition (dist_shuffler.cc:36-79) for papers100M: 151 steps, 2 samplers -> 75 + 76
    p0 = oracle.dist_shuffler_partition(1207179, 8000, 0, 2)
    p1 = oracle.dist_shuffler_partition(1207179, 8000, 1, 2)
    assert p0["epoch_step"] == 151 and p0["num_local_step"] == 75 and p1["num_local_step"] == 76
    assert p0["dataset_offset"] == 0 and p1["dataset_offset"] == 75 * 8000
    assert p0["last_batch_size"] == 8000 and p1["last_batch_size"] == 1207179 % 8000
    assert p0["local_data_size"] == 75 * 8000 and p1["local_data_size"] == 1207179 - 75 * 8000
    # DistAlignedShuffler (dist_shuffler_aligned.cc:45-71), papers100M on 8 workers: 1 207 179 -> 1 207 184 ids,
    # 150 898 per worker, 19 steps each (152 per epoch), last batch 150 898 - 18 * 8000 = 6898
    for w in range(8):
        a = oracle.aligned_shuffler_partition(1207179, 8000, w, 8)
        assert a == dict(padded_size=1207184, local_data_size=150898, num_local_step=19, epoch_step=152,
                         step_offset=19 * w, dataset_offset=150898 * w, last_batch_size=6898)
    a = oracle.aligned_shuffler_partition(16000, 8000, 1, 2)
    assert a["padded_size"] == 16000 and a["num_local_step"] == 1 and a["last_batch_size"] == 8000
    # the batches of all workers of an epoch are the shuffled padded set, cut per worker
    train = np.arange(100, 1103, dtype=np.uint32)  # 1003 ids, 4 workers -> padded with the first id
    got = {}
    for w in range(4):
        for epoch, step, ids in oracle.aligned_shuffler_batches(train, 100, w, 4, 2):
            got[(epoch, step)] = ids
    assert sorted(got) == [(e, s) for e in range(2) for s in range(12)]
    data = np.concatenate([train, train[:1]])
    for e in range(2):
        data = oracle.shuffle_minstd0(data, e)
        for w in range(4):
            mine = np.concatenate([got[(e, 3 * w + k)] for k in range(3)])
            np.testing.assert_array_equal(mine, data[251 * w:251 * (w + 1)])
            assert [len(got[(e, 3 * w + k)]) for k in range(3)] == [100, 100, 51]


def test_cache_split_and_rank(oracle):
    # presample rank (dist/pre_sampler.cc:131-162): freq desc, node id desc on ties
    freq = np.array([3, 0, 5, 3, 0, 1], dtype=np.uint32)
    rank = oracle.presample_rank(freq)
    np.testing.assert_array_equal(rank, [2, 3, 0, 5, 4, 1])
    table = oracle.cache_table_build(rank, 3, 6)
    np.testing.assert_array_equal(table, [2, oracle.EMPTY, 0, 1, oracle.EMPTY, oracle.EMPTY])
    nodes = np.array([5, 3, 1, 2, 0, 4], dtype=np.uint32)
    ms, md, cs, cd = oracle.get_miss_cache_index(table, nodes)
    np.testing.assert_array_equal(ms, [5, 1, 4])      # global ids of misses, in batch order
    np.testing.assert_array_equal(md, [0, 2, 5])      # their batch positions
    np.testing.assert_array_equal(cs, [1, 0, 2])      # cache slots of hits
    np.testing.assert_array_equal(cd, [1, 3, 4])
    feat = np.arange(6 * 4, dtype=np.float32).reshape(6, 4)
    cache = oracle.extract(feat, rank[:3])
    out = np.zeros((6, 4), dtype=np.float32)
    oracle.combine(out, oracle.extract(feat, ms), None, md)
    oracle.combine(out, cache, cs, cd)
    np.testing.assert_array_equal(out, feat[nodes])


def test_python_constants_fixture_present(golden_dir):
    with open(os.path.join(golden_dir, "py_constants.json")) as f:
        c = json.load(f)
    assert c["constants"]["kKHop2"] == 5 and c["constants"]["kArch5"] == 5


def test_hash_dedup_sampler_invariants(oracle):
    """weighted_khop_hash_dedup has no CPU twin in the reference; the restatement is checked against what
    cuda_sampling_weighted_khop_hash_dedup.cu:83-111 guarantees: rows of length <= fanout are taken whole in CSR order,
    longer rows yield `fanout` DISTINCT values, each one a neighbour or its alias, seeds in input order."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fgnn-artifacts_amd"))
    from fgnn_hip import synth
    num_node, fanout = 3000, 7
    indptr, indices = synth.powerlaw_csr(num_node, 60000, seed=5)
    prob, alias = synth.alias_tables(indptr, indices)
    rng = oracle.make_rng(oracle.RNG_PHILOX, 99)
    inp = np.random.default_rng(3).choice(num_node, 900, replace=False).astype(np.uint32)
    src, dst = oracle.sample_weighted_khop_hash_dedup(indptr, indices, prob, alias, inp, fanout, rng, 1, 0)
    pos = 0
    for rid in inp:
        a, b = int(indptr[rid]), int(indptr[rid + 1])
        k = 0
        while pos + k < len(src) and src[pos + k] == rid:
            k += 1
        mine = dst[pos:pos + k]
        if b - a <= fanout:
            np.testing.assert_array_equal(mine, indices[a:b])
        else:
            selectable = set(indices[a:b].tolist()) | set(alias[a:b].tolist())
            assert 1 <= k <= fanout and len(set(mine.tolist())) == k and set(mine.tolist()) <= selectable
            # short only when the row offers few distinct values (the build's give-up rule; the reference would spin)
            assert k == fanout or len(selectable) < 2 * fanout
        pos += k
    assert pos == len(dst)
    # same seed and key: same draws
    src2, dst2 = oracle.sample_weighted_khop_hash_dedup(indptr, indices, prob, alias, inp, fanout, rng, 1, 0)
    np.testing.assert_array_equal(dst, dst2)


def test_reference_cpu_build_agrees_with_oracle_twin(oracle):
    """bench.py times oracle/_ref (the reference's own CPU sources) as its CPU baseline; the same entry point run with
    one thread must reproduce the oracle's CPU-twin mode (default-seeded mt19937) edge for edge."""
    if not oracle.RefBaseline.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle_py as o
from fgnn_hip import synth
N, fan, bs = 50000, [7, 4], 900
indptr, indices = synth.powerlaw_csr(N, 900000, seed=5)
cap = o.predict_num_nodes(bs, fan)
seeds = np.random.default_rng(1).choice(N, bs, replace=False).astype(np.uint32)
feat = synth.node_features(1 << 12, 8)
out = np.zeros((cap, 8), dtype=np.float32)
rb = o.RefBaseline(N, bs * 5 * 7 + bs * 4, cap, 1)
ind = indices.copy()
e, n = rb.sample_batch(indptr, ind, seeds, fan, o.KHOP2, feat, 12, out)
oind = indices.copy()
w = o.do_sample(indptr, oind, seeds, fan, o.KHOP2, o.make_rng(o.RNG_MT_CPU_TWIN, 0), 0, o.HashTable(N, cap))
assert (e, n) == (w["total_edges"], len(w["input_nodes"])), (e, n, w["total_edges"])
assert (ind == oind).all()
assert (out[:n] == feat[w["input_nodes"] & 4095]).all()
print("same")
''' % (os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"),
       os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fgnn-artifacts_amd"))
    # own process: the reference keeps its mt19937 thread_local per process, it must start fresh
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert p.returncode == 0 and "same" in p.stdout, p.stderr[-2000:]


def test_all_neighbour_restatement_is_the_closed_neighbourhood(oracle):
    """DoGPUSampleAllNeighbour (cuda_loops.cc:500-571) has no random draw: its input_nodes are the closed L-hop
    neighbourhood of the seeds.  Checked against sparse matrix reachability (T2: formula read from the source)."""
    import scipy.sparse as sp
    from fgnn_hip import synth
    indptr, indices = synth.powerlaw_csr(500, 4000, seed=5)
    n = len(indptr) - 1
    a = sp.csr_matrix((np.ones(len(indices), dtype=np.int64), indices.astype(np.int64), indptr.astype(np.int64)),
                      shape=(n, n))
    seeds = np.random.default_rng(9).permutation(n)[:11].astype(np.uint32)
    nbrs = oracle.extract_neighbour(indptr, indices, seeds)
    np.testing.assert_array_equal(nbrs, np.concatenate([indices[indptr[s]:indptr[s + 1]] for s in seeds]))
    for layers in (0, 1, 2, 3):
        reach = np.zeros(n, dtype=np.int64)
        reach[seeds] = 1
        for _ in range(layers):
            reach = reach + a.T.dot(reach)  # row v lists the neighbours v pulls in
        got = oracle.sample_all_neighbour(indptr, indices, seeds, layers)
        np.testing.assert_array_equal(got[:len(seeds)], seeds)
        assert len(np.unique(got)) == len(got)
        np.testing.assert_array_equal(np.sort(got), np.flatnonzero(reach).astype(np.uint32))
