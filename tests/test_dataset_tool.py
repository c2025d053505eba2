"""tools/dataset/fgnn_dataset: the files it writes against independent numpy restatements of the reference's
generators (utility/data-process/toolkit/{cache,weight,generator}) and a committed standard-library golden vector."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
TOOL = os.path.join(ROOT, "tools", "dataset", "fgnn_dataset")


@pytest.fixture(scope="module")
def tool():
    src = TOOL + ".cc"
    if not os.path.exists(TOOL) or os.path.getmtime(TOOL) < os.path.getmtime(src):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fopenmp", "-o", TOOL, src], check=True)
    return TOOL


def _run(tool, *args):
    p = subprocess.run([tool] + [str(a) for a in args], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return p.stdout


def test_rankings_and_weight_tables(tool, tmp_path):
    from fgnn_hip import synth
    num_node = 3000
    d = synth.write_dataset(str(tmp_path), "g", num_node, 50000, 4, 5, 300, 50, 50, seed=3)
    indptr = np.fromfile(os.path.join(d, "indptr.bin"), dtype=np.uint32)
    indices = np.fromfile(os.path.join(d, "indices.bin"), dtype=np.uint32)
    assert "ok" in _run(tool, "check", d)
    _run(tool, "cache-by-degree", d)
    np.testing.assert_array_equal(np.fromfile(os.path.join(d, "cache_by_degree.bin"), dtype=np.uint32),
                                  synth.cache_by_degree(indices, num_node))
    _run(tool, "cache-by-random", d)
    r = np.fromfile(os.path.join(d, "cache_by_random.bin"), dtype=np.uint32)
    assert sorted(r.tolist()) == list(range(num_node))
    _run(tool, "prob-prefix-table", d)
    np.testing.assert_array_equal(np.fromfile(os.path.join(d, "prob_prefix_table.bin"), dtype=np.float32),
                                  synth.prob_prefix_table(indptr, indices))
    _run(tool, "alias-table", d, "kSrcSuffix")
    prob, alias = synth.alias_tables(indptr, indices)
    np.testing.assert_array_equal(np.fromfile(os.path.join(d, "prob_table.bin"), dtype=np.float32), prob)
    np.testing.assert_array_equal(np.fromfile(os.path.join(d, "alias_table.bin"), dtype=np.uint32), alias)
    # the random policies are reproducible and stay valid tables
    _run(tool, "prob-prefix-table", d, "kDefault")
    t = np.fromfile(os.path.join(d, "prob_prefix_table.bin"), dtype=np.float32)
    for row in (0, 17, num_node - 1):
        seg = t[indptr[row]:indptr[row + 1]]
        assert (np.diff(seg) >= 1.0).all() and (seg[:1] >= 1.0).all()


def test_heuristic_degree_hop_rankings_and_64bit_copies(tool, tmp_path):
    """cache_by_heuristic.cc:54-88, cache_by_degree_hop.cc:30-165 and generator/32to64.cc:33-82 restated in numpy"""
    from fgnn_hip import synth
    num_node = 2000
    d = synth.write_dataset(str(tmp_path), "g", num_node, 9000, 4, 5, 40, 20, 20, seed=8)
    indptr = np.fromfile(os.path.join(d, "indptr.bin"), dtype=np.uint32).astype(np.int64)
    indices = np.fromfile(os.path.join(d, "indices.bin"), dtype=np.uint32)
    train = np.fromfile(os.path.join(d, "train_set.bin"), dtype=np.uint32)

    def by_degree(deg):  # descending (degree, id)
        return np.lexsort((-np.arange(num_node), -deg.astype(np.int64))).astype(np.uint32)

    out_deg = np.bincount(indices, minlength=num_node)
    _run(tool, "cache-by-heuristic", d)
    want, seen = list(train), set(train.tolist())
    for v in train:
        for u in indices[indptr[v]:indptr[v + 1]]:
            if u not in seen:
                seen.add(u)
                want.append(u)
    want += [v for v in by_degree(out_deg) if v not in seen]
    np.testing.assert_array_equal(np.fromfile(os.path.join(d, "cache_by_heuristic.bin"), dtype=np.uint32),
                                  np.array(want, dtype=np.uint32))
    _run(tool, "cache-by-degree-hop", d)
    reached = np.zeros(num_node, dtype=bool)
    reached[train] = True
    for _ in range(2):
        rows = np.flatnonzero(reached)
        reached[np.concatenate([indices[indptr[v]:indptr[v + 1]] for v in rows])] = True
    rows = np.flatnonzero(reached)
    sub = np.bincount(np.concatenate([indices[indptr[v]:indptr[v + 1]] for v in rows]), minlength=num_node)
    deg = np.where(reached, sub | 0x40000000, out_deg)
    got = np.fromfile(os.path.join(d, "cache_by_degree_hop.bin"), dtype=np.uint32)
    np.testing.assert_array_equal(got, by_degree(deg))
    assert 0 < reached.sum() < num_node and reached[got[:reached.sum()]].all()
    _run(tool, "32to64", d)
    for name in ("indptr", "indices", "train_set", "test_set", "valid_set"):
        a32 = np.fromfile(os.path.join(d, name + ".bin"), dtype=np.uint32)
        np.testing.assert_array_equal(np.fromfile(os.path.join(d, name + "64.bin"), dtype=np.uint64), a32)


def test_fake_optimal_ranking(tool, tmp_path):
    """cache_by_fake_optimal.cc:66-185 restated with Python floats (IEEE doubles, same multiplication order): expected
    number of one-node batches whose 2-hop [25,10]-style sample touches a node; ranking = descending (value, id)."""
    from fgnn_hip import synth
    num_node = 1500
    d = synth.write_dataset(str(tmp_path), "g", num_node, 30000, 4, 5, 120, 20, 20, seed=4)
    indptr = np.fromfile(os.path.join(d, "indptr.bin"), dtype=np.uint32).astype(np.int64)
    indices = np.fromfile(os.path.join(d, "indices.bin"), dtype=np.uint32)
    train = np.fromfile(os.path.join(d, "train_set.bin"), dtype=np.uint32)

    def expectation(f0, f1, batch):
        exp = [0.0] * num_node
        for b0 in range(0, len(train), batch):
            mine = [int(t) for t in train[b0:b0 + batch]]
            m1, m2, order = {}, {}, list(mine)
            for t in mine:
                row = indices[indptr[t]:indptr[t + 1]]
                miss = max(0.0, 1 - f1 / len(row)) if len(row) else 0.0
                for v in row:
                    v = int(v)
                    m1[v] = m1.get(v, 1.0) * miss
                    if v not in order:
                        order.append(v)
            for t in mine:
                m1[t] = 0.0
            for h in list(order):
                row = indices[indptr[h]:indptr[h + 1]]
                if not len(row):
                    continue
                path_miss = 1 - (1 - m1.get(h, 1.0)) * min(1.0, f0 / len(row))
                for w in row:
                    w = int(w)
                    m2[w] = m2.get(w, 1.0) * path_miss
                    if w not in order:
                        order.append(w)
            for t in mine:
                m2[t] = 0.0
            for c in order:
                a, b = m1.get(c, 1.0), m2.get(c, 1.0)
                if not (a == 1 and b == 1):
                    exp[c] += 1 - a * b
        return np.array(exp)

    for args, (f0, f1, batch) in (((), (25, 10, 1)), ((3, 2, 7), (3, 2, 7))):
        _run(tool, "cache-by-fake-optimal", d, *args)
        got = np.fromfile(os.path.join(d, "cache_by_fake_optimal.bin"), dtype=np.uint32)
        exp = expectation(f0, f1, batch)
        want = np.lexsort((-np.arange(num_node), -exp)).astype(np.uint32)
        np.testing.assert_array_equal(got, want)
        assert exp[got[0]] >= 1.0 and sorted(got.tolist()) == list(range(num_node))


def test_cache_by_random_matches_standard_library_golden(tool, tmp_path, golden_dir):
    d = tmp_path / "g"
    d.mkdir()
    n = 1000
    (d / "meta.txt").write_text(f"NUM_NODE {n}\nNUM_EDGE 0\nFEAT_DIM 1\nNUM_CLASS 1\nNUM_TRAIN_SET 0\nNUM_VALID_SET 0\n"
                                "NUM_TEST_SET 0\n")
    np.zeros(n + 1, dtype=np.uint32).tofile(d / "indptr.bin")
    np.zeros(0, dtype=np.uint32).tofile(d / "indices.bin")
    _run(tool, "cache-by-random", d)
    want = np.loadtxt(os.path.join(golden_dir, "cache_by_random_1000.txt"), dtype=np.uint32)
    np.testing.assert_array_equal(np.fromfile(d / "cache_by_random.bin", dtype=np.uint32), want)


def test_coo_to_dataset(tool, tmp_path):
    from fgnn_hip import synth
    num_node, num_edge = 2000, 30000
    indptr, indices = synth.powerlaw_csr(num_node, num_edge, seed=8)
    dst = np.repeat(np.arange(num_node, dtype=np.uint32), np.diff(indptr.astype(np.int64)))
    perm = np.random.default_rng(1).permutation(num_edge)
    coo = np.stack([indices[perm], dst[perm]], axis=1).astype(np.uint32)  # (src, dst) pairs
    d = tmp_path / "g"
    d.mkdir()
    (d / "meta.txt").write_text(f"NUM_NODE {num_node}\nNUM_EDGE {num_edge}\nFEAT_DIM 1\nNUM_CLASS 1\nNUM_TRAIN_SET 200\n"
                                "NUM_VALID_SET 40\nNUM_TEST_SET 60\n")
    coo.tofile(tmp_path / "coo.bin")
    _run(tool, "coo-to-dataset", d, tmp_path / "coo.bin")
    np.testing.assert_array_equal(np.fromfile(d / "indptr.bin", dtype=np.uint32), indptr)
    got = np.fromfile(d / "indices.bin", dtype=np.uint32)
    for row in range(num_node):  # rows are written with ascending sources
        a, b = int(indptr[row]), int(indptr[row + 1])
        np.testing.assert_array_equal(got[a:b], np.sort(indices[a:b]))
    sets = [np.fromfile(d / f"{s}_set.bin", dtype=np.uint32) for s in ("train", "valid", "test")]
    assert [len(s) for s in sets] == [200, 40, 60]
    allv = np.concatenate(sets)
    assert len(set(allv.tolist())) == 300 and (np.diff(indptr.astype(np.int64))[allv] > 0).all()
    assert "ok" in _run(tool, "check", d)
