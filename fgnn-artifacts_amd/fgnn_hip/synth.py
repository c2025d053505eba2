"""Synthetic datasets in the reference's on-disk layout (samgraph/common/engine.cc:73-264).

No real dataset can be downloaded here, so every test and bench input is generated: a CSR whose row
degrees follow a truncated power law and whose neighbour ids are skewed towards low ids (hubs),
which reproduces the two properties the hot path is sensitive to: many rows longer than the
fanout, and heavy duplication among sampled neighbours.
"""
import os

import numpy as np

# datagen/products.py:92-98, datagen/papers100M.py:88-94, datagen/twitter.sh:35-43, datagen/uk-2006-05.sh:35-43
DATASET_SHAPES = {
    "products": dict(num_node=2449029, num_edge=123718152, feat_dim=100, num_class=47, num_train=196615),
    "papers100M": dict(num_node=111059956, num_edge=1615685872, feat_dim=128, num_class=172, num_train=1207179),
    "twitter": dict(num_node=41652230, num_edge=1468365182, feat_dim=256, num_class=150, num_train=416500),
    "uk-2006-05": dict(num_node=77741046, num_edge=2965197340, feat_dim=256, num_class=150, num_train=1000000),
}


# a small graph with learnable labels (learnable_labels below) for the examples' --report-acc runs
LEARNABLE_SHAPE = dict(num_node=50000, num_edge=1000000, feat_dim=64, num_class=8, num_train=20000, num_valid=2000,
                       num_test=2000)


def powerlaw_degrees(num_node, num_edge, rng, alpha=1.8, zero_frac=0.02):
    """Degrees ~ Pareto(alpha) scaled so that they sum to num_edge exactly; a few isolated rows."""
    raw = rng.pareto(alpha, size=num_node) + 0.05
    raw[rng.random(num_node) < zero_frac] = 0.0
    deg = np.floor(raw * (num_edge / raw.sum())).astype(np.int64)
    short = int(num_edge - deg.sum())
    if short > 0:
        bump = rng.integers(0, num_node, size=short)
        np.add.at(deg, bump, 1)
    return deg


def powerlaw_csr(num_node, num_edge, seed=42, alpha=1.8, skew=2.0):
    """Returns (indptr u32[N+1], indices u32[E]).  Row = destination/seed, entries = sources."""
    rng = np.random.default_rng(seed)
    deg = powerlaw_degrees(num_node, num_edge, rng, alpha)
    indptr = np.zeros(num_node + 1, dtype=np.int64)
    np.cumsum(deg, out=indptr[1:])
    assert indptr[-1] == num_edge and num_edge < 2**32
    u = rng.random(num_edge)
    indices = np.minimum((num_node * u**skew).astype(np.int64), num_node - 1)
    # scatter hub ids over the id space so that hash-table and cache behaviour is not id-ordered
    perm_mul = 2654435761 % num_node
    while np.gcd(perm_mul, num_node) != 1:
        perm_mul += 1
    indices = (indices * perm_mul) % num_node
    return indptr.astype(np.uint32), indices.astype(np.uint32)


def prob_prefix_table(indptr, indices, seed=7):
    """Per-row inclusive prefix sums of edge weights (create_prob_prefix_table.cc:83-124):
    weight 100 if out-degree(src) < 10 else 1."""
    num_node = len(indptr) - 1
    out_deg = np.bincount(indices, minlength=num_node)
    w = np.where(out_deg[indices] < 10, 100.0, 1.0).astype(np.float32)
    prefix = np.empty_like(w)
    ip = indptr.astype(np.int64)
    for r in range(num_node):
        a, b = ip[r], ip[r + 1]
        if b > a:
            acc = np.float32(0)
            for k in range(a, b):  # sequential f32 accumulation like the tool
                acc = np.float32(acc + w[k])
                prefix[k] = acc
    return prefix


def alias_tables(indptr, indices, seed=9):
    """prob_table f32[E], alias_table u32[E] (node ids) by the queue-based Vose construction of
    utility/data-process/toolkit/weight/create_alias_table.cc:100-170, weights = kSrcSuffix policy."""
    from collections import deque
    num_node = len(indptr) - 1
    out_deg = np.bincount(indices, minlength=num_node)
    prob = np.ones(len(indices), dtype=np.float32)
    alias = np.zeros(len(indices), dtype=np.uint32)  # columns with prob 1 keep alias 0 (create_alias_table.cc:211)
    ip = indptr.astype(np.int64)
    for r in range(num_node):
        a, b = ip[r], ip[r + 1]
        n = int(b - a)
        if n == 0:
            continue
        w = np.where(out_deg[indices[a:b]] < 10, 100.0, 1.0).astype(np.float32)
        w = (w / w.sum(dtype=np.float32) * np.float32(n)).astype(np.float32)
        small, large = deque(i for i in range(n) if w[i] < 1.0), deque(i for i in range(n) if w[i] >= 1.0)
        while small and large:
            si, li = small.popleft(), large.popleft()
            prob[a + si] = w[si]
            alias[a + si] = indices[a + li]
            w[li] = np.float32(w[li] - (np.float32(1.0) - w[si]))
            (small if w[li] < 1.0 else large).append(li)
        for li in large:
            prob[a + li] = 1.0
        for si in small:
            prob[a + si] = 1.0
    return prob, alias.astype(np.uint32)


def cache_by_degree(indices, num_node):
    """Node ranking by descending (out-degree, id) -- utility/data-process/toolkit/cache/cache_by_degree.cc:29-48
    (a node's out-degree = the number of rows it appears in, graph_loader.cc:126-137)."""
    out = np.bincount(indices, minlength=num_node).astype(np.int64)
    ids = np.arange(num_node, dtype=np.int64)
    return np.lexsort((-ids, -out)).astype(np.uint32)


def node_features(num_node, dim, seed=3, dtype=np.float32):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((num_node, dim), dtype=np.float32).astype(dtype)


def learnable_labels(indptr, indices, feat, num_class, seed=5):
    """Labels a GNN can learn from the graph (for accuracy checks: random labels teach nothing): the class of node v is
    the argmax of a fixed random projection of x_v + mean_{u in row v} x_u -- what one mean-aggregator layer computes
    from a node's own features and its neighbours'.  Isolated nodes are labelled from their own features."""
    num_node = len(indptr) - 1
    ip = indptr.astype(np.int64)
    deg = np.diff(ip)
    nz = np.nonzero(deg)[0]
    sums = np.zeros((num_node, feat.shape[1]), dtype=np.float64)
    if len(nz):
        sums[nz] = np.add.reduceat(feat[indices.astype(np.int64)].astype(np.float64), ip[:-1][nz], axis=0)
    mean = sums / np.maximum(deg, 1)[:, None]
    proj = np.random.default_rng(seed).standard_normal((feat.shape[1], num_class))
    return np.argmax((feat.astype(np.float64) + mean) @ proj, axis=1).astype(np.uint64)


def write_dataset(root, name, num_node, num_edge, feat_dim, num_class, num_train, num_valid=0, num_test=0,
                  seed=42, with_prefix=False, with_alias=False, learnable=False):
    """Writes <root>/<name>/{meta.txt,indptr.bin,indices.bin,feat.bin,label.bin,*_set.bin}.  learnable: labels that
    follow from the features and the graph (learnable_labels) instead of uniform random ones."""
    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    indptr, indices = powerlaw_csr(num_node, num_edge, seed)
    rng = np.random.default_rng(seed + 1)
    perm = rng.permutation(num_node).astype(np.uint32)
    train, valid, test = perm[:num_train], perm[num_train:num_train + num_valid], \
        perm[num_train + num_valid:num_train + num_valid + num_test]
    indptr.tofile(os.path.join(d, "indptr.bin"))
    indices.tofile(os.path.join(d, "indices.bin"))
    feat = node_features(num_node, feat_dim, seed + 2)
    feat.tofile(os.path.join(d, "feat.bin"))
    random_labels = rng.integers(0, num_class, size=num_node, dtype=np.uint64)  # (drawn either way: same sets below)
    (learnable_labels(indptr, indices, feat, num_class) if learnable else random_labels).tofile(
        os.path.join(d, "label.bin"))
    del feat
    train.tofile(os.path.join(d, "train_set.bin"))
    valid.tofile(os.path.join(d, "valid_set.bin"))
    test.tofile(os.path.join(d, "test_set.bin"))
    if with_prefix:
        prob_prefix_table(indptr, indices).tofile(os.path.join(d, "prob_prefix_table.bin"))
    if with_alias:
        prob, alias = alias_tables(indptr, indices)
        prob.tofile(os.path.join(d, "prob_table.bin"))
        alias.tofile(os.path.join(d, "alias_table.bin"))
    # file-backed cache rankings (engine.cc:216-256): by in-degree (cache_by_degree.bin) and random
    cache_by_degree(indices, num_node).tofile(os.path.join(d, "cache_by_degree.bin"))
    rng.permutation(num_node).astype(np.uint32).tofile(os.path.join(d, "cache_by_random.bin"))
    with open(os.path.join(d, "meta.txt"), "w") as f:
        f.write(f"NUM_NODE {num_node}\nNUM_EDGE {num_edge}\nFEAT_DIM {feat_dim}\nNUM_CLASS {num_class}\n"
                f"NUM_TRAIN_SET {len(train)}\nNUM_VALID_SET {len(valid)}\nNUM_TEST_SET {len(test)}\n")
    return d
