"""The sampling LAWS of the oracle (tier T3: "chi-square uniformity / weight-proportionality").

The GPU kernels are bit-identical to the oracle under the same Philox stream (tests/test_hip_parity.py), and the oracle's
khop0 / khop2 bodies are pinned by the reference's own CPU sources.  The with-replacement samplers and the random walk have
no CPU twin in the reference: their oracle bodies follow the CUDA kernels by reading.  What CAN be checked without the
reference is that each body samples from the distribution its kernel is supposed to sample from -- a wrong index, a
swapped table, an off-by-one in a binary search or a missing restart shows up here:
  * khop0 / khop2 (cuda_sampling_khop0.cu:41-90, khop2.cu:41-89): every neighbour of a long row is included with
    probability fanout / degree;
  * khop1 (khop1.cu:42-72): uniform with replacement;
  * weighted_khop_prefix (prefix.cu:41-92) and weighted_khop / alias (weighted_khop.cu:41-76): proportional to the
    edge weights;
  * random walk (random_walk.cu:43-109): uniform steps, geometric stopping with the restart probability, top-K counts.
Deterministic (fixed seeds): a pass is a pass for good.  Thresholds: chi-square quantile at p = 1e-9."""
import os
import sys

import numpy as np
import pytest
from scipy.stats import chi2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

SEED = 0x5A4D47
DEG = 12          # neighbours of every sampled row: node ids 0 .. DEG-1
NUM_SEEDS = 1500  # nodes DEG .. DEG+NUM_SEEDS-1 all have the same row


@pytest.fixture(scope="module")
def oracle():
    import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="module")
def star():
    """CSR in which every seed has the row [0..DEG-1]; the neighbours have that row too (second walk step)."""
    n = DEG + NUM_SEEDS
    indptr = (np.arange(n + 1, dtype=np.uint64) * DEG).astype(np.uint32)
    indices = np.tile(np.arange(DEG, dtype=np.uint32), n)
    seeds = np.arange(DEG, n, dtype=np.uint32)
    return indptr, indices, seeds


def _chi2_ok(observed, expected, scale=1.0):
    """sum (O-E)^2 / (E * scale) against the chi-square quantile at 1e-9 with len-1 degrees of freedom"""
    observed, expected = np.asarray(observed, dtype=np.float64), np.asarray(expected, dtype=np.float64)
    stat = float(((observed - expected) ** 2 / (expected * scale)).sum())
    limit = float(chi2.isf(1e-9, len(observed) - 1))
    return stat, limit


@pytest.mark.parametrize("kind", ["khop0", "khop2"])
def test_without_replacement_inclusion_is_uniform(oracle, star, kind):
    indptr, indices, seeds = star
    indices = indices.copy()  # khop2 permutes the rows in place; the SET of neighbours of a row never changes
    fn = oracle.sample_khop0 if kind == "khop0" else oracle.sample_khop2
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    fanout, calls = 5, 8
    counts = np.zeros(DEG)
    for b in range(calls):
        src, dst = fn(indptr, indices, seeds, fanout, rng, 100 + b, 0)
        assert len(dst) == len(seeds) * fanout
        per_seed = dst.reshape(len(seeds), fanout)
        assert (np.sort(per_seed, axis=1)[:, 1:] != np.sort(per_seed, axis=1)[:, :-1]).all()  # no repeats within a seed
        counts += np.bincount(dst, minlength=DEG)
    draws = calls * len(seeds)
    p = fanout / DEG
    # inclusion counts of simple random samples: covariance B p (1-p) n/(n-1) (I - J/n)
    stat, limit = _chi2_ok(counts, np.full(DEG, draws * p), scale=(1 - p) * DEG / (DEG - 1))
    assert stat < limit, (kind, stat, limit, counts)
    if kind == "khop2":
        assert (np.sort(indices.reshape(-1, DEG), axis=1) == np.arange(DEG)).all()  # rows are permutations still


def _weights():
    return np.arange(1, DEG + 1, dtype=np.float64)  # 1, 2, ..., 12: every neighbour has its own probability


def _one_draw_counts(fn, indptr, indices, seeds, rng, calls, *tables):
    counts = np.zeros(DEG)
    for b in range(calls):
        src, dst = fn(indptr, indices, *tables, seeds, 1, rng, 200 + b, 0)
        assert len(dst) == len(seeds)  # fanout 1: one draw per seed, nothing to de-duplicate
        counts += np.bincount(dst, minlength=DEG)
    return counts


def test_khop1_is_uniform_with_replacement(oracle, star):
    indptr, indices, seeds = star
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    calls = 10
    counts = _one_draw_counts(oracle.sample_khop1, indptr, indices, seeds, rng, calls)
    stat, limit = _chi2_ok(counts, np.full(DEG, calls * len(seeds) / DEG))
    assert stat < limit, (stat, limit, counts)
    # and with a fan-out: at most `fanout` per seed, only a draw equal to the seed's NEXT draw is dropped
    src, dst = oracle.sample_khop1(indptr, indices, seeds, 6, rng, 999, 0)
    per = np.bincount(src - DEG if src.min() >= DEG else src, minlength=len(seeds))
    assert per.max() <= 6 and per.min() >= 1
    same_seed = src[1:] == src[:-1]
    assert not (same_seed & (dst[1:] == dst[:-1])).any()


def test_weighted_prefix_is_proportional_to_the_weights(oracle, star):
    indptr, indices, seeds = star
    w = _weights()
    prefix = np.tile(np.cumsum(w).astype(np.float32), len(indptr) - 1)
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    calls = 10
    counts = _one_draw_counts(oracle.sample_weighted_khop_prefix, indptr, indices, seeds, rng, calls, prefix)
    stat, limit = _chi2_ok(counts, calls * len(seeds) * w / w.sum())
    assert stat < limit, (stat, limit, counts)
    # a uniform law must FAIL the same check (the test has teeth)
    stat_u, _ = _chi2_ok(np.full(DEG, calls * len(seeds) / DEG), calls * len(seeds) * w / w.sum())
    assert stat_u > 10 * limit


def _vose(w):
    """prob / alias (positions) of one row, create_alias_table.cc:100-170"""
    n = len(w)
    q = w / w.sum() * n
    prob, alias = np.ones(n, dtype=np.float32), np.zeros(n, dtype=np.int64)
    small, large = [i for i in range(n) if q[i] < 1.0], [i for i in range(n) if q[i] >= 1.0]
    while small and large:
        s, l = small.pop(0), large.pop(0)
        prob[s], alias[s] = q[s], l
        q[l] -= 1.0 - q[s]
        (small if q[l] < 1.0 else large).append(l)
    return prob, alias


def test_alias_method_is_proportional_to_the_weights(oracle, star):
    indptr, indices, seeds = star
    w = _weights()
    prob_row, alias_row = _vose(w)
    prob = np.tile(prob_row, len(indptr) - 1)
    alias = np.tile(np.arange(DEG, dtype=np.uint32)[alias_row], len(indptr) - 1)  # the table holds node ids
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    calls = 10
    counts = _one_draw_counts(oracle.sample_weighted_khop, indptr, indices, seeds, rng, calls, prob, alias)
    stat, limit = _chi2_ok(counts, calls * len(seeds) * w / w.sum())
    assert stat < limit, (stat, limit, counts)


def test_hash_dedup_sampler_returns_distinct_neighbours(oracle, star):
    indptr, indices, seeds = star
    prob_row, alias_row = _vose(_weights())
    prob = np.tile(prob_row, len(indptr) - 1)
    alias = np.tile(np.arange(DEG, dtype=np.uint32)[alias_row], len(indptr) - 1)
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    src, dst = oracle.sample_weighted_khop_hash_dedup(indptr, indices, prob, alias, seeds, 5, rng, 7, 0)
    assert len(dst) == 5 * len(seeds) and dst.max() < DEG
    per_seed = np.sort(dst.reshape(len(seeds), 5), axis=1)
    assert (per_seed[:, 1:] != per_seed[:, :-1]).all()
    # heavier neighbours are selected more often
    counts = np.bincount(dst, minlength=DEG)
    assert counts[-1] > counts[0] * 2


def test_random_walk_steps_stops_and_counts(oracle, star):
    indptr, indices, seeds = star
    rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
    # one step per walk: W uniform draws over the row, all DEG distinct values kept (K = DEG): counts = visit counts
    W = 8
    src, dst, cnt = oracle.sample_random_walk(indptr, indices, seeds, 1, 0.5, W, DEG, rng, 11, 0)
    tot = np.zeros(DEG)
    np.add.at(tot, dst, cnt)
    assert tot.sum() == W * len(seeds)
    stat, limit = _chi2_ok(tot, np.full(DEG, W * len(seeds) / DEG))
    assert stat < limit, (stat, limit, tot)
    # per seed: counts non-increasing in output order, distinct destinations
    first = np.flatnonzero(np.r_[True, src[1:] != src[:-1]])
    for a, b in zip(first[:200], np.r_[first[1:], len(src)][:200]):
        assert (np.diff(cnt[a:b].astype(np.int64)) <= 0).all() and len(set(dst[a:b])) == b - a
    # two steps, restart probability p: a walk continues to its second step with probability 1 - p
    for p in (0.0, 0.5, 0.9):
        src, dst, cnt = oracle.sample_random_walk(indptr, indices, seeds, 2, p, W, DEG, rng, 12, 0)
        walks = W * len(seeds)
        second = float(cnt.sum()) - walks  # every walk makes its first step (no empty rows here)
        sigma = (walks * p * (1 - p)) ** 0.5
        assert abs(second - walks * (1 - p)) <= 6 * sigma + 1e-9, (p, second, walks * (1 - p), sigma)
    # top-K keeps the K most visited: with K < DEG the kept counts are the largest ones
    src, dst, cnt = oracle.sample_random_walk(indptr, indices, seeds[:50], 1, 0.5, 40, 3, rng, 13, 0)
    srcf, dstf, cntf = oracle.sample_random_walk(indptr, indices, seeds[:50], 1, 0.5, 40, DEG, rng, 13, 0)
    for sd in np.unique(src):
        kept = np.sort(cnt[src == sd])[::-1]
        full = np.sort(cntf[srcf == sd])[::-1]
        assert (kept == full[:len(kept)]).all() and len(kept) == min(3, len(full))
