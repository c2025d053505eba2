#!/usr/bin/env python3
"""Drives the engine through the reference's Python API (import samgraph.torch as sam) exactly like
example/samgraph/train_graphsage.py (arch1) and example/samgraph/multi_gpu/train_graphsage.py (arch5) do,
and checks every batch bit-for-bit against the oracle.  Run as a subprocess by tests/test_engine_gpu.py
(the engine is a process-wide singleton, like the reference's).

usage: engine_runner.py <arch1|arch5> <sample_type> <workdir> [num_sampler] [num_trainer] [cache_pct]
"""
import faulthandler
import multiprocessing as mp
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

SEED = 0x5A4D47
NUM_NODE, NUM_EDGE, DIM, NUM_CLASS, NUM_TRAIN = 20000, 300000, 16, 47, int(os.environ.get("FGNN_TEST_NUM_TRAIN", "2000"))
BATCH, NUM_EPOCH = 256, int(os.environ.get("FGNN_TEST_NUM_EPOCH", "2"))
# FGNN_TEST_CACHE_POLICY=static: the kCacheByPreSampleStatic policy (whole neighbourhoods instead of sampled ones);
# degree | random | heuristic | degree_hop | fake_optimal: the file-backed rankings (engine.cc:216-256), the file written
# by tools/dataset/fgnn_dataset before the engine starts
CACHE_POLICY = os.environ.get("FGNN_TEST_CACHE_POLICY", "")
STATIC_PRESAMPLE = CACHE_POLICY == "static"
FILE_POLICY = CACHE_POLICY if CACHE_POLICY in ("degree", "random", "heuristic", "degree_hop", "fake_optimal") else None
DATASET_TOOL = os.path.join(ROOT, "tools", "dataset", "fgnn_dataset")
# where the arch5 workers run: both on cuda:0 by default; FGNN_TEST_TRAINER_DEVICE=cuda:1 puts the trainers on a second
# GPU (the hand-off then crosses xGMI: the sampler's HBM ring is mapped by hipIpcOpenMemHandle and read peer to peer)
SAMPLER_DEV = os.environ.get("FGNN_TEST_SAMPLER_DEVICE", "cuda:0")
TRAINER_DEV = os.environ.get("FGNN_TEST_TRAINER_DEVICE", "cuda:0")


def dataset(workdir, sample_type):
    from fgnn_hip import synth
    d = synth.write_dataset(workdir, "synth", NUM_NODE, NUM_EDGE, DIM, NUM_CLASS, NUM_TRAIN, 100, 100, seed=17,
                            with_prefix=(sample_type == "weighted_khop_prefix"),
                            with_alias=(sample_type in ("weighted_khop", "weighted_khop_hash_dedup")))
    if os.environ.get("FGNN_TEST_DUP_SEED"):  # a corrupt train set: one id twice (SAMGRAPH_SANITY_CHECK must trip)
        t = np.fromfile(os.path.join(d, "train_set.bin"), dtype=np.uint32)
        t[-1] = t[0]
        t.tofile(os.path.join(d, "train_set.bin"))
    if FILE_POLICY:
        import subprocess
        if not os.path.exists(DATASET_TOOL) or os.path.getmtime(DATASET_TOOL) < os.path.getmtime(DATASET_TOOL + ".cc"):
            subprocess.run(["g++", "-O2", "-std=c++17", "-fopenmp", "-o", DATASET_TOOL, DATASET_TOOL + ".cc"], check=True)
        extra = ["5", "3", "16"] if FILE_POLICY == "fake_optimal" else []  # the runner's fanout, 16 train nodes at a time
        subprocess.run([DATASET_TOOL, "cache-by-" + FILE_POLICY.replace("_", "-"), d] + extra, check=True,
                       stdout=subprocess.DEVNULL)
    return d


def base_config(path, arch, sample_type):
    import samgraph.common as sc
    policy = sc.kCacheByPreSampleStatic if STATIC_PRESAMPLE else sc.kCacheByPreSample
    if FILE_POLICY:
        policy = sc.cache_policies[FILE_POLICY]
    cfg = dict(dataset_path=path, _arch=arch, _sample_type=sc.sample_types[sample_type], batch_size=BATCH,
               num_epoch=NUM_EPOCH, _cache_policy=policy, cache_percentage=0.0, max_sampling_jobs=10,
               max_copying_jobs=2, omp_thread_num=8, seed=SEED, presample_epoch=1, barriered_epoch=0)
    if sample_type == "random_walk":
        cfg.update(random_walk_length=3, random_walk_restart_prob=0.5, num_random_walk=4, num_neighbor=5, num_layer=3)
    else:
        fan = [5, 3] if sample_type != "weighted_khop_prefix" else [3, 4, 2]
        cfg.update(num_fanout=len(fan), fanout=fan)
    return cfg


class OracleReplay:
    """Replays what sampler `worker` of `num_sampler` does, batch by batch, with the oracle."""

    def __init__(self, path, sample_type, worker=0, num_sampler=1, presample=False, aligned=False):
        import oracle_py as oracle
        self.o = oracle
        self.worker, self.num_sampler, self.aligned = worker, num_sampler, aligned
        self.indptr = np.fromfile(os.path.join(path, "indptr.bin"), dtype=np.uint32)
        self.indices = np.fromfile(os.path.join(path, "indices.bin"), dtype=np.uint32)
        self.feat = np.fromfile(os.path.join(path, "feat.bin"), dtype=np.float32).reshape(NUM_NODE, DIM)
        self.label = np.fromfile(os.path.join(path, "label.bin"), dtype=np.int64)
        self.train0 = np.fromfile(os.path.join(path, "train_set.bin"), dtype=np.uint32)
        self.prefix = (np.fromfile(os.path.join(path, "prob_prefix_table.bin"), dtype=np.float32)
                       if sample_type == "weighted_khop_prefix" else None)
        self.st = dict(khop0=oracle.KHOP0, khop1=oracle.KHOP1, khop2=oracle.KHOP2, weighted_khop=oracle.WEIGHTED_KHOP,
                       weighted_khop_hash_dedup=oracle.WEIGHTED_KHOP_HASH_DEDUP,
                       weighted_khop_prefix=oracle.WEIGHTED_KHOP_PREFIX, random_walk=oracle.RANDOM_WALK)[sample_type]
        self.kw = {}
        if sample_type == "random_walk":
            self.fan = [5, 5, 5]
            self.kw = dict(walk_len=3, num_walks=4, num_neighbor=5, restart_prob=0.5)
        elif sample_type == "weighted_khop_prefix":
            self.fan = [3, 4, 2]
            self.kw = dict(prob_prefix=self.prefix)
        elif sample_type in ("weighted_khop", "weighted_khop_hash_dedup"):
            self.fan = [5, 3]
            self.kw = dict(prob_prefix=np.fromfile(os.path.join(path, "prob_table.bin"), dtype=np.float32),
                           alias_table=np.fromfile(os.path.join(path, "alias_table.bin"), dtype=np.uint32))
        else:
            self.fan = [5, 3]
        self.rng = oracle.make_rng(oracle.RNG_PHILOX, SEED)
        self.ht = oracle.HashTable(NUM_NODE, oracle.predict_num_nodes(BATCH, self.fan))
        self.plain_step = (NUM_TRAIN + BATCH - 1) // BATCH  # the presampler always walks the whole set
        if aligned:  # arch6 / arch7: DistAlignedShuffler
            self.part = oracle.aligned_shuffler_partition(NUM_TRAIN, BATCH, worker, num_sampler)
            self.num_step = self.part["epoch_step"]
        else:
            self.part = oracle.dist_shuffler_partition(NUM_TRAIN, BATCH, worker, num_sampler)
            self.num_step = self.plain_step
        self.rank = None
        if presample and FILE_POLICY:
            # the ranking the engine must have loaded (engine.cc:216-256): the file, as the tool wrote it
            self.rank = np.fromfile(os.path.join(path, "cache_by_%s.bin" % FILE_POLICY), dtype=np.uint32)
            assert sorted(self.rank.tolist()) == list(range(NUM_NODE))
        elif presample and worker == 0:
            self._presample()

    def _sample(self, seeds, key):
        return self.o.do_sample(self.indptr, self.indices, seeds, self.fan, self.st, self.rng, key, self.ht, **self.kw)

    def _presample(self):
        data = self.train0.copy()
        freq = np.zeros(NUM_NODE, dtype=np.uint32)
        data = self.o.shuffle_minstd0(data, 0)
        for step in range(self.plain_step):
            seeds = data[step * BATCH:(step + 1) * BATCH]
            if STATIC_PRESAMPLE:  # DoGPUSampleAllNeighbour: no draw, no khop2 row mutation
                nodes = self.o.sample_all_neighbour(self.indptr, self.indices, seeds, len(self.fan))
            else:
                nodes = self._sample(seeds, (1 << 63) | step)["input_nodes"]
            np.add.at(freq, nodes, 1)
        self.rank = self.o.presample_rank(freq)

    def epochs(self):
        """yields (key, seeds, task) for this sampler's batches in order"""
        if self.aligned:
            for epoch, step, seeds in self.o.aligned_shuffler_batches(self.train0, BATCH, self.worker,
                                                                      self.num_sampler, NUM_EPOCH):
                key = epoch * self.num_step + step
                yield key, seeds, self._sample(seeds, key)
            return
        data = self.train0.copy()
        for epoch in range(NUM_EPOCH):
            data = self.o.shuffle_minstd0(data, epoch)
            first = self.part["dataset_offset"] // BATCH
            for ls in range(self.part["num_local_step"]):
                step = first + ls
                seeds = data[step * BATCH:min(NUM_TRAIN, (step + 1) * BATCH)]
                key = epoch * self.num_step + step
                yield key, seeds, self._sample(seeds, key)


def check_batch(sam, key, seeds, task, rep, what="", with_feat=True):
    import torch  # noqa: F401
    nl = len(task["graphs"])
    blocks, feat, label = (sam.get_dgl_blocks_with_weights if rep.st == rep.o.RANDOM_WALK else sam.get_dgl_blocks)(
        key, nl, with_feat)

    def u32(t):
        return t.cpu().numpy().view(np.uint32)

    for li in range(nl):
        g = task["graphs"][li]
        assert sam.get_graph_num_src(key, li) == g["num_src"], what
        assert sam.get_graph_num_dst(key, li) == g["num_dst"], what
        assert sam.get_graph_num_edge(key, li) == g["num_edge"], what
        np.testing.assert_array_equal(u32(blocks[li].row), g["row"], err_msg=what)
        np.testing.assert_array_equal(u32(blocks[li].col), g["col"], err_msg=what)
        if g["data"] is not None:
            np.testing.assert_array_equal(u32(blocks[li].edata["weights"]), g["data"], err_msg=what)
    if not with_feat:
        # arch7: the script gathers features itself from the host tensors (sgnn_dgl/train_graphsage.py:165-167)
        assert feat is None and label is None
        feat, label = sam.load_subtensor(key, sam.get_dataset_feat(), sam.get_dataset_label(), "cuda:0")
    if os.environ.get("SAMGRAPH_EMPTY_FEAT", "0") in ("", "0"):
        assert feat.cpu().numpy().tobytes() == rep.feat[task["input_nodes"]].tobytes(), what + " feat"
    else:  # mock extraction from an uninitialised 2^k-row table: only the shape is defined
        assert tuple(feat.shape) == (len(task["input_nodes"]), DIM), what + " feat shape"
    np.testing.assert_array_equal(label.cpu().numpy(), rep.label[seeds], err_msg=what)
    np.testing.assert_array_equal(u32(sam.get_graph_output_nodes(key)), seeds, err_msg=what)
    inp = sam.get_graph_input_nodes(key)
    if inp.numel():
        np.testing.assert_array_equal(u32(inp), task["input_nodes"], err_msg=what)
    # the reference's own getter names (the pybind module of adapter.cc:48-192, `from samgraph.torch import c_lib`):
    # same memory as the ctypes path above, no copy
    from samgraph.torch import c_lib
    import torch
    for li in range(nl):
        for name, mine in (("row", sam.get_graph_row), ("col", sam.get_graph_col)):
            a, b = getattr(c_lib, "samgraph_torch_get_graph_" + name)(key, li), mine(key, li)
            assert a.dtype == torch.int32 and a.device == b.device and a.shape == b.shape, (what, name)
            assert a.numel() == 0 or (a.data_ptr() == b.data_ptr() and torch.equal(a, b)), (what, name)
        if task["graphs"][li]["data"] is not None:
            np.testing.assert_array_equal(u32(c_lib.samgraph_torch_get_graph_data(key, li)), task["graphs"][li]["data"])
    np.testing.assert_array_equal(u32(c_lib.samgraph_torch_get_graph_output_nodes(key)), seeds, err_msg=what)
    if inp.numel():
        assert c_lib.samgraph_torch_get_graph_input_nodes(key).data_ptr() == inp.data_ptr()
    if with_feat:
        f2, l2 = c_lib.samgraph_torch_get_graph_feat(key), c_lib.samgraph_torch_get_graph_label(key)
        assert f2.dtype == torch.float32 and tuple(f2.shape) == tuple(feat.shape) and f2.data_ptr() == feat.data_ptr()
        assert l2.dtype == torch.int64 and torch.equal(l2, label)
    else:
        df, dl = c_lib.samgraph_torch_get_dataset_feat(), c_lib.samgraph_torch_get_dataset_label()
        assert df.device.type == "cpu" and tuple(df.shape) == (NUM_NODE, DIM) and dl.dtype == torch.int64
        assert df.data_ptr() == sam.get_dataset_feat().data_ptr() and torch.equal(dl, sam.get_dataset_label())


def run_arch1(sample_type, workdir):
    path = dataset(workdir, sample_type)
    import samgraph.torch as sam
    cfg = base_config(path, sam.kArch1, sample_type)
    cfg.update(sampler_ctx="cuda:0", trainer_ctx="cuda:0")
    sam.config(cfg)
    sam.init()
    assert sam.num_class() == NUM_CLASS and sam.feat_dim() == DIM and sam.num_epoch() == NUM_EPOCH
    rep = OracleReplay(path, sample_type)
    assert sam.steps_per_epoch() == rep.num_step
    n = 0
    for key, seeds, task in rep.epochs():
        sam.sample_once()
        got = sam.get_next_batch()
        assert got == key, (got, key)
        check_batch(sam, key, seeds, task, rep, "arch1 key %d" % key)
        epoch, step = key // rep.num_step, key % rep.num_step
        assert sam.get_log_step_value(epoch, step, sam.kLogL1NumSample) == task["total_edges"]
        assert sam.get_log_step_value(epoch, step, sam.kLogL1NumNode) == len(task["input_nodes"])
        n += 1
    assert n == NUM_EPOCH * rep.num_step
    sam.report_step_average(NUM_EPOCH - 1, rep.num_step - 1)
    sam.shutdown()
    print("arch1 %s ok: %d batches" % (sample_type, n))


def run_inproc(arch, sample_type, workdir, cache_pct, threaded):
    """arch2 / arch3 / arch4: one process samples and extracts (the reference's default single-process scripts use
    arch3); `threaded` = samgraph_start's background threads instead of sample_once per step."""
    path = dataset(workdir, sample_type)
    import samgraph.torch as sam
    cfg = base_config(path, {"arch2": sam.kArch2, "arch3": sam.kArch3, "arch4": sam.kArch4}[arch], sample_type)
    cfg.update(sampler_ctx="cuda:0", trainer_ctx="cuda:0", cache_percentage=cache_pct)
    sam.config(cfg)
    sam.init()
    rep = OracleReplay(path, sample_type, 0, 1, cache_pct > 0)
    assert sam.steps_per_epoch() == rep.num_step and sam.num_epoch() == NUM_EPOCH
    if threaded:
        sam.start()
    n = miss_total = 0
    for key, seeds, task in rep.epochs():
        if not threaded:
            sam.sample_once()
        got = sam.get_next_batch()
        assert got == key, (got, key)
        check_batch(sam, key, seeds, task, rep, "%s key %d" % (arch, key))
        if cache_pct > 0:
            # the ranking itself is observable through the miss volume of every batch (kLogL1MissBytes,
            # cuda_loops.cc:1098-1101): rows of nodes outside the first num_cached entries of the rank list
            cached = np.zeros(NUM_NODE, dtype=bool)
            cached[rep.rank[:int(NUM_NODE * cache_pct)]] = True
            misses = int((~cached[task["input_nodes"]]).sum())
            epoch, step = key // rep.num_step, key % rep.num_step
            assert sam.get_log_step_value(epoch, step, sam.kLogL1MissBytes) == misses * DIM * 4, (key, misses)
            miss_total += misses
        n += 1
    assert n == NUM_EPOCH * rep.num_step
    sam.report_step_average(NUM_EPOCH - 1, rep.num_step - 1)
    sam.shutdown()
    print("%s %s cache %.2f %s%s ok: %d batches, %d miss rows" % (arch, sample_type, cache_pct,
                                                                "threads" if threaded else "inline",
                                                                " static-presample" if STATIC_PRESAMPLE else
                                                                " policy-%s" % FILE_POLICY if FILE_POLICY else "", n,
                                                                miss_total))


def run_inproc_early_stop(arch, sample_type, workdir, cache_pct, take):
    """samgraph_start's threads stopped long before their last batch (the reference's loops poll ShouldShutdown,
    cuda_loops_arch3.cc:178-196): by then the sampler thread is blocked on a full queue and the copy thread holds
    batches nobody will ask for -- shutdown must come back, and the batches taken until then are still exact."""
    path = dataset(workdir, sample_type)
    import samgraph.torch as sam
    cfg = base_config(path, {"arch2": sam.kArch2, "arch3": sam.kArch3, "arch4": sam.kArch4}[arch], sample_type)
    cfg.update(sampler_ctx="cuda:0", trainer_ctx="cuda:0", cache_percentage=cache_pct, num_epoch=40)
    sam.config(cfg)
    sam.init()
    rep = OracleReplay(path, sample_type, 0, 1, cache_pct > 0)
    sam.start()
    n = 0
    for key, seeds, task in rep.epochs():
        got = sam.get_next_batch()
        assert got == key, (got, key)
        check_batch(sam, key, seeds, task, rep, "%s key %d" % (arch, key))
        n += 1
        if n == take:
            break
    time.sleep(0.3)  # the threads run ahead until the queue and the graph pool are full
    t0 = time.time()
    sam.shutdown()
    assert time.time() - t0 < 20.0
    print("%s %s early stop ok: %d of %d batches taken, shutdown in %.2f s" % (arch, sample_type, n, 40 * rep.num_step,
                                                                               time.time() - t0))


def run_dynamic(sample_type, workdir, threaded):
    """arch4 with `_cache_policy = dynamic_cache` (cuda_loops_arch4.cc:56-97,136-187): input nodes are the whole
    neighbourhood of the layer-1 frontier, hits are rows of the previous batch's feature tensor."""
    path = dataset(workdir, sample_type)
    import samgraph.common as sc
    import samgraph.torch as sam
    cfg = base_config(path, sam.kArch4, sample_type)
    cfg.update(sampler_ctx="cuda:0", trainer_ctx="cuda:0", _cache_policy=sc.kDynamicCache, cache_percentage=0.0)
    sam.config(cfg)
    sam.init()
    rep = OracleReplay(path, sample_type)
    rep.ht = rep.o.HashTable(NUM_NODE, NUM_NODE)
    kw = {}
    if sample_type == "weighted_khop":
        kw = dict(prob=rep.kw["prob_prefix"], alias=rep.kw["alias_table"])
    rep._sample = lambda seeds, key: rep.o.do_sample_dycache(rep.indptr, rep.indices, seeds, rep.fan, rep.st, rep.rng,
                                                             key, rep.ht, **kw)
    assert sam.steps_per_epoch() == rep.num_step
    if threaded:
        sam.start()
    n = hits = 0
    prev = np.empty(0, dtype=np.uint32)
    for key, seeds, task in rep.epochs():
        if not threaded:
            sam.sample_once()
        got = sam.get_next_batch()
        assert got == key, (got, key)
        check_batch(sam, key, seeds, task, rep, "dynamic key %d" % key)
        # the cache of this batch was the previous batch's node list (ReplaceCacheGPU, cuda_loops.cc:1261-1265)
        misses = int((~np.isin(task["input_nodes"], prev)).sum())
        epoch, step = key // rep.num_step, key % rep.num_step
        assert sam.get_log_step_value(epoch, step, sam.kLogL1MissBytes) == misses * DIM * 4, (key, misses)
        hits += len(task["input_nodes"]) - misses
        prev = task["input_nodes"]
        assert len(prev) > sum(g["num_edge"] for g in task["graphs"][1:])  # whole neighbourhoods, not samples
        n += 1
    assert n == NUM_EPOCH * rep.num_step and hits > 0
    sam.shutdown()
    print("dynamic-cache %s %s ok: %d batches, %d rows served by the previous batch" % (
        sample_type, "threads" if threaded else "inline", n, hits))


def _join_all(procs, roles, limit=600.0):
    """Waits for all children; as soon as one dies with an error (or the time is up) the rest are terminated, so a
    failure is reported at once instead of after the survivors' time-outs."""
    import time
    t0 = time.time()
    bad = 0
    while any(p.is_alive() for p in procs):
        for i, p in enumerate(procs):
            if not p.is_alive() and p.exitcode not in (0, None):
                bad = 1
        if bad or time.time() - t0 > limit:
            bad = 1
            break
        time.sleep(0.05)
    for i, p in enumerate(procs):
        if p.is_alive():
            print("process %d (%s) did not finish" % (i, roles[i]), file=sys.stderr)
            p.terminate()
        elif p.exitcode != 0:
            print("process %d (%s) exited with %s" % (i, roles[i], p.exitcode), file=sys.stderr)
            bad = 1
    for p in procs:
        p.join(timeout=10)
    return bad


def _sampler_proc(worker, num_sampler, barrier, err):
    try:
        faulthandler.dump_traceback_later(400, exit=True)  # a stuck child shows where it is stuck
        import samgraph.torch as sam
        sam.sample_init(worker, SAMPLER_DEV)
        barrier.wait()
        num_step = sam.steps_per_epoch()
        local = num_step - (num_step // num_sampler) * worker if worker == num_sampler - 1 else num_step // num_sampler
        assert sam.num_local_step() == local
        for _ in range(NUM_EPOCH):
            for _ in range(local):
                sam.sample_once()
        # tests: a script whose sampler loop ends and then WAITS (an epoch barrier) without calling anything -- the last
        # batch's tail must be finished and published by the engine itself (publisher thread), not by the next call
        linger = float(os.environ.get("FGNN_TEST_SAMPLER_LINGER", "0"))
        if linger:
            import time
            time.sleep(linger)
        sam.shutdown()
    except BaseException:
        traceback.print_exc()
        err.value = 1
        os._exit(1)


def _trainer_proc(worker, num_trainer, num_sampler, path, sample_type, presample, pipeline, barrier, err, cache_pct=0.0):
    try:
        faulthandler.dump_traceback_later(400, exit=True)  # a stuck child shows where it is stuck
        import samgraph.torch as sam
        barrier.wait()  # samplers (and the presample) are done initialising
        sam.train_init(worker, TRAINER_DEV)
        hold = os.environ.get("FGNN_TEST_HOLD_TRAINER")  # tests: this trainer starts receiving when the file appears
        while hold and not os.path.exists(hold):
            import time
            time.sleep(0.01)
        # expected batches of every sampler, by key
        expected = {}
        rank = None
        for w in range(num_sampler):
            rep = OracleReplay(path, sample_type, w, num_sampler, presample)
            if w == 0:
                rank = rep.rank
            for key, seeds, task in rep.epochs():
                expected[key] = (seeds, task, rep)
        total = len(expected)
        mine = total // num_trainer + (1 if worker < total % num_trainer else 0)
        if pipeline:
            sam.extract_start(mine)
        seen = miss_total = 0
        import time
        max_wait = 0.0
        for _ in range(mine):
            if not pipeline:
                sam.sample_once()
            t_get = time.time()
            key = sam.get_next_batch()
            max_wait = max(max_wait, time.time() - t_get)
            seeds, task, rep = expected.pop(key)
            check_batch(sam, key, seeds, task, rep, "arch5 key %d" % key)
            if presample:
                # which rows were misses is the ranking made observable (kLogL1MissBytes, dist_loops.cc:830-835): the
                # sampler split the batch against table[rank[i]] = i, i < num_cached (dist_engine.cc:193-229) and this
                # trainer fetched exactly the rows outside the first num_cached entries from host memory
                cached = np.zeros(NUM_NODE, dtype=bool)
                cached[rank[:int(NUM_NODE * cache_pct)]] = True
                misses = int((~cached[task["input_nodes"]]).sum())
                epoch, step = key // rep.plain_step, key % rep.plain_step
                got = sam.get_log_step_value(epoch, step, sam.kLogL1MissBytes)
                assert got == misses * DIM * 4, (key, got, misses)
                miss_total += misses
            seen += 1
        if num_trainer == 1:
            assert not expected
        sam.shutdown()
        print("trainer %d checked %d batches, %d miss rows (rank head %s), longest wait for a batch %.3f s"
              % (worker, seen, miss_total, None if rank is None else rank[:4], max_wait))
    except BaseException:
        traceback.print_exc()
        err.value = 1
        os._exit(1)


def run_arch5(sample_type, workdir, num_sampler, num_trainer, cache_pct, pipeline=True):
    path = dataset(workdir, sample_type)
    import samgraph.torch as sam
    cfg = base_config(path, sam.kArch5, sample_type)
    cfg.update(num_sample_worker=num_sampler, num_train_worker=num_trainer, cache_percentage=cache_pct)
    sam.config(cfg)
    sam.data_init()  # before fork, no GPU touched
    ctx = mp.get_context("fork")
    barrier = ctx.Barrier(num_sampler + num_trainer)
    err = ctx.Value("i", 0)
    procs = [ctx.Process(target=_sampler_proc, args=(w, num_sampler, barrier, err)) for w in range(num_sampler)]
    procs += [ctx.Process(target=_trainer_proc,
                          args=(w, num_trainer, num_sampler, path, sample_type, cache_pct > 0, pipeline, barrier, err,
                                cache_pct))
              for w in range(num_trainer)]
    for p in procs:
        p.start()
    bad = _join_all(procs, ["sampler"] * num_sampler + ["trainer"] * num_trainer)
    # hand-off statistics (include/samgraph_ext.h): the counters live in the shared queue region this parent created
    for w in range(num_sampler):
        print("ring %d stats %s" % (w, sam.ext_queue_stats(w)))
    if bad or err.value:
        sys.exit(1)
    print("arch5 %s %dS+%dT cache %.2f%s ok" % (sample_type, num_sampler, num_trainer, cache_pct,
                                                " policy-%s" % FILE_POLICY if FILE_POLICY else ""))


class _FileBarrier:
    """barrier between processes that share nothing but a directory (the torchrun launch style)"""

    def __init__(self, workdir, me, n):
        self.dir, self.me, self.n, self.round = workdir, me, n, 0

    def wait(self, limit=300.0):
        import time
        tag = "barrier%d." % self.round
        self.round += 1
        open(os.path.join(self.dir, tag + self.me), "w").close()
        t0 = time.time()
        while len([f for f in os.listdir(self.dir) if f.startswith(tag)]) < self.n:
            if time.time() - t0 > limit:
                raise RuntimeError("barrier %s: only %s arrived" % (tag, os.listdir(self.dir)))
            time.sleep(0.01)


class _Err:
    value = 0


def run_arch5_named_role(sample_type, workdir, role, idx, num_sampler, num_trainer, cache_pct):
    """One worker of an arch5 job whose processes have NO common forking parent (one process per GPU started by a
    launcher such as torchrun): every process runs config + data_init itself and meets the others in the named shared
    regions of SAMGRAPH_SHM_PREFIX (set by the test, which also wrote the dataset)."""
    import samgraph.torch as sam
    path = os.path.join(workdir, "synth")
    cfg = base_config(path, sam.kArch5, sample_type)
    cfg.update(num_sample_worker=num_sampler, num_train_worker=num_trainer, cache_percentage=cache_pct)
    sam.config(cfg)
    sam.data_init()
    barrier = _FileBarrier(workdir, "%s%d" % (role, idx), num_sampler + num_trainer)
    barrier.wait()  # every process has attached to every region
    if role == "s":
        _sampler_proc(idx, num_sampler, barrier, _Err())
    else:
        _trainer_proc(idx, num_trainer, num_sampler, path, sample_type, cache_pct > 0, True, barrier, _Err(), cache_pct)
    print("arch5-named %s%d ok" % (role, idx))


def _sgnn_worker(arch, worker, num_worker, path, sample_type, cache_pct, background, barrier, err):
    """One worker of the reference's SGNN baselines: arch6 = example/samgraph/sgnn/train_graphsage.py:115-190
    (sample_init + train_init in the same process, sample_once / get_next_batch per step), arch7 =
    example/samgraph/sgnn_dgl/train_graphsage.py:95-170 (config + init per worker, blocks without features)."""
    try:
        faulthandler.dump_traceback_later(400, exit=True)
        import samgraph.torch as sam
        if arch == "arch6":
            sam.sample_init(worker, "cuda:0")
            sam.train_init(worker, "cuda:0")
        else:
            cfg = base_config(path, sam.kArch7, sample_type)
            cfg.update(worker_id=worker, num_worker=num_worker, sampler_ctx="cuda:0", trainer_ctx="cuda:0")
            sam.config(cfg)
            sam.init()
        rep = OracleReplay(path, sample_type, worker, num_worker, cache_pct > 0, aligned=True)
        assert sam.steps_per_epoch() == rep.num_step and sam.num_local_step() == rep.part["num_local_step"]
        assert sam.num_epoch() == NUM_EPOCH
        barrier.wait()
        if background:
            sam.extract_start(0)
        n = 0
        for key, seeds, task in rep.epochs():
            if not background:
                sam.sample_once()
            got = sam.get_next_batch()
            assert got == key, (got, key)
            check_batch(sam, key, seeds, task, rep, "%s worker %d key %d" % (arch, worker, key),
                        with_feat=(arch == "arch6"))
            n += 1
        assert n == NUM_EPOCH * rep.part["num_local_step"]
        barrier.wait()
        sam.shutdown()
        print("%s worker %d/%d checked %d batches" % (arch, worker, num_worker, n))
    except BaseException:
        traceback.print_exc()
        err.value = 1
        os._exit(1)


def run_sgnn(arch, sample_type, workdir, num_worker, cache_pct, background):
    path = dataset(workdir, sample_type)
    if arch == "arch6":
        import samgraph.torch as sam
        cfg = base_config(path, sam.kArch6, sample_type)
        cfg.update(num_worker=num_worker, cache_percentage=cache_pct)
        sam.config(cfg)
        sam.data_init()  # before fork, no GPU touched (sgnn/train_graphsage.py: run_init)
    ctx = mp.get_context("fork")
    barrier = ctx.Barrier(num_worker)
    err = ctx.Value("i", 0)
    procs = [ctx.Process(target=_sgnn_worker,
                         args=(arch, w, num_worker, path, sample_type, cache_pct, background, barrier, err))
             for w in range(num_worker)]
    for p in procs:
        p.start()
    bad = _join_all(procs, ["worker"] * num_worker)
    if bad or err.value:
        sys.exit(1)
    print("%s %s %d workers cache %.2f ok" % (arch, sample_type, num_worker, cache_pct))


def _switch_sampler_proc(barrier, sem, stop, err):
    try:
        faulthandler.dump_traceback_later(400, exit=True)  # a stuck child shows where it is stuck
        import samgraph.torch as sam
        sam.sample_init(0, "cuda:0")
        barrier.wait()
        for _ in range(NUM_EPOCH):
            for _ in range(sam.num_local_step()):
                sam.sample_once()
                sem.release()  # one permit per published batch (balance_switcher/train_pinsage.py: mq_sem)
        stop.set()
        sam.shutdown()
    except BaseException:
        traceback.print_exc()
        err.value = 1
        os._exit(1)


def _switch_consumer_proc(is_switcher, path, sample_type, barrier, sem, stop, seen_keys, err):
    try:
        faulthandler.dump_traceback_later(400, exit=True)  # a stuck child shows where it is stuck
        import samgraph.torch as sam
        barrier.wait()
        if is_switcher:
            # the sampler GPU turns into a trainer once its sampling is done (dist_engine.cc:425-431)
            sam.switch_init(0, "cuda:0", 0.1)
            stop.wait()
        else:
            sam.train_init(0, "cuda:0")
        rep = OracleReplay(path, sample_type, 0, 1, True)
        expected = {key: (seeds, task) for key, seeds, task in rep.epochs()}
        n = 0
        while sem.acquire(timeout=2.0 if (is_switcher or stop.is_set()) else 20.0):
            sam.sample_once()
            key = sam.get_next_batch()
            seeds, task = expected[key]
            check_batch(sam, key, seeds, task, rep, "%s key %d" % ("switcher" if is_switcher else "trainer", key))
            with seen_keys.get_lock():
                seen_keys[key] += 1
            n += 1
        sam.shutdown()
        print("%s checked %d batches" % ("switcher" if is_switcher else "trainer", n))
    except BaseException:
        traceback.print_exc()
        err.value = 1
        os._exit(1)


def run_arch5_switcher(sample_type, workdir):
    """BASELINE config 5's control flow in miniature: 1 sampler + 1 trainer + the sampler's switcher, `have_switcher`
    on (input nodes are shipped, every consumer splits hits / misses against its own cache, task_queue.cc:93-95)."""
    path = dataset(workdir, sample_type)
    import samgraph.torch as sam
    cfg = base_config(path, sam.kArch5, sample_type)
    cfg.update(num_sample_worker=1, num_train_worker=1, cache_percentage=0.2, have_switcher=1)
    sam.config(cfg)
    sam.data_init()
    ctx = mp.get_context("fork")
    barrier = ctx.Barrier(3)
    sem, stop, err = ctx.Semaphore(0), ctx.Event(), ctx.Value("i", 0)
    total = NUM_EPOCH * ((NUM_TRAIN + BATCH - 1) // BATCH)
    seen = ctx.Array("i", total)
    procs = [ctx.Process(target=_switch_sampler_proc, args=(barrier, sem, stop, err)),
             ctx.Process(target=_switch_consumer_proc, args=(False, path, sample_type, barrier, sem, stop, seen, err)),
             ctx.Process(target=_switch_consumer_proc, args=(True, path, sample_type, barrier, sem, stop, seen, err))]
    for p in procs:
        p.start()
    bad = _join_all(procs, ["sampler", "trainer", "switcher"])
    if bad or err.value or list(seen) != [1] * total:
        print("seen", list(seen))
        sys.exit(1)
    print("arch5 switcher %s ok: %d batches, each consumed once" % (sample_type, total))


if __name__ == "__main__":
    mode, st, wd = sys.argv[1:4]
    if mode == "dataset":
        dataset(wd, st)
    elif mode == "arch5_named":
        run_arch5_named_role(st, wd, sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]),
                             float(sys.argv[8]))
    elif mode == "arch1":
        run_arch1(st, wd)
    elif mode == "switcher":
        run_arch5_switcher(st, wd)
    elif mode in ("arch6", "arch7"):
        run_sgnn(mode, st, wd, int(sys.argv[4]), float(sys.argv[5]), len(sys.argv) > 6 and sys.argv[6] == "background")
    elif mode == "dynamic":
        run_dynamic(st, wd, len(sys.argv) > 4 and sys.argv[4] == "threads")
    elif mode in ("arch2", "arch3", "arch4") and sys.argv[5] == "early_stop":
        run_inproc_early_stop(mode, st, wd, float(sys.argv[4]), int(sys.argv[6]))
    elif mode in ("arch2", "arch3", "arch4"):
        run_inproc(mode, st, wd, float(sys.argv[4]), sys.argv[5] == "threads")
    else:
        run_arch5(st, wd, int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6]),
                  pipeline=(len(sys.argv) < 8 or sys.argv[7] == "pipeline"))
