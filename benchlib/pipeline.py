"""benchlib.pipeline -- bench.py at N >= 2: the reference's factored pipeline, one process per GPU through samgraph.torch /
c_lib.so (arch5): roles (chosen from measured rates), spans and windows, the backends (engine / GPU-less rehearsal),
the link self-test, the like-for-like N = 1 point and the calibration children."""
import json
import os
import subprocess
import sys
import time

from .common import (  # noqa: F401
    sampler_config_keys,
    BENCH_PY, HBM_PEAK_GBS, HOST_LINK_GBS, QUEUE_SLOTS, ROOT, WORKLOADS, XGMI_LINK_GBS,
    gpu_numa_node, local_step_range, no_gc, numa_nodes_with_memory, torch, write_dataset)


def default_samplers(n_gpus):
    """1S+1T at 2 GPUs, 2S+6T at 8 (exp/table4/run.py:329-330 for GraphSAGE / papers100M); one sampler below 8"""
    return max(1, n_gpus // 4)


def choose_samplers(world, t_sampler_ms, t_trainer_ms):
    """S of 1 .. world-1 minimising the pipeline's time per batch max(t_s / S, t_t / (world - S)) for the measured
    per-process rates (a sampler process alone, a trainer process alone); ties go to fewer samplers.  The reference
    tunes S per workload by hand (exp/table4/README.md:79-90: 4S / 2S / 2S / 1S).  Returns (S, {S: predicted ms})."""
    pred = {s: max(t_sampler_ms / s, t_trainer_ms / (world - s)) for s in range(1, world)}
    best = min(pred, key=lambda s: (pred[s], s))
    return best, pred


class _FileBarrier:
    """barrier between processes that share nothing but a directory"""

    def __init__(self, d, me, n):
        self.dir, self.me, self.n, self.round = d, me, n, 0

    def wait(self, limit=600.0):
        tag = "cal%d." % self.round
        self.round += 1
        open(os.path.join(self.dir, tag + self.me), "w").close()
        t0 = time.time()
        while len([f for f in os.listdir(self.dir) if f.startswith(tag)]) < self.n:
            if time.time() - t0 > limit:
                raise RuntimeError("calibration barrier %s: only %s arrived" % (tag, sorted(os.listdir(self.dir))))
            time.sleep(0.002)


def run_calibrate_child():
    """--samplers auto: one role of a 1S+1T arch5 job of its own (named regions, the job's dataset) in a child process
    that rank 0 (sampler) / rank 1 (trainer) started before touching the GPU.  The sampler child fills the queue
    ALONE (nobody consumes: warm + K batches, fewer than the queue has slots), then the trainer child drains it ALONE
    -- each stage's own time per batch, the --decoupled measurement in miniature.  Request: one JSON line on stdin
    (an empty line: not needed); answer: one JSON line on stdout."""
    line = sys.stdin.readline()
    if not line.strip():
        return
    req = json.loads(line)
    for k, v in req["env"].items():
        os.environ[k] = v
    import samgraph.torch as sam
    torch.cuda.set_device(req["dev_id"])
    ctx = "cuda:%d" % req["dev_id"]
    warm, K = req["warm"], req["steps"]
    spe = req["steps_per_epoch"]
    cfg = dict(dataset_path=req["dir"], _arch=sam.kArch5, _sample_type=sam.sample_types[req["sample_type"]],
               batch_size=req["batch_size"], num_epoch=(warm + K + spe - 1) // spe + 1,
               _cache_policy=sam.cache_policies["pre_sample"], presample_epoch=req["presample_epochs"],
               cache_percentage=req["cache_ratio"], max_sampling_jobs=10, max_copying_jobs=2, omp_thread_num=8,
               num_sample_worker=1, num_train_worker=1, seed=req["seed"], **req["sampler_keys"])
    sam.config(cfg)
    sam.data_init()
    bar = _FileBarrier(req["sync_dir"], req["role"], 2)
    bar.wait()  # both children have attached to every region
    now = lambda: time.clock_gettime(time.CLOCK_MONOTONIC)  # noqa: E731
    if req["role"] == "s":
        sam.sample_init(0, ctx)  # pre-samples: the ranking the trainer's cache is built from
        bar.wait()
        bar.wait()  # the trainer has built its cache
        for _ in range(warm):
            sam.sample_once()
        sam.get_log_step_value((warm - 1) // spe, (warm - 1) % spe, sam.kLogL1NumSample)  # everything so far published
        with no_gc():
            t0 = now()
            for _ in range(K):
                sam.sample_once()
            last = warm + K - 1
            sam.get_log_step_value(last // spe, last % spe, sam.kLogL1NumSample)  # ... the K timed ones too
            ms = (now() - t0) / K * 1e3
        bar.wait()  # the queue holds warm + K batches
        bar.wait()  # drained
    else:
        bar.wait()
        sam.train_init(0, ctx)
        bar.wait()
        bar.wait()
        sam.extract_start(warm + K)
        for _ in range(warm):
            sam.get_next_batch()
        with no_gc():
            t0 = now()
            for _ in range(K):
                sam.get_next_batch()
            ms = (now() - t0) / K * 1e3
        bar.wait()
    print(json.dumps({"role": req["role"], "ms_per_batch": ms, "steps": K, "warm": warm}), flush=True)
    sam.shutdown()


def calibrate_roles(args, w, dist, rank, world, dev_id, job, steps_per_epoch, cal_child, warm=16, steps=96):
    """collective over the job's ranks: ranks 0 and 1 drive their calibration children (run_calibrate_child), rank 0
    chooses S from the two measured rates and every rank learns it.  A calibration that fails costs the choice, not the
    job: the reference's split is used and the line says why.  --rehearse: no GPU, the rates are --rehearse-rates."""
    mine = None
    if args.rehearse:
        rates = [float(x) for x in args.rehearse_rates.split(",")]
        mine = {"role": "s" if rank == 0 else "t", "ms_per_batch": rates[0] if rank == 0 else rates[1]} if rank < 2 else None
    elif cal_child is not None:
        cal_dev = job.get("dev0", dev_id)
        sync = os.path.join(job["dir"], "calibration_sync")
        os.makedirs(sync, exist_ok=True)
        req = {"env": {"SAMGRAPH_SHM_PREFIX": job["prefix"] + "_cal", "SAMGRAPH_SHM_KEEP": "1",
                       "SAMGRAPH_EMPTY_FEAT": str(args.empty_feat_bits),
                       "SAMGRAPH_LOG_LEVEL": os.environ.get("SAMGRAPH_LOG_LEVEL", "warn")},
               # both children on rank 0's GPU, one after the other (the sampler fills the queue, then the trainer
               # drains it): the same-device hand-off is the path every GPU test exercises; a calibration must not
               # be the first thing that ever runs across two devices
               "role": "s" if rank == 0 else "t", "dev_id": cal_dev, "dir": job["dir"], "sync_dir": sync, "warm": warm,
               "steps": steps, "steps_per_epoch": steps_per_epoch, "sample_type": args.sample_type,
               "batch_size": w["batch_size"], "cache_ratio": args.cache_ratio,
               "presample_epochs": max(1, args.presample_epochs), "fanout": w["fanout"], "seed": args.seed,
               "sampler_keys": sampler_config_keys(w, args.sample_type)}
        try:
            o, _ = cal_child.communicate((json.dumps(req) + "\n").encode(),
                                         timeout=float(os.environ.get("FGNN_BENCH_CAL_TIMEOUT", "240")))
            lines = [ln for ln in o.decode(errors="replace").splitlines() if ln.startswith("{")]
            mine = json.loads(lines[-1]) if lines else {"error": "calibration child: rc %s" % cal_child.returncode}
        except Exception as e:
            cal_child.kill()
            mine = {"error": "%s: %s" % (type(e).__name__, e)}
    got = [None] * world
    dist.all_gather_object(got, mine)
    if rank == 0 and not args.rehearse and os.path.isdir("/dev/shm"):
        for f in os.listdir("/dev/shm"):
            if f.startswith(job["prefix"] + "_cal"):
                try:
                    os.unlink(os.path.join("/dev/shm", f))
                except OSError:
                    pass
    ts = next((g["ms_per_batch"] for g in got if g and g.get("role") == "s" and "ms_per_batch" in g), None)
    tt = next((g["ms_per_batch"] for g in got if g and g.get("role") == "t" and "ms_per_batch" in g), None)
    ref = default_samplers(world)
    if ts is None or tt is None:  # (every rank sees the same list: the same decision everywhere)
        return ref, {"mode": "auto: calibration failed, the reference's split is used", "reference_split": ref,
                     "errors": [g.get("error") for g in got if g and "error" in g]}
    S, pred = choose_samplers(world, ts, tt)
    return S, {"mode": "auto", "sampler_ms_per_batch_alone": ts, "trainer_ms_per_batch_alone": tt,
               "predicted_ms_per_batch_by_samplers": {str(k): v for k, v in pred.items()}, "chosen": S,
               "reference_split": ref, "batches_timed": steps,
               "note": "a sampler process alone (queue filling, nobody consuming) and a trainer process alone (draining "
                       "it) as a 1S+1T job of their own on rank 0's GPU, one after the other, before the roles are given "
                       "out; S = argmin max(t_s / S, t_t / (N - S)); --samplers <n> overrides"}


def split_count(total, parts, index):
    """how many of `total` units part `index` of `parts` takes (the trainers' share, multi_gpu/train_graphsage.py:293-298)"""
    return total // parts + (1 if index < total % parts else 0)


def train_region_batches(steps, train_steps, trainers):
    """(warm-up, timed) batches of the region with a training step per batch.  The trainers all-reduce their gradients
    every step, so each of them must take the SAME number of batches -- a remainder would leave the trainers with one
    batch more waiting for the others' all-reduce forever: both counts are multiples of the trainer count (the
    reference pads the train set to equal shares for the same reason, dist_shuffler_aligned.cc:50-59)."""
    t = max(trainers, 1)
    timed = max(min(steps, train_steps) // t, 1) * t
    warm = max(min(8, timed) // t, 1) * t
    return warm, timed


def span_margins(warmup, trainers, decoupled=False):
    """(lead, tail) batches around the timed windows of a pipeline span: consumed, stamped, not counted.  The lead is the
    warm-up AND the queue's transient: the span starts on an empty queue, and where the trainers are the slower side
    (every default role split at cache 0.2) the steady state is a FULL queue -- reached after about
    slots x t_sample / (t_extract - t_sample) consumed batches, ~70 at 1S+1T and ~320 at 2S+6T; twice the queue's slots
    covers both (340 batches of 0.05-0.45 ms each).  On separate GPUs the fill level does not change the trainers' rate;
    where sampler and trainer SHARE one GPU (the development box) it does: 0.34 ms per batch while the queue fills,
    0.44 once it is full (profiles/r05_b_windows_transient.txt).  tail: two batches per trainer cover the spread of the
    trainers' finishing times."""
    if decoupled:
        return 1, 0
    return max(warmup, 2 * trainers, 2 * QUEUE_SLOTS), max(warmup, 2 * trainers)


def span_total(lead, windows, steps, tail, trainers, train):
    """batches of one span; with a training step per batch every trainer must take the same number (all-reduce)"""
    total = lead + windows * steps + tail
    return (total + trainers - 1) // trainers * trainers if train else total


def read_windows(stamps, lead, windows, steps):
    """stamps: [(t, key)] of every consumed batch of a span, any order.  Returns (merged, [(t_begin, t_end, keys)] per
    window): window j = the batches lead + j*steps .. lead + (j+1)*steps - 1 in consumption order, its clock runs from
    the stamp of the batch consumed just before it to the stamp of its last batch."""
    merged = sorted(stamps)
    out = []
    for j in range(windows):
        a = lead + j * steps
        out.append((merged[a - 1][0], merged[a + steps - 1][0], [k for _, k in merged[a:a + steps]]))
    return merged, out


def pipeline_roles(world, samplers=None):
    """(samplers, trainers); samplers: a count, or None / 0 / "auto" before the choice is made = the reference's split"""
    s = int(samplers) if samplers and samplers != "auto" else default_samplers(world)
    if not (0 < s < world):
        raise ValueError("need at least one sampler and one trainer: %d samplers of %d ranks" % (s, world))
    return s, world - s


class EngineBackend:
    """the product: arch5 through samgraph.torch / c_lib.so on this rank's GPU"""

    def __init__(self, args, w, job, S, T, is_sampler, idx, dev_id, num_epoch):
        import samgraph.torch as sam
        self.sam, self.w, self.is_sampler, self.idx, self.ctx = sam, w, is_sampler, idx, "cuda:%d" % dev_id
        self.dev_id = dev_id
        cfg = dict(dataset_path=job["dir"], _arch=sam.kArch5, _sample_type=sam.sample_types[args.sample_type],
                   batch_size=w["batch_size"], num_epoch=num_epoch, _cache_policy=sam.cache_policies["pre_sample"],
                   presample_epoch=max(1, args.presample_epochs), cache_percentage=args.cache_ratio,
                   max_sampling_jobs=10, max_copying_jobs=2, omp_thread_num=8, num_sample_worker=S, num_train_worker=T,
                   seed=args.seed, **sampler_config_keys(w, args.sample_type))
        self.weighted_blocks = args.sample_type == "random_walk"  # PinSAGE: visit counts as edge weights
        sam.config(cfg)
        sam.data_init()  # attaches to / creates the job's shared regions; no GPU touched

    def role_init(self):
        torch.cuda.set_device(self.dev_id)
        if self.is_sampler:
            self.sam.sample_init(self.idx, self.ctx)  # sampler 0 pre-samples; the others wait for it inside
        else:
            self.sam.train_init(self.idx, self.ctx)

    def num_local_step(self):
        return self.sam.num_local_step()

    def sample_once(self):
        self.sam.sample_once()

    def extract_start(self, n):
        self.sam.extract_start(n)

    def next_batch(self):
        return self.sam.get_next_batch()

    def blocks(self, key):
        get = self.sam.get_dgl_blocks_with_weights if self.weighted_blocks else self.sam.get_dgl_blocks
        return get(key, len(self.w["fanout"]))

    def sampler_stats(self, keys):
        sam = self.sam
        return {"edges": sum(sam.get_log_step_value(e, s, sam.kLogL1NumSample) for e, s in keys)}

    def trainer_stats(self, keys):
        sam, row_b = self.sam, self.w["feat_dim"] * 4

        def tot(item):
            return sum(sam.get_log_step_value(e, s, item) for e, s in keys)
        return {"rows": tot(sam.kLogL1FeatureBytes) / row_b, "miss_rows": tot(sam.kLogL1MissBytes) / row_b,
                "graph_bytes": tot(sam.kLogL1GraphBytes), "ms_miss": tot(sam.kLogL3CacheCombineMissTime) * 1e3,
                "ms_cache": tot(sam.kLogL3CacheCombineCacheTime) * 1e3}

    def queue_stats(self, rings):
        return [self.sam.ext_queue_stats(r) for r in range(rings)]

    def ring_mappings(self, rings):
        """per sampler ring: how THIS process read its payloads (samgraph_ext_ring_mapping)"""
        return [self.sam.ext_ring_mapping(r) for r in range(rings)]

    def shutdown(self):
        self.sam.shutdown()


class RehearsalBackend:
    """--rehearse: the job's control plane without a GPU -- launcher, rendezvous, roles, step ranges, the REAL shared
    ring of the engine (its host-only hooks library) between the rank processes (named regions), reductions and the JSON line; a batch is an empty
    message {key, a number of edges derived from the key}.  Numbers printed in this mode measure nothing."""
    SLOTS, SLOT_BYTES = 8, 4096

    def __init__(self, args, w, job, S, T, is_sampler, idx, dev_id, num_epoch):
        import ctypes as C
        self.C = C
        self.eng = C.CDLL(os.path.join(ROOT, "fgnn-artifacts_amd", "samgraph", "torch", "fgnn_engine_hooks.so"))
        self.eng.fgnn_host_queue_open.restype = C.c_void_p
        self.q = C.c_void_p(self.eng.fgnn_host_queue_open(C.c_size_t(self.SLOTS), C.c_size_t(self.SLOT_BYTES)))
        bs = w["batch_size"]
        self.steps_per_epoch = (w["num_train"] + bs - 1) // bs
        self.first, self.local = local_step_range(self.steps_per_epoch, idx, S) if is_sampler else (0, 0)
        self.j = 0
        self.got = {}

    @staticmethod
    def edges_of(key):
        return 1000 + key % 97

    def role_init(self):
        pass

    def num_local_step(self):
        return self.local

    def sample_once(self):
        key = (self.j // self.local) * self.steps_per_epoch + self.first + self.j % self.local
        self.j += 1
        self.eng.fgnn_host_queue_send(self.q, self.C.c_uint64(key), self.C.c_uint64(self.edges_of(key)))

    def extract_start(self, n):
        pass

    def next_batch(self):
        k, v = self.C.c_uint64(), self.C.c_uint64()
        self.eng.fgnn_host_queue_recv(self.q, self.C.byref(k), self.C.byref(v))
        self.got[k.value] = v.value
        return k.value

    def sampler_stats(self, keys):
        return {"edges": float(sum(self.edges_of(e * self.steps_per_epoch + s) for e, s in keys))}

    def trainer_stats(self, keys):
        assert all(self.got[e * self.steps_per_epoch + s] == self.edges_of(e * self.steps_per_epoch + s) for e, s in keys)
        return {"rows": float(len(keys)), "miss_rows": 0.0, "graph_bytes": 0.0, "ms_miss": 0.0, "ms_cache": 0.0}

    def queue_stats(self, rings):
        return [None] * rings

    def ring_mappings(self, rings):
        return [None] * rings

    def shutdown(self):
        self.eng.fgnn_host_queue_close(self.q)


def run_pipeline_rank(args, rank, world):
    import datetime
    import shutil
    if os.environ.get("FGNN_BENCH_WATCHDOG"):  # a stuck rank shows where it is stuck, then exits
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["FGNN_BENCH_WATCHDOG"]), exit=True)
    import torch.distributed as dist
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=900))
    n_dev = torch.cuda.device_count()
    if n_dev == 0 and not args.rehearse:
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    dev_id = local_rank % max(n_dev, 1)  # more ranks than GPUs (a functional check on one GPU): they share
    n1_child = None
    exit_msg = None
    child_env = {k: v for k, v in os.environ.items()
                 if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                              "TORCHELASTIC_RUN_ID", "FGNN_BENCH_CHILD")}
    if rank == 0 and not (args.rehearse or args.no_n1_point or args.decoupled):
        # started now, before this process touches the GPU; it sleeps on its stdin until the spans are done
        # FGNN_BENCH_N1_WRAP (tools): a profiler in front of the child, e.g. "rocprofv3 --kernel-trace --stats -d DIR --"
        wrap = os.environ.get("FGNN_BENCH_N1_WRAP", "").split()
        n1_child = subprocess.Popen(wrap + [sys.executable, BENCH_PY, "--n1-point-child"], env=child_env,
                                    stdin=subprocess.PIPE, stdout=subprocess.PIPE)
    # --samplers auto (the default): how many of the ranks sample is chosen from MEASURED rates -- a sampler process
    # alone and a trainer process alone, a few dozen batches each, in two child processes of ranks 0 and 1 (a 1S+1T job
    # of their own over the job's dataset; started now, before anything here touches the GPU).  Two ranks leave no choice
    auto = str(args.samplers).lower() in ("auto", "0", "none")
    calibrate = auto and world >= 3 and not args.decoupled
    cal_child = None
    if calibrate and not args.rehearse and rank in (0, 1):
        cal_child = subprocess.Popen([sys.executable, BENCH_PY, "--calibrate-child"], env=child_env,
                                     stdin=subprocess.PIPE, stdout=subprocess.PIPE)
    w = WORKLOADS[args.workload]
    if args.num_walks and "num_walks" in w:
        w = dict(w, num_walks=args.num_walks)
    if args.sample_type is None:
        args.sample_type = w["sample_type"]
    bs = w["batch_size"]
    W, K = args.warmup, args.steps
    steps_per_epoch = (w["num_train"] + bs - 1) // bs
    # ---- job-wide names from rank 0: shared-memory prefix (the processes have no common forking parent) and the
    # dataset directory
    obj = [None]
    if rank == 0:
        tag = "fgnn_bench_%d_%x" % (os.getpid(), int(time.time() * 1e3) & 0xFFFFFF)
        base = "/dev/shm" if os.path.isdir("/dev/shm") and \
            os.statvfs("/dev/shm").f_bavail * os.statvfs("/dev/shm").f_frsize > (48 << 30) else "/tmp"
        obj[0] = {"prefix": tag, "dir": os.path.join(base, tag + "_ds"), "dev0": dev_id}
    dist.broadcast_object_list(obj, 0)
    job = obj[0]
    os.environ["SAMGRAPH_SHM_PREFIX"] = job["prefix"]
    os.environ["SAMGRAPH_SHM_KEEP"] = "1"  # rank 0 removes the names after the last barrier
    os.environ["SAMGRAPH_EMPTY_FEAT"] = str(args.empty_feat_bits)
    os.environ.setdefault("SAMGRAPH_LOG_LEVEL", "warn")
    t_setup = time.time()
    info = [None]
    if rank == 0:
        if args.rehearse:
            info[0] = {"num_edge": w["num_edge"], "graph": "none (rehearsal)"}
        else:
            torch.cuda.set_device(dev_id)
            ne, desc = write_dataset(args, w, torch.device("cuda", dev_id), job["dir"])
            info[0] = {"num_edge": ne, "graph": desc}
    dist.broadcast_object_list(info, 0)
    # ---- roles
    sampler_choice = {"mode": "fixed (--samplers)" if not auto else "auto: two ranks leave no choice" if world < 3
                      else "auto switched off by --decoupled (a per-stage diagnostic run)",
                      "reference_split": default_samplers(world)}
    S = pipeline_roles(world, None if auto else args.samplers)[0]
    if calibrate:
        S, sampler_choice = calibrate_roles(args, w, dist, rank, world, dev_id, job, steps_per_epoch, cal_child)
        cal_child = None
    T = world - S
    is_sampler = rank < S
    idx = rank if is_sampler else rank - S
    # Steady-state timing (no barrier inside the measured span): one SPAN of lead + R x K + tail batches goes through the
    # pipeline with the samplers free-running (bounded by the ring) and every trainer stamping CLOCK_MONOTONIC -- one
    # node, one clock for all ranks -- when a batch has been consumed; rank 0 merges the stamps and reads R back-to-back
    # windows of K consecutively consumed batches out of the middle.  `lead` covers the pipeline's fill (first message =
    # one sample chain + one extract) and the warm-up, `tail` the drain (trainers finishing their shares at slightly
    # different times).  The reference times the same loop per epoch (multi_gpu/train_graphsage.py:286-330).
    R = 1 if args.decoupled else max(1, args.windows)
    T_ = T
    lead, tail = span_margins(W, T_, args.decoupled)
    K2 = 0 if (args.no_train_leg or args.rehearse) else train_region_batches(K, args.train_steps, T_)[1]
    total1 = span_total(lead, R, K, tail, T_, False)
    total2 = span_total(lead, R, K2, tail, T_, True) if K2 else 0
    try:
        min_local = steps_per_epoch // S
        per_sampler = max(sum(split_count(n, S, 0) for n in (total1, total2)), 1)
        num_epoch = (per_sampler + min_local - 1) // min_local + 1
        # hand-off self-check at the start of the span's lead: every sampler checksums its first W // S messages, the
        # trainer that receives one recomputes the sum through the address it reads the payload from (the sampler's
        # HBM slot mapped over xGMI, or the pinned host slot) and the job dies on a mismatch (eng_engine.cc)
        check_n = max(W, 1) // S if W // S else 1  # a few: the receiver verifies synchronously (a long checked lead
        # would hold the trainers back and delay the steady state the windows are read from)
        os.environ["SAMGRAPH_HANDOFF_CHECK"] = str(check_n)
        be = (RehearsalBackend if args.rehearse else EngineBackend)(args, w, job, S, T, is_sampler, idx, dev_id, num_epoch)
        dist.barrier()  # every process has attached to every shared region
        if is_sampler:
            be.role_init()
            dist.barrier()  # the rank list is in shared memory: trainers may build their caches
        else:
            dist.barrier()
            be.role_init()
        dist.barrier()
        t_setup = time.time() - t_setup

        first_step, local_steps = local_step_range(steps_per_epoch, idx, S) if is_sampler else (0, 0)
        if is_sampler:
            assert be.num_local_step() == local_steps, (be.num_local_step(), local_steps)
        sampled = [0]  # batches this sampler has produced
        keys = []

        links, rccl_ok = link_selftest(dist, rank, world, dev_id, n_dev, args.rehearse)
        if not rccl_ok:  # every rank has the same verdict: no region that needs RCCL
            K2, total2 = 0, 0
        model = opt = loss_fcn = None
        if K2 and T > 1:
            # gradient all-reduce between the trainers: RCCL ("nccl") when each has its own GPU, gloo when ranks
            # share one (functional check on a single-GPU box: RCCL refuses two ranks on one device).  new_group is
            # a collective call: every rank takes part
            tgroup = dist.new_group(ranks=list(range(S, world)), backend="nccl" if n_dev >= world else "gloo",
                                    timeout=datetime.timedelta(seconds=600))
        if not is_sampler and K2:
            sys.path.insert(0, os.path.join(ROOT, "examples"))
            from models import MODELS
            # the reference's model per workload: GraphSAGE (khop2), GCN (weighted sampling, multi_gpu/train_gcn.py),
            # PinSAGE (random walks, visit counts as edge weights, multi_gpu/train_pinsage.py)
            kind = {"weighted_khop_prefix": "gcn", "random_walk": "pinsage"}.get(args.sample_type, "graphsage")
            model = MODELS[kind](w["feat_dim"], 256, w["num_class"], len(w["fanout"]), 0.5).to("cuda:%d" % dev_id)
            if T > 1:
                model = torch.nn.parallel.DistributedDataParallel(
                    model, device_ids=[dev_id] if n_dev >= world else None, process_group=tgroup)
            loss_fcn = torch.nn.CrossEntropyLoss()
            from fgnn_hip.nn import softmax_xent
            from fgnn_hip.nn import Adam as FusedAdam
            opt = FusedAdam(model.parameters(), lr=0.003)
            (model.module if T > 1 else model).dropout_step = opt.step_count
            model.train()

        now = lambda: time.clock_gettime(time.CLOCK_MONOTONIC)  # noqa: E731  (one node: every rank reads the same clock)

        def span(total, train):
            """`total` batches through the pipeline with NO barrier between the first and the last: this rank's share as
            a sampler (sample_once, free-running against the ring) or as a trainer (get_next_batch [+ training step],
            one CLOCK_MONOTONIC stamp per consumed batch).  Returns (stamps [(t, key)], seconds in this rank's loop)."""
            mine = split_count(total, S, idx) if is_sampler else split_count(total, T, idx)
            stamps = []
            import gc
            gc.collect()
            gc.disable()  # (no_gc: a collector pass inside a rank's loop is a multi-millisecond hole in the stamps)
            dist.barrier()
            t0 = now()
            if is_sampler:
                for _ in range(mine):
                    be.sample_once()
                    j = sampled[0]
                    sampled[0] += 1
                    keys.append((j // local_steps, first_step + j % local_steps))
                loop_s = now() - t0
                if args.decoupled:
                    time.sleep(0.02)  # the publisher thread publishes the last batches as their GPU work completes
                    dist.barrier()
            else:
                if args.decoupled:
                    dist.barrier()  # diagnostic: the samplers have filled the queue, the trainers run alone
                    t0 = now()
                if mine:
                    be.extract_start(mine)
                for _ in range(mine):
                    key = be.next_batch()
                    if train:
                        blocks, feat, label = be.blocks(key)
                        out = model(blocks, feat)
                        loss, g = softmax_xent(out, label)  # CrossEntropyLoss + its gradient, one launch
                        opt.zero_grad()
                        out.backward(g)
                        opt.step()
                        torch.cuda.current_stream().synchronize()
                    stamps.append((now(), key))
                    keys.append((key // steps_per_epoch, key % steps_per_epoch))
                loop_s = now() - t0
            gc.enable()
            dist.barrier()  # every batch of the span has been consumed
            return stamps, loop_s

        def collect(stamps, per_key):
            """every rank's stamps and per-batch figures on rank 0 (after the span: nothing of this is timed)"""
            got = [None] * world
            dist.all_gather_object(got, (stamps, per_key))
            all_stamps = [x for st, _ in got for x in st]
            merged_pk = {}
            for _, pk in got:
                for k, v in pk.items():
                    merged_pk.setdefault(k, {}).update(v)
            return all_stamps, merged_pk

        # ---- span 1: sample -> hand-off -> cached extraction
        stamps, loop_s = span(total1, False)
        # a sampler's publisher thread logs a batch when it publishes it: all published by the barrier above
        per_key = {}
        for e, st in keys:
            g = e * steps_per_epoch + st
            per_key[g] = ({"edges": be.sampler_stats([(e, st)])["edges"]} if is_sampler
                          else be.trainer_stats([(e, st)]))
        all_stamps, pk = collect(stamps, per_key)
        n_produced = len(keys) if is_sampler else 0
        del keys[:]
        # ---- span 2: the same with a training step per consumed batch
        train_stamps = []
        if K2:
            stamps2, _ = span(total2, True)
            train_stamps, _ = collect(stamps2, {})
            del keys[:]

        def red(vals, op):
            t = torch.tensor(vals, dtype=torch.float64)
            dist.all_reduce(t, op=op)
            return [float(x) for x in t]
        s_loop, t_loop, setup_max = red([loop_s if is_sampler else 0.0, loop_s if not is_sampler else 0.0, t_setup],
                                        dist.ReduceOp.MAX)
        nb_s, = red([n_produced], dist.ReduceOp.SUM)
        dist.barrier()  # every trainer has verified what it was going to verify
        rings = be.queue_stats(S) if rank == 0 else None  # shared counters: any process of the job can read them
        # where every rank's GPU hangs (NUMA node of its PCIe root) next to where the shared host feature table lives
        gnodes = [None] * world
        dist.all_gather_object(gnodes, None if args.rehearse else gpu_numa_node(dev_id))
        # every trainer: how it read each sampler's ring (mapped device to device, or copied back through the host slot)
        maps = [None] * world
        dist.all_gather_object(maps, None if is_sampler else {"rank": rank, "device": dev_id,
                                                             "rings": be.ring_mappings(S)})
        # open descriptors per rank while every ring is still mapped (a trainer imports up to S x 170 ring slots, each an
        # IPC buffer of its own) next to the soft limit the process runs under
        import resource
        nfiles = [None] * world
        try:
            mine_fd = len(os.listdir("/proc/self/fd"))
        except OSError:
            mine_fd = None
        dist.all_gather_object(nfiles, mine_fd)
        be.shutdown()
        dist.barrier()
        n1_point = None
        if rank == 0:
            n1_point = {"value": None, "why": "control-plane rehearsal" if args.rehearse else "not requested"}
            if n1_child is not None:
                req = {"env": {"SAMGRAPH_SHM_PREFIX": job["prefix"] + "_n1", "SAMGRAPH_EMPTY_FEAT": str(args.empty_feat_bits),
                               "SAMGRAPH_LOG_LEVEL": os.environ.get("SAMGRAPH_LOG_LEVEL", "warn")},
                       "dev_id": dev_id, "dir": job["dir"], "lead": lead, "windows": R, "steps": K, "tail": tail,
                       "steps_per_epoch": steps_per_epoch, "sample_type": args.sample_type, "batch_size": bs,
                       "cache_ratio": args.cache_ratio, "presample_epochs": max(1, args.presample_epochs),
                       "fanout": w["fanout"], "seed": args.seed, "sampler_keys": sampler_config_keys(w, args.sample_type),
                       "row_bytes": w["feat_dim"] * 4}
                try:
                    o, _ = n1_child.communicate((json.dumps(req) + "\n").encode(), timeout=float(
                        os.environ.get("FGNN_BENCH_N1_TIMEOUT", "300")))
                    lines = [ln for ln in o.decode(errors="replace").splitlines() if ln.startswith("{")]
                    n1_point = json.loads(lines[-1]) if lines else {"value": None, "error": "rc %s" % n1_child.returncode}
                except Exception as e:  # a secondary measurement must not cost the line
                    n1_child.kill()
                    n1_point = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
                n1_child = None
        if rank == 0:
            assert int(nb_s) == total1 and len(all_stamps) == total1, (nb_s, len(all_stamps), total1)
            assert len({k for _, k in all_stamps}) == total1  # every batch reached exactly one trainer
            merged, wins = read_windows(all_stamps, lead, R, K)
            if os.environ.get("FGNN_BENCH_DUMP_STAMPS"):  # tools: every consumed batch's stamp (seconds from the first)
                with open(os.environ["FGNN_BENCH_DUMP_STAMPS"], "w") as f:
                    f.write("# first stamp at CLOCK_MONOTONIC %.6f\n" % merged[0][0])
                    for t, k in merged:
                        f.write("%.6f %d\n" % (t - merged[0][0], k))
            win_ms = [(t1 - t0_) / K * 1e3 for t0_, t1, _ in wins]
            order = sorted(range(R), key=lambda r: win_ms[r])
            med = order[(R - 1) // 2]  # the median window (the slower of the middle two for an even R)
            t_max = wins[med][1] - wins[med][0]
            mkeys = wins[med][2]
            edges = sum(pk[k]["edges"] for k in mkeys)
            rows, miss_rows, graph_bytes = (sum(pk[k][n] for k in mkeys) for n in ("rows", "miss_rows", "graph_bytes"))
            # launch averages (roofline): over the launches of ALL R windows (R x K of each kind)
            wkeys = [k for _, _, ks in wins for k in ks]
            rows_a, miss_rows_a, ms_miss, ms_cache = (sum(pk[k][n] for k in wkeys)
                                                      for n in ("rows", "miss_rows", "ms_miss", "ms_cache"))
            n_launch = len(wkeys)
            t_train = train_win_ms = None
            if K2:
                _, twins = read_windows(train_stamps, lead, R, K2)
                train_win_ms = [(t1 - t0_) / K2 * 1e3 for t0_, t1, _ in twins]
                t_train = sorted(train_win_ms)[(R - 1) // 2] * K2 * 1e-3
            live = [r for r in rings if r]
            handoff = {"rings": rings, "check_messages_per_sampler": check_n,
                       "verified": sum(r["verified"] for r in live), "check_failed": sum(r["check_failed"] for r in live),
                       "sent_device": sum(r["sent_device"] for r in live), "sent_host": sum(r["sent_host"] for r in live),
                       "spilled": sum(r["spilled"] for r in live),
                       "transport": ("none (rehearsal)" if not live else
                                     "sampler HBM ring, peer-read by the trainers" if all(
                                         r["sent_host"] == 0 and r["spilled"] == 0 and r["sent_device"] > 0 for r in live)
                                     else "pinned host ring" if all(r["sent_device"] == 0 for r in live)
                                     else "MIXED: part of the messages fell back to the pinned host ring"),
                       "note": "per sampler ring: slots, messages by payload location, copies back on request, and the "
                               "warm-up messages the receiving trainers verified end to end (a mismatch aborts the job)"}
            handoff["trainers"] = [m for m in maps if m]
            if live and handoff["check_failed"]:
                sys.exit("bench.py: hand-off check failed: %s" % handoff)
            # A run that was meant to read the samplers' HBM rings peer to peer but moved payloads through pinned host
            # memory is a different (slower) system: it must not pass for the real thing.  Asked-for host transport
            # (SAMGRAPH_DEVICE_RING_SLOTS=0, the forced-spill test switch) is fine
            asked_host = os.environ.get("SAMGRAPH_DEVICE_RING_SLOTS") == "0" or \
                os.environ.get("SAMGRAPH_DEVICE_RING_FORCE_SPILL") not in (None, "", "0")
            refused = [(m["rank"], i) for m in handoff["trainers"] for i, g in enumerate(m["rings"]) if g and g["state"] == 3]
            degraded = live and (refused or any(r["sent_host"] or r["spilled"] for r in live))
            handoff["degraded"] = bool(degraded) and not asked_host
            row_b = w["feat_dim"] * 4
            hit_rows = rows - miss_rows
            handoff_bytes = graph_bytes + 8 * rows + 4 * bs * K  # COO arrays + (miss|cache) index pairs + output ids
            cache_launch_bytes = (rows_a - miss_rows_a) * (2 * row_b + 8)  # of all R windows' launches, like ms_cache
            out = {
                "metric": f"sampled-edges/sec ({args.sample_type} fanout {'/'.join(map(str, w['fanout']))}, batch {bs}, "
                          "factored pipeline: sampler GPUs (sample + dedup + remap + cache-index split) -> HBM message "
                          "ring -> trainer GPUs (cached feature extraction)); edges of K consecutively consumed batches "
                          f"/ the time the trainers took to consume them, median of {R} back-to-back windows of a "
                          "continuously full pipeline",
                "value": edges / t_max, "unit": "edges/s", "n_gpus": world, "steps": K, "warmup": W,
                "ms_per_step": t_max / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "u32", "data": "synthetic" if not args.rehearse else "none (control-plane rehearsal)",
                "windows": {"count": R, "ms_per_step": win_ms, "min": min(win_ms), "max": max(win_ms), "median_index": med,
                            "lead_batches": lead, "tail_batches": total1 - lead - R * K, "span_batches": total1,
                            "span_ms_per_step": (wins[-1][1] - wins[0][0]) / (R * K) * 1e3,
                            "median_window_keys": mkeys,
                            "clock": "CLOCK_MONOTONIC stamped by the consuming trainer after every batch; window j = "
                                     "batches lead + jK .. lead + (j+1)K - 1 in consumption order over all trainers, from "
                                     "the stamp of the batch before it to the stamp of its last; no barrier inside the "
                                     "span; value / ms_per_step come from the median window"},
                "config": {"workload": f"{args.workload}-shaped synthetic graph: {info[0]['graph']}; N={w['num_node']}, "
                                       f"E={info[0]['num_edge']}, train set {w['num_train']} uniform random ids (seed 1), "
                                       f"{args.sample_type} fanout {w['fanout']}, batch {bs}; features in host memory "
                                       f"(2^{args.empty_feat_bits} rows, ids masked = SAMGRAPH_EMPTY_FEAT), pre-sample "
                                       f"cache ratio {args.cache_ratio} (presample_epoch {max(1, args.presample_epochs)}) "
                                       "in every trainer's HBM; arch5 through "
                                       "samgraph.torch / c_lib.so, one process per GPU",
                           "global_batch": bs, "parallelism": f"{S}S+{T}T (samplers -> device ring -> trainers)"},
                "roofline": {"bound": "hbm", "kernel": "extract_fused_kernel, HBM band (CombineCacheData on the trainer GPUs; the "
                                                        "same launch's link band pulls the miss rows over the host link)",
                             "achieved": cache_launch_bytes / (ms_cache * 1e-3) / 1e9 if ms_cache else None,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": cache_launch_bytes / (ms_cache * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_cache else None,
                             "traffic": None, "avg_launch_ms": ms_cache / n_launch, "timed_launches": n_launch,
                             "algorithmic_bytes_per_launch": cache_launch_bytes / n_launch,
                             "note": "hit rows x (row read + row write + 2 index words) / the HBM band's first start .. "
                                     "last end inside the one-launch extraction (device clock stamps of its workgroups), "
                                     "summed over the trainers' batches"},
                "pipeline": {
                    "samplers": S, "trainers": T, "devices": min(n_dev, world), "sampler_choice": sampler_choice,
                    "sampler_loop_ms_per_batch": s_loop / max(split_count(total1, S, 0), 1) * 1e3,
                    "trainer_loop_ms_per_batch": t_loop / max(split_count(total1, T, 0), 1) * 1e3,
                    "loop_note": "wall time of a rank's whole span loop / its batches (MAX within the role): a sampler's "
                                 "includes its waits on a full ring, a trainer's its waits on an empty one -- each "
                                 "stage ALONE only with --decoupled",
                    "sampler_busy_s": s_loop, "trainer_busy_s": t_loop,
                    # steady state of the consuming side: the second half of the span's stamps (the first batches of a
                    # process pay its pool's first allocations; with --decoupled this is the trainers ALONE)
                    "consumed_second_half_ms_per_batch":
                        (merged[-1][0] - merged[len(merged) // 2][0]) / max(len(merged) - 1 - len(merged) // 2, 1) * 1e3,
                    "trainer_rows_per_s": rows / t_max, "hit_rate": hit_rows / max(rows, 1.0),
                    "handoff_bytes_per_step": handoff_bytes / K, "handoff_GBps": handoff_bytes / t_max / 1e9,
                    "handoff_peak_GBps": XGMI_LINK_GBS, "handoff": handoff, "links": links,
                    "open_files": {"per_rank": nfiles, "soft_limit": resource.getrlimit(resource.RLIMIT_NOFILE)[0]},
                    "numa": {"gpu_node_of_rank": gnodes, "nodes_with_memory": numa_nodes_with_memory(),
                             "host_feat_policy": os.environ.get("SAMGRAPH_HOST_FEAT_NUMA",
                                                                "interleave over the nodes with memory (default)"),
                             "note": "every trainer pulls its miss rows out of ONE shared host table (DESIGN 6)"},
                    "n1_point_of_this_curve": n1_point,
                    "miss": {"bound": "host link", "bytes_per_step": miss_rows * row_b / K,
                             "achieved": miss_rows * row_b / t_max / 1e9 / T, "peak": HOST_LINK_GBS,
                             "unit": "GB/s per trainer GPU", "frac": miss_rows * row_b / t_max / 1e9 / T / HOST_LINK_GBS,
                             "avg_band_ms": ms_miss / n_launch,
                             "band_GBps": miss_rows_a * row_b / (ms_miss * 1e-3) / 1e9 if ms_miss else None,
                             "note": "achieved = miss bytes / wall time per trainer; band = the link band of one launch "
                                     "(up to four batches' bands share a trainer's link)"},
                },
                "epoch_time_s": {"sample_plus_extract": steps_per_epoch * t_max / K,
                                 "with_training": steps_per_epoch * t_train / K2 if K2 else None,
                                 "training_steps_timed": K2, "training_windows_ms_per_step": train_win_ms,
                                 "note": f"{steps_per_epoch} steps/epoch x seconds per step of the median window; with_training "
                                         "= the same pipeline with a GraphSAGE step (examples/models.py, hidden 256, "
                                         "Adam) on every batch, gradients all-reduced over RCCL between trainers"},
                "edges_per_step": edges / K, "input_nodes_per_step": rows / K, "setup_s": setup_max,
            }
            print(json.dumps(out), flush=True)
            if handoff.get("degraded") and n_dev >= world:
                exit_msg = ("bench.py: every rank has its own GPU but payloads went through the pinned host ring "
                            "(trainer, ring) refused: %s; rings: %s -- the line above is NOT the peer-read pipeline"
                            % (refused, rings))
    finally:
        if n1_child is not None:  # never asked (an error above): an empty line ends it
            try:
                n1_child.communicate(b"\n", timeout=30)
            except Exception:
                n1_child.kill()
        try:
            dist.barrier()
        except Exception:
            pass
        if rank == 0:
            shutil.rmtree(job["dir"], ignore_errors=True)
            if os.path.isdir("/dev/shm"):
                for f in os.listdir("/dev/shm"):
                    if f.startswith(job["prefix"]):
                        try:
                            os.unlink(os.path.join("/dev/shm", f))
                        except OSError:
                            pass
    dist.destroy_process_group()
    if exit_msg:
        sys.exit(exit_msg)


def limited_collective(dist, world, body, limit_s):
    """`body` (collective calls that may never return when a peer has failed) on a helper thread; this rank waits for
    ITS OWN thread for at most limit_s seconds, then every rank learns over the job's gloo group who finished.
    Returns (body's result | None, {rank: reason} of the ranks that did not finish -- the same dict on every rank)."""
    import threading
    res = {}

    def run():
        try:
            res["rec"] = body()
        except Exception as e:
            res["err"] = "%s: %s" % (type(e).__name__, e)
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(limit_s)
    mine = None if "rec" in res else res.get("err", "no answer within %.0f s" % limit_s)
    status = [None] * world
    dist.all_gather_object(status, mine)  # gloo: works whatever the helper thread is stuck in
    return res.get("rec"), {r: e for r, e in enumerate(status) if e is not None}


def link_selftest(dist, rank, world, dev_id, n_dev, rehearse, limit_s=None):
    """First-contact proof for N >= 2, run once before the timed span: ONE RCCL all-reduce over ALL ranks (samplers
    included: `rccl_world` == world says RCCL saw every rank; the data path itself has no collective), its bus
    bandwidth on a 64 MiB payload, and the peer-access matrix between the ranks' GPUs (what the trainers' peer reads of
    the samplers' HBM rings rest on).  With fewer GPUs than ranks RCCL refuses (two ranks on one device): recorded as
    such, nothing is faked.  Every rank takes part; rank 0 gets the record.

    A rank whose RCCL initialisation fails ALONE must not leave the others inside a collective: the RCCL calls run on a
    helper thread, every rank waits for ITS OWN thread for at most `limit_s` seconds, then all ranks agree over gloo on
    who finished; on any failure every rank aborts its communicator (ncclCommAbort ends a kernel that waits for a peer)
    and the record says which ranks failed and why.  Returns (record on rank 0 | None, rccl_ok on every rank)."""
    import datetime
    if limit_s is None:
        limit_s = float(os.environ.get("FGNN_BENCH_LINK_TIMEOUT", "150"))
    if rehearse:
        return {"rccl_world": None, "why": "control-plane rehearsal: no GPU work"}, True
    ok = True
    if n_dev < world:
        rec = {"rccl_world": None, "why": "%d ranks share %d GPU(s): RCCL needs a device per rank" % (world, n_dev)}
    else:
        grp = dist.new_group(ranks=list(range(world)), backend="nccl", timeout=datetime.timedelta(seconds=limit_s + 30))
        dev = torch.device("cuda", dev_id)

        def body():
            torch.cuda.set_device(dev)
            if os.environ.get("FGNN_BENCH_LINK_FAIL_RANK") == str(rank):  # tests: this rank fails alone
                raise RuntimeError("injected failure (FGNN_BENCH_LINK_FAIL_RANK)")
            one = torch.ones(1, device=dev)
            dist.all_reduce(one, group=grp)
            torch.cuda.synchronize(dev)
            buf = torch.ones(16 << 20, dtype=torch.float32, device=dev)  # 64 MiB
            dist.all_reduce(buf, group=grp)  # first use of the size
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(5):
                dist.all_reduce(buf, group=grp)
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / 5
            nbytes = buf.numel() * 4
            return {"rccl_world": int(round(float(one.item()))), "allreduce_64MiB_ms": dt * 1e3,
                    "allreduce_busbw_GBps": 2 * (world - 1) / world * nbytes / dt / 1e9,
                    "note": "one all-reduce over all ranks before the span (samplers too); busbw = 2(n-1)/n x "
                            "bytes / time (ranks enter the timed loop unsynchronised: a lower bound)"}
        res, bad = limited_collective(dist, world, body, limit_s)
        if bad:
            ok = False
            try:  # end whatever this rank's thread still has in flight
                grp._get_backend(dev).abort()
            except Exception:
                pass
            rec = {"rccl_world": None, "error": "RCCL self-test failed", "failed_ranks": {str(r): e for r, e in bad.items()},
                   "limit_s": limit_s, "consequence": "the training region (gradient all-reduce over RCCL) is skipped; the "
                                                      "data path has no collective and is measured as usual"}
        else:
            rec = res
    # row of the peer matrix for this rank's GPU: which other ranks' GPUs it can map
    row = []
    for r in range(world):
        other = r % max(n_dev, 1)
        try:
            row.append(True if other == dev_id else bool(torch.cuda.can_device_access_peer(dev_id, other)))
        except Exception:
            row.append(None)
    rows = [None] * world
    dist.all_gather_object(rows, row)
    rec["peer_access"] = {"matrix": rows, "note": "matrix[i][j]: rank i's GPU can map rank j's GPU memory "
                                                  "(hipDeviceCanAccessPeer); ranks sharing a GPU read True"}
    return (rec if rank == 0 else None), ok


# ---- the like-for-like N = 1 point of the N >= 2 pipeline, measured inside the same job ------------------------------
def run_n1_point_child():
    """Child of rank 0 of an N >= 2 job, started BEFORE rank 0 touched the GPU (a process that has initialised the GPU
    never starts another program); waits for one JSON request on stdin -- sent after the job's spans, when the ranks have
    shut their engines down -- then runs the SAME pipeline on ONE GPU: the engine's arch3 (sampler and extractor halves
    of arch5 in one process, background threads, in-process ring; the reference's default single-GPU mode,
    cuda_loops_arch3.cc) on the job's dataset, features in host memory behind the same pre-sample cache, timed by the
    same stamps-and-windows rule.  Prints one JSON line."""
    line = sys.stdin.readline()
    if not line.strip():
        return
    if os.environ.get("FGNN_BENCH_WATCHDOG"):
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["FGNN_BENCH_WATCHDOG"]), exit=True)
    req = json.loads(line)
    for k, v in req["env"].items():
        os.environ[k] = v
    import samgraph.torch as sam
    torch.cuda.set_device(req["dev_id"])
    ctx = "cuda:%d" % req["dev_id"]
    lead, R, K, tail = req["lead"], req["windows"], req["steps"], req["tail"]
    total = lead + R * K + tail
    spe = req["steps_per_epoch"]
    cfg = dict(dataset_path=req["dir"], _arch=sam.kArch3, _sample_type=sam.sample_types[req["sample_type"]],
               batch_size=req["batch_size"], num_epoch=(total + spe - 1) // spe + 1,
               _cache_policy=sam.cache_policies["pre_sample"], presample_epoch=req["presample_epochs"],
               cache_percentage=req["cache_ratio"],
               max_sampling_jobs=10, max_copying_jobs=2, omp_thread_num=8, sampler_ctx=ctx, trainer_ctx=ctx,
               seed=req["seed"], **req["sampler_keys"])
    sam.config(cfg)
    t0 = time.time()
    sam.init()
    setup = time.time() - t0
    sam.start()
    stamps = []
    with no_gc():
        for _ in range(total):
            key = sam.get_next_batch()
            stamps.append((time.clock_gettime(time.CLOCK_MONOTONIC), key))
    _, wins = read_windows(stamps, lead, R, K)
    win_ms = [(b - a) / K * 1e3 for a, b, _ in wins]
    med = sorted(range(R), key=lambda r: win_ms[r])[(R - 1) // 2]
    edges = sum(sam.get_log_step_value(k // spe, k % spe, sam.kLogL1NumSample) for k in wins[med][2])
    rows = sum(sam.get_log_step_value(k // spe, k % spe, sam.kLogL1FeatureBytes) for k in wins[med][2]) / req["row_bytes"]
    miss = sum(sam.get_log_step_value(k // spe, k % spe, sam.kLogL1MissBytes) for k in wins[med][2]) / req["row_bytes"]
    out = {"value": edges / (wins[med][1] - wins[med][0]), "unit": "edges/s", "n_gpus": 1, "ms_per_step": win_ms[med],
           "windows_ms_per_step": win_ms, "steps": K, "hit_rate": (rows - miss) / max(rows, 1.0), "setup_s": setup,
           "what": "the same pipeline on ONE GPU of this job: arch3 through samgraph.torch / c_lib.so (sampler + extractor "
                   "threads in one process, in-process ring), same dataset, features in host memory behind the same "
                   "pre-sample cache, same windows rule -- the like-for-like N = 1 point of this line (the N = 1 "
                   "bench line itself is config 2's shape: features HBM-resident)"}
    print(json.dumps(out), flush=True)
    sam.shutdown()
