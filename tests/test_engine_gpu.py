"""End-to-end through the drop-in boundary on a real GPU: `import samgraph.torch as sam` -> samgraph_* C ABI ->
HIP kernels, every batch compared bit-for-bit with the oracle (tests/engine_runner.py)."""
import os
import re
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
RUNNER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "engine_runner.py")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, *args, env=None):
    p = subprocess.run([sys.executable, RUNNER, args[0], args[1], str(tmp_path)] + [str(a) for a in args[2:]],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-6000:]
    return p.stdout + ("\n" + p.stderr if env and "SAMGRAPH_LOG_LEVEL" in env else "")


@pytest.mark.parametrize("sample_type", ["khop2", "khop0", "weighted_khop_prefix", "random_walk", "khop1",
                                         "weighted_khop", "weighted_khop_hash_dedup"])
def test_arch1_single_gpu(tmp_path, sample_type):
    """BASELINE config 2 in miniature: one GPU samples and extracts (cuda_loops_arch1.cc:44-80)."""
    assert "ok" in _run(tmp_path, "arch1", sample_type)


@pytest.mark.parametrize("mode,args", [("arch1", ["khop2"]), ("arch5", ["khop2", 2, 1, 0.25, "pipeline"])])
def test_five_epochs(tmp_path, mode, args):
    """Every epoch's permutation (and its device copy) is prepared by a helper thread during the epoch before, out of three
    rotating device arrays, with no flush at the epoch boundary (eng_shuffler.h): five epochs cycle every buffer, and
    each batch is still the one the reference's shuffle (dist_shuffler.cc:98-137) and the oracle produce."""
    assert "ok" in _run(tmp_path, mode, *args, env={"FGNN_TEST_NUM_EPOCH": "5"})


@pytest.mark.parametrize("mode,args", [("arch5", ["khop2", 1, 1, 0.25, "pipeline"]), ("arch5", ["khop0", 2, 1, 0.0, "pipeline"]),
                                       ("arch3", ["khop2", 0.25, "threads"])])
def test_many_epochs_shorter_than_the_pipeline(tmp_path, mode, args):
    """A train set of two batches per epoch (one per sampler with two samplers) over 12 epochs: the sampler runs up to
    six batches = several EPOCHS ahead of the GPU, so the helper that prepares epoch e+1's device seed array must wait
    for the readers of the array it recycles (epoch e-2's, eng_shuffler.h: TrackStream) -- the reference drains its
    pipeline at every epoch boundary instead (dist_loops_arch5.cc:131-137).  Every batch still equals the oracle's."""
    assert "ok" in _run(tmp_path, mode, *args, env={"FGNN_TEST_NUM_EPOCH": "12", "FGNN_TEST_NUM_TRAIN": "500"})


@pytest.mark.parametrize("mode,args", [("arch1", ["khop2"]), ("arch5", ["khop2", 2, 1, 0.25, "pipeline"])])
def test_sanity_check_passes_on_a_clean_train_set(tmp_path, mode, args):
    """SAMGRAPH_SANITY_CHECK=1 (run_config.cc:91, dist_shuffler.cc:169-176): every batch is checked on the GPU for
    invalid ids and for ids already handed out in the epoch; a clean train set runs through, results unchanged."""
    assert "ok" in _run(tmp_path, mode, *args, env={"SAMGRAPH_SANITY_CHECK": "1"})


def test_sanity_check_trips_on_a_duplicated_seed(tmp_path):
    """a train set holding one id twice: fatal at the batch that hands the id out again, like the reference's device
    assert "duplicate batch input" (cuda_sanity_check.cu:37-40); without the flag the same run goes through the engine
    unnoticed (the duplicate only shows in the comparison with the oracle's replay, which is why the flag exists)"""
    p = subprocess.run([sys.executable, RUNNER, "arch1", "khop2", str(tmp_path)], capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, SAMGRAPH_SANITY_CHECK="1", FGNN_TEST_DUP_SEED="1"))
    assert p.returncode != 0
    assert "duplicate batch input" in p.stderr, p.stderr[-3000:]


@pytest.mark.parametrize("mode,args", [("arch1", ["khop2"]), ("arch3", ["khop2", 0.25, "inline"]),
                                       ("arch5", ["khop2", 1, 1, 0.25, "pipeline"])])
def test_empty_feat_mock_extraction(tmp_path, mode, args):
    """SAMGRAPH_EMPTY_FEAT=k (set by the reference's scripts for the *_empty datasets): feat.bin is not read, the
    table has 2^k rows and every feature gather masks the node id -- no out-of-range row is ever touched."""
    assert "ok" in _run(tmp_path, mode, *args, env={"SAMGRAPH_EMPTY_FEAT": "9"})


@pytest.mark.parametrize("arch,sample_type,cache,mode", [
    ("arch3", "khop2", 0.25, "inline"),      # the reference's default for its single-process scripts (common_config.py:48)
    ("arch3", "khop2", 0.25, "threads"),     # samgraph_start: sampler thread + copy/extract thread
    ("arch3", "random_walk", 0.0, "threads"),
    ("arch2", "khop0", 0.0, "inline"),
    ("arch4", "weighted_khop_prefix", 0.3, "threads"),
])
def test_single_process_two_role_archs(tmp_path, arch, sample_type, cache, mode):
    """arch2/3/4: sampler + copy/extract in one process (cuda_loops_arch{2,3,4}.cc), both contexts on cuda:0 here."""
    assert "ok" in _run(tmp_path, arch, sample_type, cache, mode)


@pytest.mark.parametrize("take", [3, 12])
def test_threads_stopped_before_their_last_batch(tmp_path, take):
    """samgraph_shutdown with samgraph_start's loops still running (the sampler blocked on a full queue, the extractor
    on a full graph pool or on an empty queue): no hang, no abort (bench.py's N = 1 point of the pipeline stops so)."""
    assert "early stop ok" in _run(tmp_path, "arch3", "khop2", 0.25, "early_stop", take)


@pytest.mark.parametrize("arch,sample_type,cache,mode", [("arch3", "khop2", 0.25, "inline"),
                                                         ("arch2", "random_walk", 0.1, "threads")])
def test_static_presample_policy(tmp_path, arch, sample_type, cache, mode):
    """kCacheByPreSampleStatic: frequencies over whole L-hop neighbourhoods (cuda/pre_sampler.cc:69-71,
    DoGPUSampleAllNeighbour); the ranking is checked through the miss volume of every batch."""
    out = _run(tmp_path, arch, sample_type, cache, mode, env={"FGNN_TEST_CACHE_POLICY": "static"})
    assert "static-presample ok" in out


@pytest.mark.parametrize("policy", ["degree", "random", "heuristic", "degree_hop", "fake_optimal"])
@pytest.mark.parametrize("mode,args", [("arch3", ["khop2", 0.25, "inline"]), ("arch5", ["khop2", 2, 1, 0.2, "pipeline"])])
def test_file_backed_cache_policies(tmp_path, policy, mode, args):
    """The rankings the engine loads from cache_by_<policy>.bin (engine.cc:216-256; consumed by the cache managers,
    dist_cache_manager_host.cc:60-119, dist_engine.cc:193-229), the file written by tools/dataset/fgnn_dataset: every
    batch bit-exact, and every batch's miss volume equals the rows outside the first cache_percentage * N entries of
    that file -- in the single-process engine and across sampler / trainer processes."""
    out = _run(tmp_path, mode, *args, env={"FGNN_TEST_CACHE_POLICY": policy})
    assert "policy-%s ok" % policy in out
    import re
    miss = [int(m) for m in re.findall(r"(\d+) miss rows", out)]
    assert miss and sum(miss) > 0  # a quarter of the nodes cached: some rows do come from host memory


@pytest.mark.parametrize("sample_type,mode", [("khop0", "inline"), ("khop1", "threads"), ("weighted_khop", "inline")])
def test_arch4_dynamic_cache(tmp_path, sample_type, mode):
    """the arch4 dynamic-cache prototype (DoGPUSampleDyCache + DoDynamicCacheFeatureCopy): blocks, the prefetch node
    list, features and the per-batch miss volume against the oracle's restatement"""
    assert "dynamic-cache %s %s ok" % (sample_type, mode) in _run(tmp_path, "dynamic", sample_type, mode)


def test_static_presample_is_refused_by_the_multi_process_engine(tmp_path):
    """dist/pre_sampler.cc:87-88: LOG(FATAL) "kCacheByPreSampleStatic is not implemented in DistEngine now!" """
    p = subprocess.run([sys.executable, RUNNER, "arch5", "khop2", str(tmp_path), "1", "1", "0.25"],
                       capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, FGNN_TEST_CACHE_POLICY="static"))
    assert p.returncode != 0
    assert "kCacheByPreSampleStatic is not implemented in DistEngine now!" in p.stderr


@pytest.mark.parametrize("sample_type,ns,nt,cache,mode", [
    ("khop2", 1, 1, 0.25, "pipeline"),       # BASELINE config 3 in miniature: 1S + 1T, presample cache
    ("khop2", 1, 1, 0.0, "inline"),          # no cache: input nodes shipped, all rows fetched from host memory
    ("khop2", 2, 1, 0.25, "pipeline"),       # two samplers with disjoint step ranges, one trainer
    ("weighted_khop_prefix", 1, 2, 0.3, "pipeline"),  # config 4 in miniature
    ("random_walk", 2, 2, 0.2, "inline"),    # config 5 in miniature (edge data shipped)
    ("khop1", 1, 1, 0.2, "pipeline"),
])
def test_arch5_multi_process(tmp_path, sample_type, ns, nt, cache, mode):
    """FGNN: sampler processes -> shared pinned queue -> trainer processes, all on cuda:0
    (dist_loops_arch5.cc; the reference's --single-gpu topology, common_config.py:186-191)."""
    assert "ok" in _run(tmp_path, "arch5", sample_type, ns, nt, cache, mode)


@pytest.mark.parametrize("mq_bytes,ns,nt", [(300000, 1, 1), (420000, 1, 2), (300000, 2, 1)])
def test_arch5_pipelined_trainer_on_a_tiny_queue(tmp_path, mq_bytes, ns, nt):
    """SAMGRAPH_MQ_BYTES that leaves the shared ring 2-3 slots (what GCN-sized fan-outs leave under the default budget
    with many trainers): the extraction thread keeps up to 4 batches in flight and must never block for a message
    while it holds the slots its sampler needs (eng_engine.cc: StartExtract, MemoryQueue::TryRecv)."""
    out = _run(tmp_path, "arch5", "khop2", ns, nt, 0.25, "pipeline", env={"SAMGRAPH_MQ_BYTES": str(mq_bytes)})
    assert "ok" in out


def test_last_batch_of_a_sampler_loop_is_published_without_a_further_call(tmp_path):
    """An arch5 sampler enqueues the sampling chain of batch k and the TAIL of batch k - 1 per sample_once (the chain
    must not wait for the host to get through a tail).  The reference's scripts end a sampler's epoch loop with a
    barrier while the trainers still need the epoch's last batch: that batch's tail is finished by the engine's
    publisher thread 300 us after the last call -- here the samplers sit for four seconds after their loops before
    they call shutdown, and no trainer ever waits anywhere near that long for a batch."""
    import re
    out = _run(tmp_path, "arch5", "khop2", 2, 1, 0.25, "pipeline", env={"FGNN_TEST_SAMPLER_LINGER": "4"})
    assert "ok" in out
    waits = [float(x) for x in re.findall(r"longest wait for a batch ([0-9.]+) s", out)]
    assert waits and max(waits) < 2.0, out[-2000:]


@pytest.mark.parametrize("slots", [None, 0])
def test_handoff_check_verifies_and_counts(tmp_path, slots):
    """SAMGRAPH_HANDOFF_CHECK=n: every sampler appends a checksum to its first n messages, the receiving trainer
    recomputes it through the address it reads the payload from (the sampler's HBM ring slot mapped with
    hipIpcOpenMemHandle, or the pinned host slot with SAMGRAPH_DEVICE_RING_SLOTS=0) before it uses the batch;
    samgraph_ext_queue_stats reports per ring what went where and how many messages verified."""
    import ast
    import re
    env = {"SAMGRAPH_HANDOFF_CHECK": "5"}
    if slots is not None:
        env["SAMGRAPH_DEVICE_RING_SLOTS"] = str(slots)
    out = _run(tmp_path, "arch5", "khop2", 2, 2, 0.25, "pipeline", env=env)
    assert "ok" in out
    stats = [ast.literal_eval(m) for m in re.findall(r"ring \d+ stats (\{.*\})", out)]
    assert len(stats) == 2
    for st in stats:
        assert st["verified"] == 5 and st["check_failed"] == 0 and st["spilled"] == 0
        assert st["sent_device"] + st["sent_host"] == 8  # 2 epochs x 4 local steps
        assert (st["ring_slots"] == 0 and st["sent_device"] == 0) if slots == 0 else st["sent_device"] > 0


def test_handoff_check_refuses_a_corrupted_payload(tmp_path):
    """a payload word of a published message overwritten between the sampler and the trainer -- from OUTSIDE the engine
    (the test flips it in the job's shared ring through the host-only hooks library; the product has no switch that
    corrupts anything): the receiving trainer aborts, it does not train on the batch"""
    import ctypes as C
    import time
    subprocess.run([sys.executable, RUNNER, "dataset", "khop2", str(tmp_path)], check=True, timeout=600)
    prefix = "fgnn_test_%d_flip" % os.getpid()
    go = os.path.join(str(tmp_path), "go")
    # payloads in the host ring (no device ring): that is the memory this test can reach
    env = dict(os.environ, SAMGRAPH_SHM_PREFIX=prefix, SAMGRAPH_SHM_KEEP="1", SAMGRAPH_HANDOFF_CHECK="3",
               SAMGRAPH_DEVICE_RING_SLOTS="0", FGNN_TEST_HOLD_TRAINER=go)
    procs = {}
    try:
        for role in ("t", "s"):
            procs[role] = subprocess.Popen([sys.executable, RUNNER, "arch5_named", "khop2", str(tmp_path), role, "0", "1",
                                            "1", "0.25"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                           text=True)
        s_out = procs["s"].communicate(timeout=600)[0]  # the sampler has published every batch and left
        assert procs["s"].returncode == 0, s_out[-3000:]
        hooks = C.CDLL(os.path.join(ROOT, "fgnn-artifacts_amd", "samgraph", "torch", "fgnn_engine_hooks.so"))
        rcs = [hooks.fgnn_host_queue_flip_word(("/%s.%d" % (prefix, k)).encode(), C.c_size_t(1), C.c_size_t(13))
               for k in range(32) if os.path.exists("/dev/shm/%s.%d" % (prefix, k))]
        assert rcs.count(0) == 1, rcs  # exactly one of the job's regions is the message ring
        open(go, "w").close()
        t_out = procs["t"].communicate(timeout=600)[0]
        assert procs["t"].returncode != 0
        assert "hand-off check: message 1" in t_out and "does not verify" in t_out, t_out[-3000:]
    finally:
        for p in procs.values():
            if p.poll() is None:
                p.kill()
        for f in os.listdir("/dev/shm"):
            if f.startswith(prefix):
                os.unlink(os.path.join("/dev/shm", f))


def _two_gpu_env():
    import torch
    n = torch.cuda.device_count()  # counting devices does not initialise the GPU in this process
    if n < 2:
        pytest.skip("needs two GPUs (sampler on cuda:0, trainers on cuda:1): this box has %d" % n)
    return {"FGNN_TEST_SAMPLER_DEVICE": "cuda:0", "FGNN_TEST_TRAINER_DEVICE": "cuda:1", "SAMGRAPH_LOG_LEVEL": "info"}


@pytest.mark.parametrize("sample_type,ns,nt,cache,slots", [
    ("khop2", 1, 1, 0.25, None),             # BASELINE config 3's topology: sampler GPU -> trainer GPU
    ("random_walk", 2, 2, 0.2, 2),           # few ring slots: part of the messages travel through the host ring
    ("khop2", 1, 2, 0.0, None),              # no cache: input nodes shipped
    ("khop2", 2, 1, 0.25, -1),               # the receiver pretends it cannot map the ring: copied back on request
])
def test_arch5_across_two_devices(tmp_path, sample_type, ns, nt, cache, slots):
    """The hand-off across devices (task_queue.cc:154-347 replaced by the HBM message ring): samplers on cuda:0,
    trainers on cuda:1 -- the ring is mapped with hipIpcOpenMemHandle and read peer to peer over xGMI.  Same batches,
    bit for bit; the forced-spill fallback is exercised across devices as well.  Skipped on a one-GPU box."""
    import re
    env = _two_gpu_env()
    if slots is not None and slots >= 0:
        env["SAMGRAPH_DEVICE_RING_SLOTS"] = str(slots)
    if slots == -1:
        env["SAMGRAPH_DEVICE_RING_FORCE_SPILL"] = "1"
    out = _run(tmp_path, "arch5", sample_type, ns, nt, cache, "pipeline", env=env)
    assert "ok" in out
    used = [tuple(int(x) for x in m) for m in
            re.findall(r"device ring \d+: (\d+) messages through HBM, (\d+) through the host ring, (\d+) copied", out)]
    assert used and all(a > 0 for a, _, _ in used), out[-2000:]
    if slots == -1:
        assert sum(c for _, _, c in used) > 0


@pytest.mark.parametrize("sample_type,ns,nt,cache", [("khop2", 1, 1, 0.25), ("khop2", 2, 2, 0.2),
                                                     ("random_walk", 1, 2, 0.0)])
def test_arch5_unrelated_processes_named_regions(tmp_path, sample_type, ns, nt, cache):
    """The MI355X launch style -- one process per GPU started by a launcher, no forking parent: every worker runs
    config + data_init itself and the queue, the sampler barrier, the rank list and the feature table live in named
    shared-memory regions (SAMGRAPH_SHM_PREFIX).  Same batches, bit for bit."""
    subprocess.run([sys.executable, RUNNER, "dataset", sample_type, str(tmp_path)], check=True, timeout=600)
    prefix = "fgnn_gputest_%d" % os.getpid()
    env = dict(os.environ, SAMGRAPH_SHM_PREFIX=prefix, SAMGRAPH_SHM_KEEP="1")
    procs = []
    try:
        for role, n in (("t", nt), ("s", ns)):
            for i in range(n):
                procs.append(subprocess.Popen([sys.executable, RUNNER, "arch5_named", sample_type, str(tmp_path), role,
                                               str(i), str(ns), str(nt), str(cache)], env=env, stdout=subprocess.PIPE,
                                              stderr=subprocess.STDOUT, text=True))
        outs = [p.communicate(timeout=900)[0] for p in procs]
        assert [p.returncode for p in procs] == [0] * len(procs), "\n".join(o[-3000:] for o in outs)
        assert all("arch5-named" in o and "ok" in o for o in outs)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for f in os.listdir("/dev/shm"):
            if f.startswith(prefix):
                os.unlink(os.path.join("/dev/shm", f))


@pytest.mark.parametrize("samplers", ["2", "auto"])
def test_bench_pipeline_five_ranks_on_one_gpu(tmp_path, samplers):
    """The first multi-GPU run rehearsed as far as one GPU allows: bench.py --gpus 5 through bench.main's launcher with
    the REAL engine on every rank (the box lets six processes use its GPU at once and this test process is one of them,
    hence five ranks and not eight; the 8-rank control plane is rehearsed without a GPU in
    tests/test_bench_distributed.py) -- two samplers like the reference's split of an 8-GPU node (--samplers 2:
    2S+3T), and the split chosen from the measured per-role rates (--samplers auto: two calibration
    children run a 1S+1T job of their own first).  Windows, roles, the hand-off's verification counters, what
    `degraded` means on a shared GPU, and the exit code."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "5", "--workload", "small", "--steps",
                        "12", "--warmup", "4", "--no-train-leg", "--no-n1-point", "--empty-feat-bits", "16",
                        "--samplers", samplers], capture_output=True, text=True, timeout=1200, env=env, cwd=str(tmp_path))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    pl = out["pipeline"]
    S, T = pl["samplers"], pl["trainers"]
    assert out["n_gpus"] == 5 and S + T == 5 and 1 <= S <= 4
    ch = pl["sampler_choice"]
    if samplers == "auto":
        assert ch["mode"] == "auto" and ch["chosen"] == S, ch
        assert ch["sampler_ms_per_batch_alone"] > 0 and ch["trainer_ms_per_batch_alone"] > 0
        pred = ch["predicted_ms_per_batch_by_samplers"]
        assert pred[str(S)] == min(pred.values())
    else:
        assert S == 2 and "fixed" in ch["mode"]
    assert out["config"]["parallelism"].startswith("%dS+%dT" % (S, T))
    wd = out["windows"]
    assert wd["count"] == 5 and len(wd["median_window_keys"]) == 12 == len(set(wd["median_window_keys"]))
    assert out["value"] > 0 and out["edges_per_step"] > 1000 and 0 < pl["hit_rate"] <= 1
    # every sampler's warm-up messages were verified by whichever trainer received them, none failed
    rings = pl["handoff"]["rings"]
    assert len(rings) == S and all(r["verified"] > 0 and r["check_failed"] == 0 for r in rings), rings
    # every trainer mapped every sampler's ring device to device; nothing went through pinned host memory unasked
    tr = pl["handoff"]["trainers"]
    assert len(tr) == T and all(len(t["rings"]) == S and all(r["state"] in (1, 2) for r in t["rings"]) for t in tr), tr
    assert not pl["handoff"]["degraded"]
    assert pl["links"]["rccl_world"] is None and "share" in pl["links"]["why"]  # five ranks, one GPU: no RCCL world


@pytest.mark.parametrize("sample_type", ["weighted_khop_prefix", "random_walk"])
def test_bench_pipeline_two_processes_weighted_and_walks(tmp_path, sample_type):
    """BASELINE configs 4 and 5 in their PIPELINE form through bench.main: weighted sampling (the dataset gets its
    prob_prefix_table.bin, the trainers a GCN) and random walks (walk parameters in the run config, visit counts shipped
    as edge data, the trainers a PinSAGE) -- sampler process -> HBM ring -> trainer process, training span included."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "small",
                        "--sample-type", sample_type, "--steps", "10", "--warmup", "3", "--train-steps", "4",
                        "--no-n1-point", "--empty-feat-bits", "16"], capture_output=True, text=True, timeout=900, env=env,
                       cwd=str(tmp_path))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["edges_per_step"] > 1000 and sample_type in out["metric"]
    assert 0 < out["pipeline"]["hit_rate"] <= 1 and out["epoch_time_s"]["with_training"] > 0
    rings = out["pipeline"]["handoff"]["rings"]
    assert rings[0]["verified"] > 0 and rings[0]["check_failed"] == 0 and not out["pipeline"]["handoff"]["degraded"]


def test_bench_pipeline_two_processes(tmp_path):
    """bench.py --gpus 2 = 1S+1T as two processes through bench.main's launcher, the engine and the device ring (both
    ranks share cuda:0 on a one-GPU box)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "small", "--steps",
                        "12", "--warmup", "3", "--train-steps", "4", "--empty-feat-bits", "16"], capture_output=True,
                       text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["steps"] == 12 and out["pipeline"]["samplers"] == 1
    assert out["value"] > 0 and out["edges_per_step"] > 1000 and 0 < out["pipeline"]["hit_rate"] <= 1
    assert out["epoch_time_s"]["with_training"] > 0
    # steady-state windows out of one span, and the like-for-like N = 1 point measured by rank 0's child (arch3)
    wd = out["windows"]
    assert wd["count"] == 5 and len(wd["ms_per_step"]) == 5 and out["ms_per_step"] == sorted(wd["ms_per_step"])[2]
    assert wd["lead_batches"] >= 340 and len(wd["median_window_keys"]) == 12
    n1 = out["pipeline"]["n1_point_of_this_curve"]
    assert n1["value"] > 0 and n1["n_gpus"] == 1 and len(n1["windows_ms_per_step"]) == 5, n1
    # every trainer says how it read the sampler's ring: mapped device to device (two processes, one GPU here)
    tr = out["pipeline"]["handoff"]["trainers"]
    assert len(tr) == 1 and tr[0]["rings"][0]["state"] == 2 and not out["pipeline"]["handoff"]["degraded"]
    # the link self-test: with a GPU per rank RCCL must have seen both ranks; on a one-GPU box the line says why not
    links = out["pipeline"]["links"]
    if torch.cuda.device_count() >= 2:
        assert links["rccl_world"] == 2 and links["allreduce_busbw_GBps"] > 0
    else:
        assert links["rccl_world"] is None and "share" in links["why"]
    assert len(links["peer_access"]["matrix"]) == 2 and all(len(r) == 2 for r in links["peer_access"]["matrix"])


def test_bench_pipeline_two_samplers_three_trainers(tmp_path):
    """the shape default_samplers(8) picks -- several samplers feeding several trainers -- as far as one GPU box allows
    (at most 6 processes may use its GPU: 2S+3T ranks plus this test process): the real launcher, engine, device rings
    and gradient all-reduce between the trainers; the line carries the per-ring hand-off counters and the verified
    warm-up messages of the hand-off self-check."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "5", "--samplers", "2", "--workload",
                        "small", "--steps", "15", "--warmup", "6", "--train-steps", "6", "--empty-feat-bits", "16",
                        "--no-n1-point"],  # (its child would be the 7th process on this box's one GPU: the limit is 6)
                       capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 5 and out["pipeline"]["samplers"] == 2 and out["pipeline"]["trainers"] == 3
    h = out["pipeline"]["handoff"]
    assert len(h["rings"]) == 2 and h["check_failed"] == 0 and h["check_messages_per_sampler"] == 3
    assert h["verified"] == 6 and h["sent_device"] > 0 and h["spilled"] == 0
    assert h["transport"].startswith("sampler HBM ring")
    assert out["value"] > 0 and out["epoch_time_s"]["with_training"] > 0


@pytest.mark.parametrize("arch,sample_type,workers,cache,mode", [
    ("arch6", "khop2", 2, 0.25, "inline"),       # 2 workers, each samples + extracts its own aligned share
    ("arch6", "khop2", 3, 0.0, "background"),    # train set padded to a multiple of 3; samgraph_extract_start thread
    ("arch6", "random_walk", 1, 0.2, "inline"),
    ("arch7", "khop2", 2, 0.0, "inline"),        # sample only; features through samgraph.torch.load_subtensor
    ("arch7", "weighted_khop_prefix", 3, 0.0, "inline"),
])
def test_sgnn_baseline_archs(tmp_path, arch, sample_type, workers, cache, mode):
    """The reference's SGNN baselines behind the same API: arch6 (dist_loops_arch6.cc: every worker process samples,
    extracts and trains on its GPU) and arch7 (cuda_loops_arch7.cc: every worker runs its own sample-only engine),
    both over DistAlignedShuffler's equal shares (dist_shuffler_aligned.cc); all workers on cuda:0 here."""
    assert "ok" in _run(tmp_path, arch, sample_type, workers, cache, mode)


@pytest.mark.parametrize("slots,mode,args", [
    (8, "arch5", ["khop2", 1, 1, 0.25, "pipeline"]),
    (2, "arch5", ["random_walk", 2, 2, 0.2, "inline"]),   # 2 slots per sampler: some messages spill to the host ring
    (1, "arch5", ["khop2", 2, 1, 0.0, "pipeline"]),
    (4, "arch3", ["khop2", 0.25, "threads"]),
    (4, "arch6", ["khop2", 2, 0.25, "inline"]),
    (4, "switcher", ["khop2"]),
    (0, "arch5", ["khop2", 2, 1, 0.25, "pipeline"]),      # the reference's path: everything through the host ring
    (-1, "arch5", ["khop2", 2, 2, 0.25, "pipeline"]),     # a receiver that cannot map the ring: copied back on request
])
def test_device_ring_handoff(tmp_path, slots, mode, args):
    """SAMGRAPH_DEVICE_RING_SLOTS (default: 16 for arch5, 4 in-process): the message arrays travel through a ring in the
    sampler's HBM that the receiver maps with hipIpcOpenMemHandle (headers still through the host ring); same batches,
    bit for bit.  slots == -1 here: default slots, but the receivers pretend the mapping was refused."""
    import re
    env = {"SAMGRAPH_LOG_LEVEL": "info"}
    if slots >= 0:
        env["SAMGRAPH_DEVICE_RING_SLOTS"] = str(slots)
    else:
        env["SAMGRAPH_DEVICE_RING_FORCE_SPILL"] = "1"
    out = _run(tmp_path, mode, *args, env=env)
    assert "ok" in out
    used = [tuple(int(x) for x in m) for m in
            re.findall(r"device ring \d+: (\d+) messages through HBM, (\d+) through the host ring, (\d+) copied", out)]
    if slots == 0:
        assert not used
    elif slots > 0:
        assert used and all(a > 0 and c == 0 for a, _, c in used), out[-2000:]  # the HBM path really carried messages
    else:
        assert used and sum(c for _, _, c in used) > 0 and "cannot map the sampler's HBM ring" in out, out[-2000:]


@pytest.mark.parametrize("sample_type", ["random_walk", "khop2"])
def test_arch5_switcher(tmp_path, sample_type):
    """BASELINE config 5's switcher flow: with `have_switcher` the sampler ships input nodes, the trainer and the
    sampler's co-located switcher (samgraph_switch_init, own smaller cache) both drain the queue."""
    assert "ok" in _run(tmp_path, "switcher", sample_type)


@pytest.mark.parametrize("extra", [[], ["--arch", "arch3", "--cache-percentage", "0.2", "--pipeline"],
                                   ["--arch", "arch3", "--cache-percentage", "0.2", "--cache-policy", "presample_static"],
                                   ["--arch", "arch4", "--cache-policy", "dynamic_cache", "--sample-type", "khop0",
                                    "--pipeline"],
                                   # the reference's single-process train_gcn.py / train_pinsage.py
                                   ["--model", "gcn", "--sample-type", "khop0", "--fanout", "3", "4", "5"],
                                   ["--model", "pinsage", "--num-random-walk", "6", "--arch", "arch3", "--pipeline",
                                    "--cache-percentage", "0.1"]])
def test_training_example_runs(tmp_path, extra):
    """examples/train_graphsage.py = the reference's single-process script shape (config -> init -> [start] ->
    sample_once -> get_next_batch -> get_dgl_blocks -> fwd/bwd) with a torch-op SAGEConv; prints the reference's
    test_result lines.  Second case: the reference's default arch3 with its background threads."""
    ex = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "train_graphsage.py")
    p = subprocess.run([sys.executable, ex, "--make-dataset", "small", "--dataset-path", str(tmp_path / "small"),
                        "--num-epoch", "2", "--batch-size", "2000"] + ([] if "--fanout" in extra else ["--fanout", "10", "5"])
                       + extra, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "test_result:epoch_time:total=" in p.stdout and "test_result:sampled_edges_per_epoch=" in p.stdout


@pytest.mark.parametrize("model,extra", [
    ("graphsage", ["--cache-percentage", "0.2", "--fanout", "10", "5"]),
    ("gcn", ["--sample-type", "weighted_khop_prefix", "--fanout", "3", "4", "5", "--cache-percentage", "0.1"]),
    ("pinsage", ["--num-random-walk", "6", "--num-sample-worker", "2"]),
    # the reference's multi_gpu/async variant: gradients to one model in shared host memory, no all-reduce
    ("graphsage", ["--async", "--num-train-worker", "2", "--fanout", "10", "5", "--cache-percentage", "0.2"]),
])
def test_fgnn_training_example_runs(tmp_path, model, extra):
    """examples/multi_gpu/train_fgnn.py = the reference's multi_gpu scripts' shape (parent config + data_init, forked
    sampler and trainer processes, pinned queue in between), all workers on cuda:0."""
    ex = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "multi_gpu",
                      "train_fgnn.py")
    p = subprocess.run([sys.executable, ex, "--model", model, "--make-dataset", "small", "--dataset-path",
                        str(tmp_path / "small"), "--num-epoch", "2", "--batch-size", "2000", "--single-gpu"] + extra,
                       capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "test_result:pipeline_train_epoch_time=" in p.stdout and "test_result:sample_time=" in p.stdout


@pytest.mark.parametrize("model,extra", [
    ("pinsage", ["--num-random-walk", "6", "--num-train-worker", "2", "--cache-percentage", "0.2",
                 "--switch-cache-percentage", "0.1"]),
    ("graphsage", ["--fanout", "10", "5", "--cache-percentage", "0.2", "--no-switcher"]),
])
def test_switcher_training_example_runs(tmp_path, model, extra):
    """examples/balance_switcher/train_switcher.py = the reference's balance_switcher scripts' shape (BASELINE config 5):
    a switcher process next to every sampler starts training on the sampler's GPU once the sampler has produced its
    share of the epoch; permits per batch; parameter deltas through one model in shared host memory.  Every batch of
    every epoch is consumed exactly once (the script exits non-zero otherwise); second case: the same without switcher
    processes."""
    ex = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "balance_switcher",
                      "train_switcher.py")
    p = subprocess.run([sys.executable, ex, "--model", model, "--make-dataset", "small", "--dataset-path",
                        str(tmp_path / "small"), "--num-epoch", "2", "--batch-size", "2000", "--single-gpu"] + extra,
                       capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    consumed = int(re.search(r"test_result:batches_consumed=(\d+)", p.stdout).group(1))
    switched = int(re.search(r"test_result:batches_by_switchers=(\d+)", p.stdout).group(1))
    assert consumed == 3 * 20 and "test_result:epoch_time:total=" in p.stdout  # (2 + the warm-up epoch) x 40000 / 2000
    assert (switched == 0) == ("--no-switcher" in extra), p.stdout[-2000:]
    losses = [float(x) for x in re.findall(r"loss ([0-9.]+)", p.stdout)]
    assert losses and all(x == x and x < 20 for x in losses), losses


@pytest.mark.parametrize("extra", [["--arch", "arch6", "--num-worker", "2", "--single-gpu", "--cache-percentage", "0.2"],
                                   ["--arch", "arch7", "--num-worker", "1"]])
def test_sgnn_example_runs(tmp_path, extra):
    """examples/sgnn/train_sgnn.py = the reference's sgnn/ (arch6) and sgnn_dgl/ (arch7) script shape: one process per
    worker doing sample + extract + train, gradients synchronised across workers."""
    ex = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "sgnn", "train_sgnn.py")
    p = subprocess.run([sys.executable, ex, "--make-dataset", "small", "--dataset-path", str(tmp_path / "small"),
                        "--num-epoch", "2", "--batch-size", "2000", "--fanout", "10", "5"] + extra,
                       capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "test_result:epoch_time:total=" in p.stdout and "test_result:sampled_edges_per_epoch_per_worker=" in p.stdout
