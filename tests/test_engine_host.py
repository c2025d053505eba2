"""CPU-only tests of the engine library (samgraph/torch/c_lib.so): exported ABI, config parsing and its
abort-on-error behaviour, the shuffler's exact permutation, the wire-format sizes, and the multi-process
hand-off ring (the N>1 path of the sampler->trainer pipeline has no collective: it is this queue)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENG = os.path.join(ROOT, "fgnn-artifacts_amd", "samgraph", "torch", "c_lib.so")
HOOKS = os.path.join(ROOT, "fgnn-artifacts_amd", "samgraph", "torch", "fgnn_engine_hooks.so")


@pytest.fixture(scope="module")
def eng():
    if not os.path.exists(ENG):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "fgnn-artifacts_amd", "csrc")])
    return C.CDLL(ENG)


@pytest.fixture(scope="module")
def hooks():
    """the host-only test entry points (include/fgnn_engine_hooks.h): a library of their own over the engine's object
    files -- c_lib.so itself exports the reference's symbol list and nothing else"""
    if not os.path.exists(HOOKS):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "fgnn-artifacts_amd", "csrc")])
    return C.CDLL(HOOKS)


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:fgnn|samgraph)_[a-z0-9_]+)\s*\(", txt)))


def test_engine_exports_reference_abi(eng, hooks):
    names = _declared("samgraph.h")
    # the 38 functions the reference defines (operation.cc) + the 9 tensor getters (adapter.h:29-42)
    assert len([n for n in names if not n.startswith("samgraph_torch_")]) == 38
    assert len([n for n in names if n.startswith("samgraph_torch_")]) == 9
    missing = [n for n in names if not hasattr(eng, n)] + \
              [n for n in _declared("fgnn_engine_hooks.h") if not hasattr(hooks, n)]
    assert not missing, missing


def test_engine_library_exports_only_the_reference_symbol_list():
    """The reference links its extension with samgraph.lds (`*samgraph_*`, `*PyInit*`, `*initc_lib*` global, everything
    else local): c_lib.so must not leak the engine's C++ symbols (two differently built engines in one process)."""
    out = subprocess.run(["nm", "-D", "--defined-only", ENG], capture_output=True, text=True, check=True).stdout
    syms = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    assert syms and all(s.startswith("samgraph_") or s == "PyInit_c_lib" for s in syms), \
        [s for s in syms if not s.startswith("samgraph_")][:10]
    # the reference's list, plus the entry points of include/samgraph_ext.h (same prefix, documented as this build's)
    assert sorted(s for s in syms if s != "PyInit_c_lib") == sorted(_declared("samgraph.h") + _declared("samgraph_ext.h"))
    assert _declared("samgraph_ext.h") == ["samgraph_ext_queue_stats", "samgraph_ext_ring_mapping"]


def test_python_binding_covers_abi():
    import samgraph.common as sc
    want = {n for n in _declared("samgraph.h") if not n.startswith("samgraph_torch_")}
    assert set(sc._SIGNATURES) == want


def _cfg(**over):
    cfg = dict(dataset_path="/nonexistent", _arch=5, _sample_type=5, batch_size=8000, num_epoch=3, _cache_policy=2,
               cache_percentage=0.2, max_sampling_jobs=10, max_copying_jobs=10, omp_thread_num=40,
               num_sample_worker=2, num_train_worker=6, num_fanout=2, fanout="25 10", unknown_key="ignored")
    cfg.update(over)
    return cfg


def _probe(hooks, cfg):
    keys = [str(k).encode() for k in cfg]
    vals = [str(v).encode() for v in cfg.values()]
    out = (C.c_size_t * 4)()
    rc = hooks.fgnn_host_config_probe((C.c_char_p * len(keys))(*keys), (C.c_char_p * len(vals))(*vals),
                                    C.c_size_t(len(keys)), out)
    return rc, list(out)


def test_config_parse(hooks):
    eng = hooks
    rc, out = _probe(eng, _cfg())
    assert rc == 0 and out == [2, 25, 5, 1]
    rc, out = _probe(eng, _cfg(_arch=1, sampler_ctx="cuda:0", trainer_ctx="cuda:0"))
    assert out == [2, 25, 1, 0]  # arch1 never uses the cache (run_config.h:84-86)
    rw = _cfg(_sample_type=3, random_walk_length=3, random_walk_restart_prob=0.5, num_random_walk=4, num_neighbor=5,
              num_layer=3)
    del rw["fanout"], rw["num_fanout"]
    assert _probe(eng, rw)[1][:2] == [3, 5]
    # the SGNN baselines' keys (operation.cc:104-118)
    a6 = _cfg(_arch=6, num_worker=4)
    assert _probe(eng, a6)[1] == [2, 25, 6, 1]
    a7 = _cfg(_arch=7, num_worker=4, worker_id=3, sampler_ctx="cuda:3", trainer_ctx="cuda:3", cache_percentage=0.0)
    assert _probe(eng, a7)[1] == [2, 25, 7, 0]


@pytest.mark.parametrize("bad", [dict(drop="batch_size"), dict(_arch=6), dict(_arch=8), dict(_arch=0), dict(_arch=7, num_worker=2, worker_id=2, sampler_ctx='cuda:0', trainer_ctx='cuda:0'), dict(_sample_type=7), dict(drop="fanout"),
                                 dict(_arch=1), dict(_sample_type=6, fanout="60 5")])
def test_config_errors_abort(bad):
    """No return codes: a violated check prints file:line and abort()s (logging.h:32-45)."""
    cfg = _cfg()
    if "drop" in bad:
        del cfg[bad.pop("drop")]
    cfg.update(bad)
    code = ("import ctypes as C\nL=C.CDLL(%r)\ncfg=%r\nk=[str(x).encode() for x in cfg]\n"
            "v=[str(x).encode() for x in cfg.values()]\no=(C.c_size_t*4)()\n"
            "L.fgnn_host_config_probe((C.c_char_p*len(k))(*k),(C.c_char_p*len(v))(*v),C.c_size_t(len(k)),o)\n"
            % (HOOKS, cfg))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True)
    assert p.returncode == -6, p.stderr  # SIGABRT
    assert b"eng_config.cc:" in p.stderr


def test_shuffle_matches_reference_permutation(hooks, oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "shuffle.npz"))
    for key in g.files:
        n = int(key[1:])
        data = np.arange(n, dtype=np.uint32)
        for epoch, want in enumerate(g[key]):
            hooks.fgnn_host_shuffle_minstd0(data.ctypes.data_as(C.c_void_p), C.c_size_t(n), C.c_uint64(epoch))
            np.testing.assert_array_equal(data, want)
    data = np.arange(100003, dtype=np.uint32)
    hooks.fgnn_host_shuffle_minstd0(data.ctypes.data_as(C.c_void_p), C.c_size_t(len(data)), C.c_uint64(7))
    np.testing.assert_array_equal(data, oracle.shuffle_minstd0(np.arange(100003, dtype=np.uint32), 7))


def test_shuffler_partitions_match_oracle(hooks, oracle):
    """Both shufflers' splits (dist_shuffler.cc:47-79, dist_shuffler_aligned.cc:45-71) against the oracle's restatement."""
    out = (C.c_size_t * 7)()
    for n, b, ns in [(1207179, 8000, 1), (1207179, 8000, 2), (1207179, 8000, 8), (196615, 8000, 3), (1003, 100, 4),
                     (16000, 8000, 2), (7, 3, 2), (64, 8, 8)]:
        for sid in range(ns):
            hooks.fgnn_host_shuffler_partition(C.c_size_t(n), C.c_size_t(b), sid, ns, 0, out)
            want = oracle.dist_shuffler_partition(n, b, sid, ns)
            assert (out[0], out[1], out[2], out[3], out[4], out[5], out[6]) == (
                n, want["local_data_size"], want["num_local_step"], want["epoch_step"], want["dataset_offset"] // b,
                want["dataset_offset"], want["last_batch_size"])
            hooks.fgnn_host_shuffler_partition(C.c_size_t(n), C.c_size_t(b), sid, ns, 1, out)
            want = oracle.aligned_shuffler_partition(n, b, sid, ns)
            assert list(out) == [want[k] for k in ("padded_size", "local_data_size", "num_local_step", "epoch_step",
                                                   "step_offset", "dataset_offset", "last_batch_size")]


def test_wire_sizes(hooks):
    out = (C.c_size_t * 3)()
    fan = (C.c_size_t * 2)(25, 10)
    hooks.fgnn_host_wire_sizes(C.c_size_t(8000), fan, C.c_size_t(2), 0, out)
    assert out[0] == 40 and out[1] == 24  # sizeof(TransData), sizeof(GraphData) on LP64 (task_queue.cc:68-88)
    # 2 GraphData + 2*(80 000 + 2 200 000) edge words + 8000 output ids + 3 * 2 288 000 node words
    assert out[2] >= 40 + 2 * 24 + 8 * (80000 + 2200000) + 4 * 8000 + 12 * 2288000
    assert out[2] < 50 * 1024 * 1024  # fits the reference's hard-coded 50 MiB slot (task_queue.cc:32)


@pytest.mark.parametrize("producers,consumers,slots", [(1, 1, 2), (2, 1, 3), (2, 3, 4), (1, 2, 170)])
def test_queue_multiprocess(hooks, producers, consumers, slots):
    """world_size > 1 on CPU: forked writers and readers over the shared ring; every message exactly once."""
    rc = hooks.fgnn_host_queue_selftest(C.c_size_t(slots), C.c_size_t(4096), C.c_size_t(500), producers, consumers)
    assert rc == 0


@pytest.mark.parametrize("producers,consumers,slots,depth", [(1, 1, 2, 4), (1, 2, 3, 4), (2, 3, 4, 4), (2, 2, 170, 4),
                                                             (1, 3, 2, 2), (3, 4, 2, 4), (4, 6, 3, 4)])
def test_queue_deep_receivers_never_deadlock(hooks, producers, consumers, slots, depth):
    """The extraction thread keeps up to 4 batches in flight (unreleased queue slots).  A slot counts as "sent" from the
    moment it is claimed, so asking `send_cnt - recv_cnt` whether a message waits would block a receiver on a message
    whose sender waits for one of the receiver's own slots (2 slots, 1 trainer: certain).  TryRecv answers only for
    PUBLISHED messages; every message still arrives exactly once.  More receivers (x depth) than slots with several
    senders is also where a semaphore per slot would hand message k + N's post to the waiter for message k (and the
    release of k to the sender of k + N): the ring's hand-shakes are per-slot sequence numbers (eng_queue.cc)."""
    rc = hooks.fgnn_host_queue_selftest_deep(C.c_size_t(slots), C.c_size_t(4096), C.c_size_t(600), producers, consumers,
                                           depth)
    assert rc == 0


_NAMED_ROLE = r"""
import ctypes as C, sys
eng = C.CDLL(sys.argv[1])
role, index, peers, slots, messages = map(int, sys.argv[2:7])
sys.exit(eng.fgnn_host_queue_named_role(C.c_size_t(slots), C.c_size_t(4096), C.c_size_t(messages), role, index, peers))
"""


def test_queue_waiters_give_up_when_the_queue_is_aborted(hooks):
    """A dead peer must not hang the job: samgraph_wait_one_child marks the queue aborted, and processes blocked in
    GetPtr (ring full) or Recv (ring empty) then abort like a failed CHECK instead of waiting forever (the
    reference's sem_wait pairs, memory_queue.cc:104-138, never return in that case)."""
    assert hooks.fgnn_host_queue_abort_selftest() == 2


@pytest.mark.parametrize("producers,consumers,slots", [(1, 1, 2), (2, 3, 5)])
def test_queue_named_regions_between_unrelated_processes(hooks, producers, consumers, slots):
    """The torchrun launch style: processes that share no forking parent meet in named shared-memory regions
    (SAMGRAPH_SHM_PREFIX); whoever comes first creates and initialises the ring, the others wait for it."""
    prefix = "fgnn_test_%d_%d%d" % (os.getpid(), producers, consumers)
    env = dict(os.environ, SAMGRAPH_SHM_PREFIX=prefix, SAMGRAPH_SHM_KEEP="1")  # the test removes the names
    procs = []
    for role, peers in ((1, consumers), (0, producers)):  # consumers first: they must wait for a creator either way
        for i in range(peers):
            procs.append(subprocess.Popen([sys.executable, "-c", _NAMED_ROLE, HOOKS, str(role), str(i), str(peers),
                                           str(slots), "300"], env=env))
    try:
        assert [p.wait(timeout=120) for p in procs] == [0] * len(procs)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for f in os.listdir("/dev/shm"):
            if f.startswith(prefix):
                os.unlink(os.path.join("/dev/shm", f))
    assert not [f for f in os.listdir("/dev/shm") if f.startswith(prefix)]


_PROFILER_SCRIPT = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
from fgnn_hip import synth
import samgraph.torch as sam  # imports torch + the engine library; no GPU call below
wd = sys.argv[2]
d = synth.write_dataset(wd, "g", 2000, 20000, 4, 5, 300, 20, 20, seed=3)
sam.config(dict(dataset_path=d, _arch=sam.kArch5, _sample_type=sam.kKHop2, batch_size=100, num_epoch=2,
                _cache_policy=sam.kCacheByPreSample, cache_percentage=0.1, max_sampling_jobs=1, max_copying_jobs=1,
                omp_thread_num=1, num_sample_worker=1, num_train_worker=1, num_fanout=2, fanout=[5, 3]))
sam.data_init()                       # arch5: dataset + shared queue, no GPU touched
assert sam.steps_per_epoch() == 3 and sam.num_epoch() == 2
for step in range(3):
    sam.log_step(1, step, sam.kLogL1SampleTime, 0.5 + step)
    sam.log_step(1, step, sam.kLogL1FeatureBytes, 3 * 1024 * 1024)
    sam.log_step(1, step, sam.kLogL2CacheCopyTime, 0.25)
    sam.log_step(1, step, sam.kLogL3CacheCombineMissTime, 0.125)
    sam.log_epoch_add(1, sam.kLogEpochSampleTime, 0.5 + step)
assert sam.get_log_step_value(1, 2, sam.kLogL1SampleTime) == 2.5
assert sam.get_log_epoch_value(1, sam.kLogEpochSampleTime) == 4.5
sam.report_init()
sam.report_step(1, 2)
sam.report_step_average(1, 2)
sam.report_epoch(1)
sam.report_epoch_average(1)
key = 1 * 3 + 2
sam.trace_step_begin(key, sam.kL1Event_Train, 1000)
sam.trace_step_end(key, sam.kL1Event_Train, 1500)
sam.trace_step_begin(key, sam.kL3Event_Copy_CacheCopy_CombineMiss, 1100)
sam.trace_step_end(key, sam.kL3Event_Copy_CacheCopy_CombineMiss, 1200)
sam.trace_step_begin_now(key, sam.kL0Event_Train_Step)
sam.trace_step_end_now(key, sam.kL0Event_Train_Step)
sam.trace_step_begin(key + 1, sam.kL1Event_Copy, 2000)   # never ended: skipped with a warning
sam.dump_trace()
sys.stdout.flush()
"""


@pytest.mark.parametrize("level", [1, 3])
def test_profiler_reports_and_trace_dump(tmp_path, level):
    """samgraph_report_* gated by SAMGRAPH_PROFILE_LEVEL with the reference's item names (profiler.cc:371-557), and
    samgraph_dump_trace writing the Chrome trace JSON of profiler.cc:286-364 (B/E pairs named <item>-<key>) -- to the
    file SAMGRAPH_DUMP_TRACE names.  Runs through samgraph.torch with arch5's data_init only: no GPU needed."""
    import json
    trace = tmp_path / "trace.json"
    env = dict(os.environ, SAMGRAPH_PROFILE_LEVEL=str(level), SAMGRAPH_DUMP_TRACE=str(trace))
    p = subprocess.run([sys.executable, "-c", _PROFILER_SCRIPT, os.path.join(ROOT, "fgnn-artifacts_amd"), str(tmp_path)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    out = p.stdout
    assert "[Step(profile) Profiler Level 1 E1 S2]" in out and "SampleTime 2.5000" in out
    assert "FeatureBytes 3.00 MB" in out
    assert "[Step(average) Profiler Level 1 E1 S2]" in out
    assert "[Init Profiler Level 1]" in out and "[Epoch(profile) E1]" in out and "SampleTime 4.5000" in out
    assert ("Profiler Level 2 E1 S2]" in out) == (level >= 2) and ("CacheCopyTime 0.2500" in out) == (level >= 2)
    assert ("CacheCombineMissTime 0.1250" in out) == (level >= 3)
    ev = json.loads(trace.read_text())
    names = [(e["name"], e["ph"], e["tid"]) for e in ev]
    assert ("kL1Event_Train-5", "B", 3) in names and ("kL1Event_Train-5", "E", 3) in names
    assert ("kL3Event_Copy_CacheCopy_CombineMiss-5", "B", 2) in names and ("kL0Event_Train_Step-5", "E", 0) in names
    assert not [n for n in names if n[0].startswith("kL1Event_Copy-")] and "without end" in p.stderr
    t = {(e["name"], e["ph"]): e["ts"] for e in ev}
    assert t[("kL1Event_Train-5", "B")] == 1000 and t[("kL1Event_Train-5", "E")] == 1500


def test_fused_sage_layer_loads_the_op_by_op_checkpoint_layout():
    """examples/models.py: a state dict in the reference's layout (fc_self / fc_neigh per layer: dgl.nn.SAGEConv, this
    repo's SAGEConvMean) loads into the default fused layers, and back (advisor, round 4); same outputs on the CPU path"""
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "examples"))
    import models

    class Block:
        def __init__(self, row, col, ndst):
            self.row, self.col, self.ndst = row, col, ndst

        def number_of_dst_nodes(self):
            return self.ndst

    torch.manual_seed(0)
    ref = models.SAGE(12, 16, 5, 2, 0.0, fused=False)
    fused = models.SAGE(12, 16, 5, 2, 0.0, fused=True)
    fused.load_state_dict(ref.state_dict())
    b0 = Block(torch.randint(0, 40, (200,)), torch.sort(torch.randint(0, 20, (200,)))[0], 20)
    b1 = Block(torch.randint(0, 20, (60,)), torch.sort(torch.randint(0, 7, (60,)))[0], 7)
    x = torch.randn(40, 12)
    ref.eval()
    fused.eval()
    torch.testing.assert_close(fused([b0, b1], x), ref([b0, b1], x), rtol=1e-5, atol=1e-6)
    back = {}
    for i, layer in enumerate(fused.layers):
        back.update(layer.split_state_dict("layers.%d." % i))
    ref2 = models.SAGE(12, 16, 5, 2, 0.0, fused=False)
    ref2.load_state_dict(back)
    torch.testing.assert_close(ref2([b0, b1], x), ref([b0, b1], x))
    # DGL's two checkpoint layouts (advisor, round 5): < 0.8 keeps a bias in BOTH linear maps -- the layer adds their sum
    # --, >= 0.8 bias-free maps plus a separate `bias`; both load strictly and give the reference's outputs
    sd = ref.state_dict()
    old, new = {}, {}
    for k, v in sd.items():
        if k.endswith("fc_neigh.bias"):
            half = torch.randn_like(v)
            old[k], old[k.replace("fc_neigh.bias", "fc_self.bias")] = v - half, half
            new[k.replace("fc_neigh.bias", "bias")] = v
        else:
            old[k] = new[k] = v
    for layout in (old, new):
        m = models.SAGE(12, 16, 5, 2, 0.0, fused=True)
        m.load_state_dict(dict(layout), strict=True)
        m.eval()
        torch.testing.assert_close(m([b0, b1], x), ref([b0, b1], x), rtol=1e-5, atol=1e-6)


def test_flip_word_hook_finds_the_ring_among_a_jobs_regions_and_corrupts_one_published_message(tmp_path):
    """fgnn_host_queue_flip_word (the outside half of the hand-off check's failure test, tests/test_engine_gpu.py): in a
    subprocess with named regions, a ring is opened after another region, three messages are published, the hook -- tried
    on every region of the job -- refuses the region that is not a ring and the message that does not exist, flips
    bits of message 1's second word, and the receiver sees exactly that message changed."""
    code = r"""
import ctypes as C, os, sys
eng = C.CDLL(sys.argv[1])
eng.fgnn_host_queue_open.restype = C.c_void_p
prefix = os.environ["SAMGRAPH_SHM_PREFIX"]
first = C.c_void_p(eng.fgnn_host_queue_open(C.c_size_t(2), C.c_size_t(256)))      # region .0: a ring of another geometry
q = C.c_void_p(eng.fgnn_host_queue_open(C.c_size_t(4), C.c_size_t(4096)))          # region .1: the ring under test
for k in range(3):
    eng.fgnn_host_queue_send(q, C.c_uint64(100 + k), C.c_uint64(7))
name = lambda i: ("/%s.%d" % (prefix, i)).encode()
assert eng.fgnn_host_queue_flip_word(name(5), C.c_size_t(1), C.c_size_t(2)) == 2          # no such region
assert eng.fgnn_host_queue_flip_word(name(0), C.c_size_t(1), C.c_size_t(2)) == 1          # a ring, but message 1 was never sent there
assert eng.fgnn_host_queue_flip_word(name(1), C.c_size_t(3), C.c_size_t(2)) == 1          # no message 3
assert eng.fgnn_host_queue_flip_word(name(1), C.c_size_t(1), C.c_size_t(1 << 20)) == 1    # beyond the slot
assert eng.fgnn_host_queue_flip_word(name(1), C.c_size_t(1), C.c_size_t(2)) == 0          # low half of message 1's value
got = []
for _ in range(3):
    k, v = C.c_uint64(), C.c_uint64()
    eng.fgnn_host_queue_recv(q, C.byref(k), C.byref(v))
    got.append((k.value, v.value))
assert got == [(100, 7), (101, 7 ^ 0x5A5A5A5A), (102, 7)], got
print("ok")
"""
    prefix = "fgnn_test_%d_flipcpu" % os.getpid()
    env = dict(os.environ, SAMGRAPH_SHM_PREFIX=prefix, SAMGRAPH_SHM_KEEP="1")
    try:
        p = subprocess.run([sys.executable, "-c", code, HOOKS], env=env, capture_output=True, text=True, timeout=120)
        assert p.returncode == 0 and "ok" in p.stdout, p.stdout + p.stderr
    finally:
        for f in os.listdir("/dev/shm"):
            if f.startswith(prefix):
                os.unlink(os.path.join("/dev/shm", f))
