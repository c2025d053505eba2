"""Forward progress of the single-pass kernels under adversarial co-residency.

The k-hop sampler, the dedup count+assign and the cache split each learn their output offset from the workgroups "before"
them inside ONE launch.  A workgroup that has polled a predecessor's descriptor longer than a healthy launch ever takes
stops waiting and recomputes the missing aggregate itself (fgnn_device.h:scan_prefix_help), so every resident
workgroup terminates whatever else is on the GPU.  Two tests:
  * hostile co-residency: three streams of overlapping batches plus a foreign kernel that parks itself on 3/4 of the
    chip's wave slots for 1.5 ms at a time, 2 000 batches -- no batch flagged, every batch bit-identical to the same
    batches run one at a time on an idle GPU (which the parity tests compare with the oracle);
  * the helping path itself, forced: fgnn_debug_set_scan_help_after(0) (lib.py calls it when FGNN_SCAN_HELP_AFTER=0 is in
    the environment; the library itself reads no variable) makes every wait that is not satisfied by its first poll
    recompute, and the batch-driver parity tests (oracle, bit-exact) plus the run above must still hold, with the
    help counter showing that the path ran.
The reference never loses a batch (cuda_loops.cc:50-267)."""
import ctypes as C
import os
import subprocess
import sys
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SEED = 0x5A4D47
NUM_BATCHES = 2000


def _digest(bt, m, num_layers):
    """order-sensitive 64-bit digest of everything the batch produced (blocks, node list, cache split)"""
    parts = [bt.input_nodes()] + bt.cache_index_arrays()
    head = [int(m.num_input), int(m.num_miss), int(m.num_cache)]
    for l in range(num_layers):
        row, col, nsrc, ndst = bt.graph(l)
        parts += [row, col]
        head += [len(row), nsrc, ndst]
    flat = torch.cat([p.to(torch.int64) & 0xFFFFFFFF for p in parts])
    w = torch.arange(1, flat.numel() + 1, device=flat.device, dtype=torch.int64)
    return tuple(head) + (int((flat * (w * 0x9E3779B1 + 7)).sum().item()),)


def _run(hip, indptr, indices, table, train, fanouts, bs, n_streams, tenant, num_batches=None):
    dev = indptr.device
    d_indices = indices.clone()  # khop2 swaps entries in place: every run starts from the same CSR
    sampler = hip.Sampler(indptr, d_indices, fanouts, bs, sample_type=hip.KHOP2, seed=SEED)
    NUM_BATCHES = num_batches or globals()["NUM_BATCHES"]
    nbuf = 2 * n_streams
    batches = [sampler.new_batch(0, hip.F32, hip.I64) for _ in range(nbuf)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    torch.cuda.synchronize()  # the CSR copy above ran on the default stream; the batch streams do not wait for it
    steps = train.numel() // bs
    stop = threading.Event()
    launched = [0]

    def foreign():
        torch.cuda.set_device(dev)
        L = hip.load()
        fs = torch.cuda.Stream(device=dev)
        ev = torch.cuda.Event()
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        wgs = cus * 8 * 3 // 4  # 8 workgroups of 4 waves fill a CU's wave slots
        while not stop.is_set():
            for _ in range(2):
                assert L.fgnn_debug_occupy(C.c_size_t(wgs), C.c_uint(1500), C.c_void_p(fs.cuda_stream)) == 0
                launched[0] += 1
            ev.record(fs)
            ev.synchronize()

    th = None
    if tenant:
        th = threading.Thread(target=foreign)
        th.start()
    digests = []
    try:
        def collect(i):
            bt = batches[i % nbuf]
            m = bt.wait()  # raises on a flagged batch
            digests.append(_digest(bt, m, len(fanouts)))

        for i in range(NUM_BATCHES):
            if i >= nbuf:
                collect(i - nbuf)
            step = i % steps
            sampler.run_batch(i, train[step * bs:(step + 1) * bs], i, batches[i % nbuf], table, None, None,
                              stream=streams[i % n_streams])
        for i in range(max(0, NUM_BATCHES - nbuf), NUM_BATCHES):
            collect(i)
    finally:
        stop.set()
        if th:
            th.join()
    torch.cuda.synchronize()
    return digests, d_indices, launched[0]


def test_single_pass_kernels_under_coresidency():
    from fgnn_hip import lib as hip, rmat
    hip.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    num_node, num_edge, bs, fanouts = 400_000, 8_000_000, 2000, [25, 10]
    indptr, indices, _ = rmat.rmat_csr(num_node, num_edge, 42, dev)
    train = rmat.train_set(num_node, 100_000, 1, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    table = torch.full((num_node,), -1, dtype=torch.int32, device=dev)
    cached = torch.randperm(num_node, generator=g, device=dev)[:num_node // 5]
    table[cached] = torch.arange(cached.numel(), device=dev, dtype=torch.int32)

    serial, csr_serial, _ = _run(hip, indptr, indices, table, train, fanouts, bs, 1, tenant=False)
    shared, csr_shared, launched = _run(hip, indptr, indices, table, train, fanouts, bs, 3, tenant=True)
    assert launched >= 4, "the foreign kernel never ran beside the batches"
    assert len(serial) == len(shared) == NUM_BATCHES
    bad = [i for i, (a, b) in enumerate(zip(serial, shared)) if a != b]
    assert not bad, "batches %s differ between the idle and the shared GPU" % bad[:10]
    assert torch.equal(csr_serial, csr_shared)  # khop2's CSR mutations were applied in batch order
    # frontiers are large enough for multi-tile prefixes in all three kernels
    assert serial[0][0] > 30_000
    if os.environ.get("FGNN_SCAN_HELP_AFTER") == "0":
        helps = int(hip.load().fgnn_debug_scan_helps())
        print("helped tiles:", helps)
        assert helps > 1000, "the helping path was not exercised"


def _run_range(hip, indptr, indices, table, train, fanouts, bs, n_streams, num_batches, chunk):
    """the same batches through fgnn_sampler_run_range (the native batch loop), `chunk` batches per call"""
    dev = indptr.device
    d_indices = indices.clone()
    sampler = hip.Sampler(indptr, d_indices, fanouts, bs, sample_type=hip.KHOP2, seed=SEED)
    nbuf = 2 * n_streams
    batches = [sampler.new_batch(0, hip.F32, hip.I64) for _ in range(nbuf)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    torch.cuda.synchronize()  # (the CSR copy ran on the default stream)
    nt = train.numel() // bs * bs  # whole steps only, like _run
    digests = []
    for first in range(0, num_batches, chunk):
        n = min(chunk, num_batches - first)
        # a range's summaries come back together; the buffers hold its last `nbuf` batches
        metas, _, busy = sampler.run_range(first, n, train[:nt], bs, batches, streams, cache_table=table)
        assert busy > 0 and len(metas) == n
        for i in range(max(0, n - nbuf), n):
            bt = batches[(first + i) % nbuf]
            digests.append((first + i, _digest(bt, metas[i], len(fanouts))))
        for i, m in enumerate(metas):
            assert int(m.key) == (first + i) % (nt // bs)
    torch.cuda.synchronize()
    return dict(digests), d_indices


@pytest.mark.parametrize("n_streams", [1, 2, 3, 4, 5, 6])
def test_any_number_of_streams_gives_the_serial_result(n_streams):
    """khop2's cross-batch order (in-place CSR swaps, cuda_sampling_khop2.cu:74-83) and the reuse of the sampler's six
    slots are ordered by the stream where a slot (a predecessor) comes back on the stream it ran on, by events where it
    does not -- events that are only recorded where the last hand-over needed one (engine.hip, expect_cross / csr_cross).  1, 2, 3
    and 6 streams never need a slot event, 4 and 5 do; every count above 1 needs the CSR event.  400 overlapping batches
    must equal the same batches on one stream, digest by digest, and leave the same CSR -- through the per-batch call
    from Python and through the native batch loop (fgnn_sampler_run_range) in ranges of 1, 7 and 64 batches."""
    from fgnn_hip import lib as hip, rmat
    hip.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    num_node, num_edge, bs, fanouts = 400_000, 8_000_000, 2000, [25, 10]
    indptr, indices, _ = rmat.rmat_csr(num_node, num_edge, 42, dev)
    train = rmat.train_set(num_node, 100_000, 1, dev)
    table = torch.full((num_node,), -1, dtype=torch.int32, device=dev)
    table[::3] = torch.arange((num_node + 2) // 3, device=dev, dtype=torch.int32)
    nb = 400
    serial, csr_serial, _ = _run(hip, indptr, indices, table, train, fanouts, bs, 1, False, num_batches=nb)
    other, csr_other, _ = _run(hip, indptr, indices, table, train, fanouts, bs, n_streams, False, num_batches=nb)
    # _run numbers the batch keys 0, 1, 2, ... while the native loop uses the step inside the epoch: compare the native
    # loop with itself on one stream, and the Python loop with itself
    bad = [i for i, (a, b) in enumerate(zip(serial, other)) if a != b]
    assert not bad, "batches %s differ with %d streams" % (bad[:10], n_streams)
    assert torch.equal(csr_serial, csr_other)
    ref, csr_ref = _run_range(hip, indptr, indices, table, train, fanouts, bs, 1, nb, 1)  # every batch digested
    assert len(ref) == nb
    for chunk in (1, 7, 64):
        got, csr_got = _run_range(hip, indptr, indices, table, train, fanouts, bs, n_streams, nb, chunk)
        bad = [i for i in got if got[i] != ref.get(i, got[i])]
        assert not bad, "native loop: batches %s differ with %d streams, ranges of %d" % (bad[:10], n_streams, chunk)
        assert torch.equal(csr_ref, csr_got)


def test_helping_path_is_exact():
    if os.environ.get("FGNN_SCAN_HELP_AFTER") is not None:
        pytest.skip("already inside the forced-help run")
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-s",
                        os.path.join(here, "test_hip_parity.py"), "-k",
                        "batch_driver or layered_pipeline or cache_split or run_batch_cached or coresidency",
                        os.path.join(here, "test_coresidency_gpu.py") + "::test_single_pass_kernels_under_coresidency"],
                       capture_output=True, text=True, timeout=1500, env=dict(os.environ, FGNN_SCAN_HELP_AFTER="0"))
    assert p.returncode == 0, p.stdout[-4000:] + p.stderr[-2000:]
    assert "helped tiles:" in p.stdout
