"""benchlib.common -- constants, workloads, synthetic inputs and small helpers shared by bench.py's halves."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(ROOT, "bench.py")  # the entry script: rank / child processes are started as `python bench.py ...`
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
HOST_LINK_GBS = 64.0    # PCIe Gen5 x16 per direction (secondary bound of the miss rows)
XGMI_LINK_GBS = 153.0   # one xGMI link (sampler -> trainer peer reads)
QUEUE_SLOTS = 170       # messages the shared queue holds at most (mq_size, memory_queue.h:46 = eng_queue.h kMaxSlots)

import torch  # noqa: E402

from fgnn_hip import lib, synth  # noqa: E402

WORKLOADS = {
    # name: shape + run config (reference defaults: batch 8000, common_config.py:63; fanout train_graphsage.py:77)
    "papers100M": dict(**synth.DATASET_SHAPES["papers100M"], fanout=[25, 10], batch_size=8000, sample_type="khop2"),
    "products": dict(**synth.DATASET_SHAPES["products"], fanout=[25, 10], batch_size=8000, sample_type="khop2"),
    # BASELINE.json config 4's sampler side: GCN with weighted sampling (multi_gpu/train_gcn.py:72 fanout [5,10,15])
    "twitter": dict(**synth.DATASET_SHAPES["twitter"], fanout=[5, 10, 15], batch_size=8000,
                    sample_type="weighted_khop_prefix"),
    # config 5's sampler side: PinSAGE random walks (multi_gpu/train_pinsage.py:130-134 with num_walks = 25)
    "uk-2006-05": dict(**synth.DATASET_SHAPES["uk-2006-05"], fanout=[5, 5, 5], batch_size=8000,
                       sample_type="random_walk", walk_len=3, num_walks=25, restart_prob=0.5),
    "small": dict(num_node=1_000_000, num_edge=20_000_000, feat_dim=128, num_class=47, num_train=100_000,
                  fanout=[25, 10], batch_size=8000, sample_type="khop2"),
}
SAMPLE_TYPES = {"khop0": lib.KHOP0, "khop1": lib.KHOP1, "khop2": lib.KHOP2, "weighted_khop_prefix": lib.WEIGHTED_KHOP_PREFIX,
                "random_walk": lib.RANDOM_WALK, "weighted_khop": lib.WEIGHTED_KHOP,
                "weighted_khop_hash_dedup": lib.WEIGHTED_KHOP_HASH_DEDUP}


GRAPH_DESC = {"rmat": "R-MAT (0.57,0.19,0.19,0.05) seed 42, directed, de-duplicated, CSR by destination",
              "powerlaw": "power-law degrees, hub-skewed ids (round 1 generator)"}


class no_gc:
    """Timed regions run with Python's cyclic garbage collector off (collected right before): a generation-2 pass over
    the process's objects took 40-60 ms when it fell into a 64-batch region (the extract leg read 0.75-0.92 instead of
    0.20 ms per batch for some --steps values and not for others: which allocation crosses the collector's threshold is
    a function of everything allocated before; profiles/r05_i_gc_pause.txt)."""

    def __enter__(self):
        import gc
        gc.collect()
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()


def gen_alias_on_gpu(indices, total, seed, device):
    """prob_table f32[E] / alias_table u32[E] (node ids) for the alias-method samplers: random acceptance
    probabilities, alias = the row neighbour one position further (any node id is a valid table entry) -- same memory
    behaviour as a real table; bit-exact parity with the oracle is covered by the tests, not by the bench."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    prob = torch.empty(total, dtype=torch.float32, device=device)
    chunk = 1 << 27
    for a in range(0, total, chunk):
        prob[a:a + chunk] = torch.rand(min(chunk, total - a), generator=g, device=device)
    alias = torch.roll(indices, 1)
    return prob, alias


def gen_prefix_on_gpu(indptr, total, seed, device):
    """prob_prefix_table (f32[E], per-row inclusive prefix sums of random edge weights), built in row chunks."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty(total, dtype=torch.float32, device=device)
    ip = indptr.to(torch.int64) & 0xFFFFFFFF
    n = ip.numel() - 1
    rows_per = 1 << 22
    for r0 in range(0, n, rows_per):
        r1 = min(n, r0 + rows_per)
        a, b = int(ip[r0]), int(ip[r1])
        if b == a:
            continue
        wts = torch.rand(b - a, generator=g, device=device, dtype=torch.float32).to(torch.float64) + 1e-3
        cs = torch.cumsum(wts, 0)
        lens = ip[r0 + 1:r1 + 1] - ip[r0:r1]
        starts = ip[r0:r1] - a
        base = torch.where(starts > 0, cs[(starts - 1).clamp_(min=0)], torch.zeros((), dtype=torch.float64, device=device))
        out[a:b] = (cs - torch.repeat_interleave(base, lens)).to(torch.float32)
        del wts, cs, lens, starts, base
    return out


def gen_powerlaw_on_gpu(num_node, num_edge, seed, device):
    """--graph powerlaw (round 1's generator; no community structure).  Same construction as synth.powerlaw_csr (power-law row lengths, hub-skewed neighbour ids), done with
    torch on the GPU in chunks so that a 1.6 G-edge CSR is built in seconds without host memory."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    u = torch.rand(num_node, generator=g, device=device, dtype=torch.float64)
    raw = (1.0 - u).pow(-1.0 / 1.8) - 1.0 + 0.05
    raw[torch.rand(num_node, generator=g, device=device) < 0.02] = 0.0
    deg = torch.floor(raw * (num_edge / raw.sum())).to(torch.int64)
    # heavy tail: cap a single row at 2^24 entries and spread the remainder uniformly
    deg.clamp_(max=1 << 24)
    short = int(num_edge - int(deg.sum()))
    if short > 0:
        bump = torch.randint(0, num_node, (short,), generator=g, device=device)
        deg.index_add_(0, bump, torch.ones_like(bump))
    elif short < 0:
        big = torch.nonzero(deg > 0).flatten()
        take = big[torch.randperm(big.numel(), generator=g, device=device)[:(-short)]]
        deg[take] -= 1
    indptr64 = torch.zeros(num_node + 1, dtype=torch.int64, device=device)
    torch.cumsum(deg, 0, out=indptr64[1:])
    total = int(indptr64[-1])
    assert total < 2**32
    indptr = (indptr64 & 0xFFFFFFFF).to(torch.int32) if total >= 2**31 else indptr64.to(torch.int32)
    mul = 2654435761 % num_node
    while np.gcd(mul, num_node) != 1:
        mul += 1
    indices = torch.empty(total, dtype=torch.int32, device=device)
    chunk = 1 << 26
    for a in range(0, total, chunk):
        b = min(total, a + chunk)
        x = torch.rand(b - a, generator=g, device=device, dtype=torch.float64)
        ids = torch.clamp((num_node * x * x).to(torch.int64), max=num_node - 1)
        ids = (ids * mul) % num_node
        indices[a:b] = ids.to(torch.int32)
        del x, ids
    del deg, indptr64, raw, u
    return indptr, indices, total


def gen_features_on_gpu(num_node, dim, device):
    feat = torch.empty((num_node, dim), dtype=torch.float32, device=device)
    rows = max(1, (1 << 28) // dim)
    col = torch.arange(dim, device=device, dtype=torch.int32)[None, :] * 7
    for a in range(0, num_node, rows):
        b = min(num_node, a + rows)
        r = torch.arange(a, b, device=device, dtype=torch.int32)[:, None] * 131
        feat[a:b] = ((r + col) & 0xFFFF).to(torch.float32) * (1.0 / 65536.0)
    return feat


def reduce_over_ranks(elapsed, edges, rows, device=None):
    """Contract: time = MAX over ranks, work = SUM over ranks (no other collective touches the data path)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return elapsed, edges, rows
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    e = torch.tensor([edges, rows], dtype=torch.float64, device=device)
    dist.all_reduce(e, op=dist.ReduceOp.SUM)
    return float(t[0]), float(e[0]), float(e[1])


def local_step_range(steps_per_epoch, rank, world):
    """First step and count of this rank's contiguous step range (DistShuffler, dist/dist_shuffler.cc:59-79)."""
    first = (steps_per_epoch // world) * rank
    count = steps_per_epoch - first if rank == world - 1 else steps_per_epoch // world
    return first, count


def pmc_traffic():
    """HBM bytes / algorithmic bytes from the newest committed rocprofv3 PMC passes (profiles/r*_pmc_traffic.json,
    written by tools/pmc_summary.py: FETCH_SIZE / WRITE_SIZE corrected as MI355X_MICROARCH.md prescribes, separate
    --pmc passes of this command).  Returns (gather ratio or None, per-stage dict with every kernel family's
    traffic_over_algorithmic or None, file name)."""
    import glob
    names = sorted((os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))),
                   reverse=True)  # newest round / tag first
    for name in names:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            return d.get("traffic_over_algorithmic"), d.get("per_stage") or d.get("per_kernel"), name
        except Exception:
            continue
    return None, None, None


def pmc_requests(workload):
    """fabric requests per batch of the sampler-side stage from the newest committed counter pass
    (profiles/r*_pmc_requests.json, tools/pmc_requests.sh: TCC_EA0_RDREQ / WRREQ per kernel) for this workload, or None"""
    import glob
    for name in sorted((os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_requests.json"))),
                       reverse=True):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)["workloads"].get(workload)
            if d:
                return d, name
        except Exception:
            continue
    return None, None


def algorithmic_bytes(metas, feat_dim, batch_size):
    """per batch, 4-byte ids.  Returns dict of per-stage algorithmic bytes (sums)."""
    sample = dedup = split = gather = 0
    for m in metas:
        L = m.num_layers
        for l in range(L):
            S, E = m.num_dst[l], m.num_edge[l]
            n_new = m.num_src[l] - m.num_dst[l]
            sample += S * 12 + E * 12
            dedup += E * 16 + n_new * 4
        U = m.num_input
        split += U * 16
        gather += U * (4 + 8 * feat_dim) + m.num_output * 20
    return dict(sample=sample, dedup_remap=dedup, cache_split=split, gather=gather)


def gen_graph_on_gpu(num_node, num_edge, seed, device, graph=None):
    """(indptr, indices, num_edge) of the workload graph on `device`: the R-MAT graph of SURVEY.md 8(d) unless
    graph == "powerlaw" (or FGNN_BENCH_GRAPH=powerlaw)"""
    graph = graph or os.environ.get("FGNN_BENCH_GRAPH", "rmat")
    if graph == "rmat":
        from fgnn_hip import rmat
        return rmat.rmat_csr(num_node, num_edge, seed, device)
    return gen_powerlaw_on_gpu(num_node, num_edge, seed, device)


def gen_graph(args, w, dev):
    """(indptr, indices, num_edge, description) of the workload graph on `dev`"""
    indptr, indices, ne = gen_graph_on_gpu(w["num_node"], w["num_edge"], 42, dev, args.graph)
    return indptr, indices, ne, GRAPH_DESC[args.graph]


def gen_train_set(args, w, dev):
    """train ids (uniform random ids, seed 1), shuffled once like one DistShuffler epoch"""
    from fgnn_hip import rmat
    train = rmat.train_set(w["num_node"], w["num_train"], 1, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    return train[torch.randperm(train.numel(), generator=g, device=dev)]


# ---- NUMA placement of host tables the GPU reads over the host link ---------------------------------------------------
def gpu_numa_node(dev_id):
    """NUMA node the GPU's PCIe root hangs off (sysfs), or None"""
    try:
        p = torch.cuda.get_device_properties(dev_id)
        bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as f:
            n = int(f.read().strip())
        return n if n >= 0 else None
    except Exception:
        return None


def numa_nodes_with_memory():
    try:
        txt = open("/sys/devices/system/node/has_memory").read().strip()
        out = []
        for part in txt.split(","):
            a, _, b = part.partition("-")
            out += list(range(int(a), int(b or a) + 1))
        return out
    except Exception:
        return []


def pages_by_numa_node(addr, nbytes, samples=1024):
    """{node: pages} over `samples` evenly spaced pages of [addr, addr + nbytes) (move_pages in query mode)"""
    import ctypes as C
    try:
        numa = C.CDLL("libnuma.so.1")
        n = max(1, min(samples, nbytes // 4096))
        base = addr & ~4095
        pages = (C.c_void_p * n)(*[base + (i * (nbytes // n) & ~4095) for i in range(n)])
        status = (C.c_int * n)()
        numa.numa_move_pages.argtypes = [C.c_int, C.c_ulong, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                         C.c_int]
        if numa.numa_move_pages(0, n, pages, None, status, 0) != 0:
            return None
        hist = {}
        for v in status:
            hist[int(v)] = hist.get(int(v), 0) + 1
        return {str(k): v for k, v in sorted(hist.items())}
    except Exception:
        return None


class HostTable:
    """A pinned, GPU-readable host array placed on a chosen NUMA node: numa_alloc_onnode + first touch + hipHostRegister.
    torch's pin_memory (hipHostMalloc) leaves the placement to the runtime; on a two-socket host a table on the far
    socket costs the GPU's row reads the inter-socket hop (BENCH_r02: 35.9 GB/s on one box, 49-55 on others)."""

    def __init__(self, rows, dim, node):
        import ctypes as C
        self.C, self.nbytes, self.node = C, rows * dim * 4, node
        self.numa = C.CDLL("libnuma.so.1")
        self.numa.numa_alloc_onnode.restype = C.c_void_p
        self.numa.numa_alloc_onnode.argtypes = [C.c_size_t, C.c_int]
        self.numa.numa_free.argtypes = [C.c_void_p, C.c_size_t]
        self.ptr = self.numa.numa_alloc_onnode(self.nbytes, node)
        if not self.ptr:
            raise MemoryError("numa_alloc_onnode(%d bytes, node %d)" % (self.nbytes, node))
        self.array = np.frombuffer((C.c_char * self.nbytes).from_address(self.ptr), dtype=np.float32).reshape(rows, dim)
        self.array[:] = 0  # first touch under the node binding
        self.hip = C.CDLL("libamdhip64.so")
        self.registered = False
        rc = self.hip.hipHostRegister(C.c_void_p(self.ptr), C.c_size_t(self.nbytes), C.c_uint(3))  # portable | mapped
        if rc != 0:
            self.free()
            raise RuntimeError("hipHostRegister failed with %d" % rc)
        self.registered = True
        d = C.c_void_p()
        rc = self.hip.hipHostGetDevicePointer(C.byref(d), C.c_void_p(self.ptr), C.c_uint(0))
        if rc != 0 or not d.value:
            self.free()
            raise RuntimeError("hipHostGetDevicePointer failed with %d" % rc)
        self.device_ptr = d.value
        self.tensor = torch.from_numpy(self.array)

    def free(self):
        if self.registered:
            self.hip.hipHostUnregister(self.C.c_void_p(self.ptr))
            self.registered = False
        if self.ptr:
            self.tensor = self.array = None
            self.numa.numa_free(self.C.c_void_p(self.ptr), self.nbytes)
            self.ptr = None


def sampler_config_keys(w, sample_type):
    """the sampler's part of a samgraph run config for workload `w` (operation.cc:58-164): fan-outs for the k-hop
    samplers (train_graphsage.py:77, train_gcn.py:72), the walk parameters for random_walk (train_pinsage.py:130-134)"""
    if sample_type == "random_walk":
        # (workloads without walk parameters of their own: the reference's defaults, train_pinsage.py:130-134)
        return dict(random_walk_length=w.get("walk_len", 3), random_walk_restart_prob=w.get("restart_prob", 0.5),
                    num_random_walk=w.get("num_walks", 4), num_neighbor=w["fanout"][0], num_layer=len(w["fanout"]))
    return dict(num_fanout=len(w["fanout"]), fanout=w["fanout"])


def write_dataset(args, w, dev, out_dir):
    """the engine's on-disk layout without feat.bin (SAMGRAPH_EMPTY_FEAT, like papers100M_empty)"""
    os.makedirs(out_dir, exist_ok=True)
    indptr, indices, ne, desc = gen_graph(args, w, dev)
    indptr.cpu().numpy().view(np.uint32).tofile(os.path.join(out_dir, "indptr.bin"))
    chunk = 1 << 28
    with open(os.path.join(out_dir, "indices.bin"), "wb") as f:
        for a in range(0, ne, chunk):
            f.write(indices[a:a + chunk].cpu().numpy().view(np.uint32).tobytes())
    if args.sample_type == "weighted_khop_prefix":
        # the per-row inclusive prefix sums of the edge weights (create_prob_prefix_table.cc; the N = 1 path's generator)
        prefix = gen_prefix_on_gpu(indptr, ne, 11, dev)
        with open(os.path.join(out_dir, "prob_prefix_table.bin"), "wb") as f:
            for a in range(0, ne, chunk):
                f.write(prefix[a:a + chunk].cpu().numpy().tobytes())
        del prefix
    elif args.sample_type in ("weighted_khop", "weighted_khop_hash_dedup"):
        prob_t, alias_t = gen_alias_on_gpu(indices, ne, 12, dev)
        with open(os.path.join(out_dir, "prob_table.bin"), "wb") as f:
            for a in range(0, ne, chunk):
                f.write(prob_t[a:a + chunk].cpu().numpy().tobytes())
        with open(os.path.join(out_dir, "alias_table.bin"), "wb") as f:
            for a in range(0, ne, chunk):
                f.write(alias_t[a:a + chunk].cpu().numpy().view(np.uint32).tobytes())
        del prob_t, alias_t
    del indptr, indices
    from fgnn_hip import rmat
    train = rmat.train_set(w["num_node"], w["num_train"], 1, dev)
    train.cpu().numpy().view(np.uint32).tofile(os.path.join(out_dir, "train_set.bin"))
    np.zeros(0, dtype=np.uint32).tofile(os.path.join(out_dir, "valid_set.bin"))
    np.zeros(0, dtype=np.uint32).tofile(os.path.join(out_dir, "test_set.bin"))
    with open(os.path.join(out_dir, "meta.txt"), "w") as f:
        f.write(f"NUM_NODE {w['num_node']}\nNUM_EDGE {ne}\nFEAT_DIM {w['feat_dim']}\nNUM_CLASS {w['num_class']}\n"
                f"NUM_TRAIN_SET {w['num_train']}\nNUM_VALID_SET 0\nNUM_TEST_SET 0\n")
    torch.cuda.empty_cache()
    return ne, desc
