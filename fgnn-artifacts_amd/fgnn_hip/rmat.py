"""R-MAT graphs in the engine's CSR layout ((a, b, c, d) = (0.57, 0.19, 0.19, 0.05), seed 42, directed
edges de-duplicated, CSR by destination -- the orientation datagen/papers100M.py:66-70 builds: row = destination,
entries = sources in ascending order).

Written with torch tensor ops only, so the same code builds a 1.6 G-edge graph on the GPU in seconds (bench.py,
tools/make_big_dataset.py) and a 2^12-node graph on the CPU (tests).  Nothing here is on the measured path.

Construction
  * `scale` = ceil(log2(num_node)) levels of the quadrant recursion; one uniform draw per edge and level picks
    (src bit, dst bit) = a:(0,0) b:(0,1) c:(1,0) d:(1,1);
  * num_node is not a power of two for the reference's datasets: pairs with an endpoint >= num_node are rejected
    (keeps the recursive structure; folding ids with a modulo would superimpose two communities);
  * R-MAT puts the hubs at the low ids; real datasets do not keep their hot rows contiguous in memory, so ids are
    relabelled by a fixed bijection id -> id * m mod num_node (gcd(m, num_node) = 1).  A relabelling changes no
    structural property (degrees, neighbourhood overlap, duplicate rates of a frontier), only addresses;
  * duplicates are removed by sorting 64-bit keys (dst << 32 | src), one dst range at a time (bounded memory, works
    beyond 2^31 edges); candidates are drawn with a surplus and topped up until there are at least num_edge distinct
    pairs, then the surplus is dropped at random positions: the result has EXACTLY num_edge edges.
"""
import math

import torch

RMAT_ABCD = (0.57, 0.19, 0.19, 0.05)
# candidates drawn per requested edge in the first round (distinct/valid fraction depends on the shape; a short round
# is topped up, a long one trimmed -- the hint only saves a round)
_SURPLUS_HINT = 1.35


def _scramble_multiplier(num_node):
    m = 2654435761 % num_node
    if m < 2:
        m = 1
    while math.gcd(m, num_node) != 1:
        m += 1
    return m


def rmat_pairs(scale, count, gen, device, abcd=RMAT_ABCD):
    """`count` (src, dst) pairs of a 2^scale-node R-MAT graph, int64 tensors."""
    a, b, c, _ = abcd
    src = torch.zeros(count, dtype=torch.int64, device=device)
    dst = torch.zeros(count, dtype=torch.int64, device=device)
    for _ in range(scale):
        r = torch.rand(count, generator=gen, device=device, dtype=torch.float32)
        src_bit = r >= (a + b)                                    # quadrants c, d
        dst_bit = ((r >= a) & (r < (a + b))) | (r >= (a + b + c))  # quadrants b, d
        src.mul_(2).add_(src_bit)
        dst.mul_(2).add_(dst_bit)
        del r, src_bit, dst_bit
    return src, dst


def _candidate_keys(num_node, want, scale, gen, device, abcd, scramble_mul, chunk):
    """>= `want` valid candidate keys (dst << 32 | src), in chunks."""
    out, have = [], 0
    while have < want:
        n = min(chunk, max(1024, int((want - have) * 1.05)))
        src, dst = rmat_pairs(scale, n, gen, device, abcd)
        ok = (src < num_node) & (dst < num_node)
        src, dst = src[ok], dst[ok]
        if scramble_mul != 1:
            src = (src * scramble_mul) % num_node
            dst = (dst * scramble_mul) % num_node
        keys = (dst << 32) | src
        out.append(keys)
        have += keys.numel()
        del src, dst, ok
    return out


def _unique_by_dst_range(chunks, num_node, parts):
    """sorted distinct keys of all chunks, as a list of tensors in ascending key order (one per dst range)."""
    if parts <= 1:
        return [torch.unique(torch.cat(chunks))]
    res = []
    step = (num_node + parts - 1) // parts
    for p in range(parts):
        lo, hi = (p * step) << 32, (min(num_node, (p + 1) * step)) << 32
        sel = [c[(c >= lo) & (c < hi)] for c in chunks]
        res.append(torch.unique(torch.cat(sel)))
        del sel
    return res


def rmat_csr(num_node, num_edge, seed=42, device="cpu", abcd=RMAT_ABCD, scramble=True, chunk=1 << 27, parts=None):
    """Returns (indptr int32[N+1] (uint32 bit pattern), indices int32[E], E) on `device`: exactly num_edge distinct
    directed edges, rows = destinations, entries ascending."""
    assert 0 < num_node < (1 << 31) and 0 <= num_edge < (1 << 32)
    assert num_edge <= num_node * num_node
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    scale = max(1, math.ceil(math.log2(num_node)))
    mul = _scramble_multiplier(num_node) if scramble else 1
    if parts is None:  # keep each sort below ~2^30 keys (R-MAT puts up to ~45 % of the edges into the first of 8 ranges)
        parts = 1 if num_edge < (1 << 28) else 8 if num_edge < (1 << 31) else 16
    uniq = []
    total = 0
    want = int(num_edge * _SURPLUS_HINT) + 1024
    rounds = 0
    while total < num_edge:
        chunks = _candidate_keys(num_node, want, scale, gen, device, abcd, mul, chunk)
        if uniq:
            chunks += uniq
        uniq = _unique_by_dst_range(chunks, num_node, parts)
        del chunks
        total = sum(int(u.numel()) for u in uniq)
        rounds += 1
        assert rounds < 64, "R-MAT cannot reach the requested number of distinct edges"
        # distinct pairs saturate: ask for more than the shortfall
        want = int((num_edge - total) * 2.0) + 1024
    # drop the surplus at random positions (uniformly over the distinct edges)
    surplus = total - num_edge
    if surplus:
        sizes = [int(u.numel()) for u in uniq]
        keep = [torch.ones(s, dtype=torch.bool, device=device) for s in sizes]
        left = surplus
        while left:  # `left` draws drop at most `left` edges (repeated positions: go round again for the rest)
            pos = torch.randint(0, total, (left,), generator=gen, device=device)
            base = 0
            for k, s in zip(keep, sizes):
                k[pos[(pos >= base) & (pos < base + s)] - base] = False
                base += s
            left = surplus - (total - sum(int(k.sum()) for k in keep))
        uniq = [u[k] for u, k in zip(uniq, keep)]
        del keep
    indices = torch.empty(num_edge, dtype=torch.int32, device=device)
    deg = torch.zeros(num_node, dtype=torch.int64, device=device)
    at = 0
    for u in uniq:
        n = int(u.numel())
        if not n:
            continue
        src = u & 0xFFFFFFFF
        indices[at:at + n] = torch.where(src >= (1 << 31), src - (1 << 32), src).to(torch.int32)
        d = u >> 32
        deg += torch.bincount(d, minlength=num_node)
        at += n
        del src, d
    assert at == num_edge
    del uniq
    indptr64 = torch.zeros(num_node + 1, dtype=torch.int64, device=device)
    torch.cumsum(deg, 0, out=indptr64[1:])
    indptr = torch.where(indptr64 >= (1 << 31), indptr64 - (1 << 32), indptr64).to(torch.int32)
    return indptr, indices, num_edge


def train_set(num_node, num_train, seed=1, device="cpu"):
    """uniform random distinct node ids ('uniform random ids, seed 1')"""
    gen = torch.Generator(device=torch.device(device))
    gen.manual_seed(seed)
    return torch.randperm(num_node, generator=gen, device=device)[:num_train].to(torch.int32)
