#!/usr/bin/env python3
"""Where does a GraphSAGE training step (torch ops on the engine's COO blocks) spend GPU time?  Synthetic blocks of
the papers100M-shaped batch size.  Profiling aid."""
import os
import sys

import torch
import torch.nn as nn
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
from models import SAGE  # noqa: E402


class Block:
    def __init__(self, row, col, nsrc, ndst):
        self.row, self.col, self.nsrc, self.ndst = row, col, nsrc, ndst
        self.edata = {}

    def number_of_dst_nodes(self):
        return self.ndst


dev = "cuda:0"
g = torch.Generator(device=dev)
g.manual_seed(0)
# the papers100M-shaped R-MAT batch of bench.py: layer 0 (outer) 302 K src -> 22.5 K dst, 371 K edges; layer 1: 22.5 K src
# -> 8000 dst, 15 K edges
S0, D0, E0, E1 = 302000, 22500, 371000, 15000
b0 = Block(torch.randint(0, S0, (E0,), device=dev, generator=g, dtype=torch.int32), torch.sort(torch.randint(0, D0, (E0,), device=dev, generator=g, dtype=torch.int32))[0], S0, D0)
b1 = Block(torch.randint(0, D0, (E1,), device=dev, generator=g, dtype=torch.int32), torch.sort(torch.randint(0, 8000, (E1,), device=dev, generator=g, dtype=torch.int32))[0], D0, 8000)
x = torch.randn(S0, 128, device=dev)
y = torch.randint(0, 172, (8000,), device=dev)
model = SAGE(128, 256, 172, 2, 0.5, fused=os.environ.get('SAGE_FUSED', '1') == '1').to(dev)
# TRAIN_OPS=1 (default): the one-launch pieces of csrc/train_ops.hip as bench.py's train legs use them (fgnn_hip.nn.Adam,
# fused ReLU + dropout keyed by its step count, fgnn_softmax_xent); 0: torch's fused Adam, F.relu + Dropout, CrossEntropyLoss
TRAIN_OPS = os.environ.get('TRAIN_OPS', '1') == '1'
lossf = nn.CrossEntropyLoss()
if TRAIN_OPS:
    from fgnn_hip.nn import Adam, softmax_xent
    opt = Adam(model.parameters(), lr=0.003)
    model.dropout_step = opt.step_count
else:
    opt = torch.optim.Adam(model.parameters(), lr=0.003, fused=os.environ.get('ADAM_FUSED', '1') == '1', capturable=True)


def step():
    out = model([b0, b1], x)
    if TRAIN_OPS:
        loss, g = softmax_xent(out, y)
        opt.zero_grad()
        out.backward(g)
    else:
        loss = lossf(out, y)
        opt.zero_grad()
        loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    step()
e1.record()
torch.cuda.synchronize()
print("eager step %.3f ms (SAGE_FUSED=%s TRAIN_OPS=%s)" % (e0.elapsed_time(e1) / 20, os.environ.get('SAGE_FUSED', '1'), TRAIN_OPS))
# the same step replayed as a captured graph (what examples/graphed_step.py does per size bucket)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    for _ in range(3):
        graph.replay()
    side.synchronize()
    e0.record()
    for _ in range(20):
        graph.replay()
    e1.record()
    side.synchronize()
    print("graph replay %.3f ms per step" % (e0.elapsed_time(e1) / 20))
    with profile(activities=[ProfilerActivity.CUDA]) as gprof:
        for _ in range(5):
            graph.replay()
        side.synchronize()
    rows = [(e.key, e.count, e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total)
            for e in gprof.key_averages()]
    rows.sort(key=lambda r: -r[2])
    nk = sum(r[1] for r in rows) / 5
    print("graph: %.0f kernel / memset nodes per step, %.0f us of device time per step" % (nk, sum(r[2] for r in rows) / 5))
    for k, c, t in rows[:40]:
        print("  %-90s x%-3d %8.1f us per step" % (k[:90], c // 5, t / 5))
torch.cuda.current_stream().wait_stream(side)
if os.environ.get("GRAPH_ONLY"):
    sys.exit(0)
# weight-gradient GEMM gy^T x at the row counts a batch's layers have: library GEMM against the 32-slice batched GEMM
for m in (8000, 22500, 88000, 302000):
    for (k, n) in ((128, 256), (256, 172)):
        xx, gy = torch.randn(m, k, device=dev), torch.randn(m, n, device=dev)
        mp = m // 32 * 32

        def t(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 10 * 1e3
        a = t(lambda: gy.t().mm(xx))
        b = t(lambda: torch.bmm(gy[:mp].view(32, mp // 32, -1).transpose(1, 2), xx[:mp].view(32, mp // 32, -1)).sum(0))
        print("gw %6d x %3d x %3d: mm %.1f us, 32-slice bmm + sum %.1f us" % (m, k, n, a, b))
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=int(os.environ.get('ROWS', '30')), max_name_column_width=70))
