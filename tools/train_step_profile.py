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
# layer 0 (outer): 528 K src -> 88 K dst, 490 K edges ; layer 1: 88 K src -> 8000 dst, 80 K edges
b0 = Block(torch.randint(0, 528000, (490000,), device=dev, generator=g, dtype=torch.int32), torch.sort(torch.randint(0, 88000, (490000,), device=dev, generator=g, dtype=torch.int32))[0], 528000, 88000)
b1 = Block(torch.randint(0, 88000, (80000,), device=dev, generator=g, dtype=torch.int32), torch.sort(torch.randint(0, 8000, (80000,), device=dev, generator=g, dtype=torch.int32))[0], 88000, 8000)
x = torch.randn(528000, 128, device=dev)
y = torch.randint(0, 172, (8000,), device=dev)
model = SAGE(128, 256, 172, 2, 0.5).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=0.003, fused=os.environ.get('ADAM_FUSED', '1') == '1')
lossf = nn.CrossEntropyLoss()


def step():
    loss = lossf(model([b0, b1], x), y)
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
print("step %.3f ms" % (e0.elapsed_time(e1) / 20))
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=int(os.environ.get('ROWS', '30')), max_name_column_width=70))
