#!/usr/bin/env python3
"""How many hardware queues do N HIP streams of one process land on?  Run under
`rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/probe/hw_queue_probe.py N` with GPU_MAX_HW_QUEUES
unset / 8 / 16 and count the distinct Queue_Id values of the trace (tools/probe/hw_queue_count.py DIR)."""
import sys
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
streams = [torch.cuda.Stream() for _ in range(n)]
x = [torch.ones(1 << 20, device="cuda") for _ in range(n)]
for r in range(3):
    for s, t in zip(streams, x):
        with torch.cuda.stream(s):
            t.mul_(1.0001)
torch.cuda.synchronize()
print("ok", n)
