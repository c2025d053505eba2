#!/usr/bin/env python3
"""Weight-gradient GEMM of a layer, gw = gy^T z with tens of thousands of rows reduced into a 256 x 256 (172 x 512)
result: the library GEMM against manual split-K (batched GEMM over row slices + sum) for several slice counts, and
against the transposed formulation.  fp32.  Profiling aid for examples/models.py."""
import torch

dev = "cuda:0"
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def t(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


for m in (8000, 22500, 88000):
    for (k, n) in ((256, 256), (512, 172)):
        z, gy = torch.randn(m, k, device=dev), torch.randn(m, n, device=dev)
        ref = gy.t().mm(z)
        line = ["gw m=%6d z[.,%3d] gy[.,%3d]: mm %.1f us" % (m, k, n, t(lambda: gy.t().mm(z))),
                "(z^T gy)^T %.1f" % t(lambda: z.t().mm(gy).t())]
        for s in (8, 16, 32, 64, 128):
            mp = m // s * s

            def f():
                out = torch.bmm(gy[:mp].view(s, mp // s, -1).transpose(1, 2), z[:mp].view(s, mp // s, -1)).sum(0)
                return out + gy[mp:].t().mm(z[mp:]) if mp < m else out
            err = (f() - ref).abs().max().item() / ref.abs().max().item()
            line.append("s=%d %.1f" % (s, t(f)))
            assert err < 1e-4, err
        print("  ".join(line))
