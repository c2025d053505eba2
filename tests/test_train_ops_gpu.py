"""The fused pieces of the GraphSAGE training step (csrc/train_ops.hip, fgnn_hip/nn.py) against the torch ops they
replace -- the consumer side of the path (reference: example/samgraph/multi_gpu/
train_graphsage.py:24-51,300-330: SAGEConv, ReLU, Dropout, CrossEntropyLoss, Adam).  fp32; elementwise pieces are
bit-equal to the torch formulation, reductions within 1e-6."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nn():
    from fgnn_hip import lib, nn
    lib.load()
    return nn


@pytest.mark.parametrize("num_dst,num_src,din", [(1000, 5000, 128), (8001, 9000, 256), (1, 3, 4), (777, 777, 100)])
def test_sage_finish_z_and_grad_prep(nn, num_dst, num_src, din):
    g = torch.Generator(device="cuda").manual_seed(1)
    h = torch.randn((num_src, din), generator=g, device="cuda")
    z = torch.randn((num_dst, 2 * din), generator=g, device="cuda")
    deg = torch.randint(0, 5, (num_dst,), generator=g, device="cuda").float()
    want_inv = deg.clamp(min=1).reciprocal()
    want = z.clone()
    want[:, din:] *= want_inv.unsqueeze(1)
    want[:, :din] = h[:num_dst]
    inv = nn.sage_finish_z(z, h, deg, num_dst, din)
    assert torch.equal(inv, want_inv) and torch.equal(z, want)
    gz = torch.randn((num_dst, 2 * din), generator=g, device="cuda")
    gh, gagg = nn.sage_grad_prep(gz, inv, num_src, din)
    assert torch.equal(gagg, gz[:, din:] * inv.unsqueeze(1))
    want_gh = torch.zeros((num_src, din), device="cuda")
    want_gh[:num_dst] = gz[:, :din]
    assert torch.equal(gh, want_gh)


def test_relu_dropout_masks_follow_the_step_count(nn):
    x = torch.randn((4096, 256), device="cuda", requires_grad=True)
    step = torch.zeros(2, dtype=torch.int64, device="cuda")
    p = 0.5
    y0 = nn.relu_dropout(x, p, True, seed=7, d_step=step, tag=0)
    y0b = nn.relu_dropout(x, p, True, seed=7, d_step=step, tag=0)
    assert torch.equal(y0, y0b)  # same (seed, step, tag): same mask
    pos = x.detach() > 0
    kept = y0.detach() != 0
    assert not (kept & ~pos).any()
    frac = float(kept[pos].float().mean())
    assert abs(frac - (1 - p)) < 0.01, frac
    assert torch.equal(y0.detach()[kept], (x.detach() * (1.0 / (1.0 - p)))[kept])
    gy = torch.randn_like(y0)
    (gx,) = torch.autograd.grad(y0, x, gy)
    assert torch.equal(gx, torch.where(kept, gy * (1.0 / (1.0 - p)), torch.zeros_like(gy)))
    step[0] = 1  # the optimizer has taken a step: a fresh mask
    y1 = nn.relu_dropout(x, p, True, seed=7, d_step=step, tag=0)
    assert float(((y1 != 0) != kept).float().mean()) > 0.2
    y_tag = nn.relu_dropout(x, p, True, seed=7, d_step=step, tag=1)
    assert not torch.equal(y1, y_tag)  # another layer: another mask
    assert torch.equal(nn.relu_dropout(x, p, False), torch.relu(x))  # eval mode


@pytest.mark.parametrize("n,c,pad", [(8000, 172, 0), (8000, 172, 1), (37, 47, 0), (1, 3, 5), (1000, 1000, 0)])
def test_softmax_xent_matches_cross_entropy(nn, n, c, pad):
    g = torch.Generator(device="cuda").manual_seed(n + c)
    logits = (torch.randn((n + pad, c), generator=g, device="cuda") * 3).requires_grad_()
    y = torch.randint(0, c, (n,), generator=g, device="cuda")
    want = torch.nn.functional.cross_entropy(logits[:n], y)
    (want_g,) = torch.autograd.grad(want, logits)
    loss, grad = nn.softmax_xent(logits.detach()[:n], y, pad_rows=pad)
    loss2, _ = nn.softmax_xent(logits.detach()[:n], y, pad_rows=pad)
    assert torch.equal(loss, loss2)  # fixed summation order
    assert grad.shape == (n + pad, c)
    torch.testing.assert_close(loss, want, rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(grad, want_g, rtol=1e-5, atol=1e-8)
    if pad:
        assert not grad[n:].any()


@pytest.mark.parametrize("n,c", [(8000, 172), (37, 47), (300, 600)])
def test_softmax_xent_ignores_out_of_range_labels_like_ignore_index(nn, n, c):
    """rows labelled -100 (torch's default ignore_index): no loss, a zero gradient row, the mean over the other rows
    (advisor, round 5: the kernel used to write softmax / n into such rows and divide by n)"""
    g = torch.Generator(device="cuda").manual_seed(n)
    logits = (torch.randn((n, c), generator=g, device="cuda") * 2).requires_grad_()
    y = torch.randint(0, c, (n,), generator=g, device="cuda")
    y[torch.randperm(n, generator=g, device="cuda")[: n // 5]] = -100
    want = torch.nn.functional.cross_entropy(logits, y)
    (want_g,) = torch.autograd.grad(want, logits)
    loss, grad = nn.softmax_xent(logits.detach(), y)
    torch.testing.assert_close(loss, want, rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(grad, want_g, rtol=1e-5, atol=1e-8)
    assert not grad[y < 0].any()
    # nothing to learn from: zero loss and gradient (torch: nan)
    loss0, grad0 = nn.softmax_xent(logits.detach(), torch.full_like(y, -100))
    assert float(loss0) == 0.0 and not grad0.any()


@pytest.mark.parametrize("shapes,wd", [([(256, 256), (256,), (172, 512), (172,)], 0.0), ([(33,)] * 11, 0.01)])
def test_adam_matches_torch_adam(nn, shapes, wd):
    g = torch.Generator(device="cuda").manual_seed(3)
    ps = [torch.randn(s, generator=g, device="cuda").requires_grad_() for s in shapes]
    qs = [p.detach().clone().requires_grad_() for p in ps]
    ref = torch.optim.Adam(ps, lr=0.003, weight_decay=wd)
    opt = nn.Adam(qs, lr=0.003, weight_decay=wd)
    for step in range(6):
        for p, q in zip(ps, qs):
            grad = torch.randn(p.shape, generator=g, device="cuda")
            p.grad, q.grad = grad.clone(), grad.clone()
        ref.step()
        opt.step()
        assert int(opt.step_count[0]) == step + 1
        for p, q in zip(ps, qs):
            torch.testing.assert_close(q, p, rtol=2e-6, atol=1e-7)
    opt.zero_grad()
    assert all(q.grad is None for q in qs)


def test_adam_step_is_capturable(nn):
    """the step count lives on the device: a replayed graph keeps stepping (bias corrections included)"""
    p = torch.ones(1000, device="cuda").requires_grad_()
    q = p.detach().clone().requires_grad_()
    ref = torch.optim.Adam([p], lr=0.01)
    opt = nn.Adam([q], lr=0.01)
    q.grad = torch.full_like(q, 0.5)
    p.grad = torch.full_like(p, 0.5)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        opt.step()  # warm-up outside the capture
        ref.step()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            opt.step()
        for _ in range(4):
            gr.replay()
            ref.step()
    torch.cuda.synchronize()
    assert int(opt.step_count[0]) == 5
    torch.testing.assert_close(q, p, rtol=2e-6, atol=1e-7)
