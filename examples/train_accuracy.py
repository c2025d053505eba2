"""Validation / test accuracy for the training examples (`--report-acc`), the counterpart of the reference's
example/samgraph/multi_gpu/train_accuracy.py (load_accuracy_data + Accuracy.valid_acc / test_acc, used at
multi_gpu/train_graphsage.py:214-220,344-347,395-397).

The reference evaluates with a DGL NodeDataLoader (MultiLayerNeighborSampler over the same fanout) on the dataset's
validation / test sets -- a path of its own next to the samgraph engine, so that evaluation never disturbs the
training pipeline's queue.  Same structure here, without DGL: the evaluation batches are sampled by this repo's
kernel-level sampler (fgnn_hip.lib.Sampler, khop0 -- uniform without replacement, the CSR copy stays untouched)
on the evaluation device, features and labels are gathered by the same kernels, and the model sees the same block
objects as in training (samgraph.torch.adapter.CooBlock; DGLBlocks when DGL is installed).
"""
import os

import numpy as np
import torch

from fgnn_hip import lib
from samgraph.torch import adapter


def _meta(path):
    out = {}
    with open(os.path.join(path, "meta.txt")) as f:
        for line in f:
            k, v = line.split()
            out[k] = int(v)
    return out


def load_accuracy_data(dataset_path):
    """(graph = (indptr, indices) u32 arrays, valid_set, test_set, feat f32[N, D], label i64[N]) as host arrays mapped
    from the dataset directory (samgraph's on-disk layout, engine.cc:73-264)."""
    m = _meta(dataset_path)
    n, e, d = m["NUM_NODE"], m["NUM_EDGE"], m["FEAT_DIM"]

    def arr(name, dtype, shape):
        return np.memmap(os.path.join(dataset_path, name), dtype=dtype, mode="r", shape=shape)
    graph = (arr("indptr.bin", np.uint32, (n + 1,)), arr("indices.bin", np.uint32, (e,)))
    valid = np.array(arr("valid_set.bin", np.uint32, (m["NUM_VALID_SET"],))) if m.get("NUM_VALID_SET") else np.zeros(0, np.uint32)
    test = np.array(arr("test_set.bin", np.uint32, (m["NUM_TEST_SET"],))) if m.get("NUM_TEST_SET") else np.zeros(0, np.uint32)
    feat = arr("feat.bin", np.float32, (n, d))
    label = arr("label.bin", np.uint64, (n,))
    return graph, valid, test, feat, label


class Accuracy:
    """Accuracy(graph, valid_set, test_set, feat, label, fanout, batch_size, sample_device): the reference's
    constructor.  Graph, features and labels are uploaded to `sample_device` once (evaluation datasets of the sizes
    the examples train on fit; for larger ones pass feat / label slices that cover the evaluation sets' neighbourhoods)."""

    def __init__(self, graph, valid_set, test_set, feat, label, fanout, batch_size, sample_device, seed=0xACC):
        self.dev = torch.device(sample_device)
        self.fanout, self.batch_size = list(fanout), int(batch_size)

        def up(a):
            a = np.ascontiguousarray(a)
            if a.dtype == np.uint32:  # ids: u32 storage viewed as i32, like the engine's tensors
                a = a.view(np.int32)
            if a.dtype == np.uint64:
                a = a.view(np.int64)
            return torch.from_numpy(a).to(self.dev)
        with torch.cuda.device(self.dev):
            self.indptr, self.indices = up(graph[0]), up(graph[1])
            self.feat, self.label = up(feat), up(label)
            self.valid_set, self.test_set = up(valid_set), up(test_set)
            # its own sampler, own CSR copy, own RNG stream: nothing of the training engine's state is touched
            self.sampler = lib.Sampler(self.indptr, self.indices, self.fanout, self.batch_size, sample_type=lib.KHOP0,
                                       seed=seed)
            self.batch = self.sampler.new_batch(self.feat.shape[1], lib.F32, lib.I64)
        self.seq = 0
        self.calls = 0

    def _evaluate(self, model, ids, train_device):
        total = correct = 0
        was_training = model.training
        model.eval()
        L = len(self.fanout)
        with torch.no_grad(), torch.cuda.device(self.dev):
            for i in range(0, ids.numel(), self.batch_size):
                seeds = ids[i:i + self.batch_size]
                # batch key: a fresh draw per evaluation batch, reproducible for a given call order
                self.sampler.run_batch(self.seq, seeds, (self.calls << 24) | (i // self.batch_size), self.batch, None,
                                       self.feat, self.label)
                self.seq += 1
                self.batch.wait()
                blocks = []
                for l in range(L):
                    row, col, nsrc, ndst = self.batch.graph(l)
                    blocks.append(adapter._create_dgl_block((row.to(train_device), col.to(train_device)), nsrc, ndst))
                out = model(blocks, self.batch.feat().to(train_device))
                pred = out.argmax(1)
                correct += int((pred == self.batch.label().to(train_device)).sum())
                total += int(seeds.numel())
        self.calls += 1
        if was_training:
            model.train()
        return correct / max(total, 1)

    def valid_acc(self, model, train_device):
        return self._evaluate(model, self.valid_set, torch.device(train_device))

    def test_acc(self, model, train_device):
        return self._evaluate(model, self.test_set, torch.device(train_device))
