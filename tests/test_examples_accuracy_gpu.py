"""The end-to-end check the op-by-op parity tests cannot give: a model trained through the drop-in boundary LEARNS.
The reference reports validation / test accuracy with --report-acc (example/samgraph/multi_gpu/train_graphsage.py:
65,214-220,344-347,395-397; exp/fig16a); the examples here do the same (examples/train_accuracy.py) on a synthetic graph
whose labels follow from features and neighbourhoods (fgnn_hip.synth.learnable_labels: chance = 1 / 8)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--fanout", "10", "5", "--batch-size", "1000", "--num-epoch", "2", "--num-hidden", "64", "--lr", "0.01",
          "--report-acc", "25"]
CHANCE = 1.0 / 8


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
    from fgnn_hip import synth
    root = tmp_path_factory.mktemp("acc")
    sh = synth.LEARNABLE_SHAPE
    return synth.write_dataset(str(root), "learn", sh["num_node"], sh["num_edge"], sh["feat_dim"], sh["num_class"],
                               sh["num_train"], sh["num_valid"], sh["num_test"], learnable=True)


def _run(script, dataset, *extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script), "--dataset-path", dataset] + COMMON +
                       list(extra), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    valid = [float(x) / 100 for x in re.findall(r"Valid Acc: ([0-9.]+)%", p.stdout)]
    test = float(re.search(r"test_result:test_acc=([0-9.]+)", p.stdout).group(1))
    return valid, test


def test_arch1_training_learns_fused_and_op_by_op_layers_alike(dataset):
    """one GPU samples, extracts and trains (arch1): accuracy far above chance after two epochs (+ the warm-up one),
    rising over the run; the fused SAGE layers and the op-by-op ones reach the same accuracy within noise"""
    valid_f, test_f = _run("train_graphsage.py", dataset, "--arch", "arch1")
    valid_o, test_o = _run("train_graphsage.py", dataset, "--arch", "arch1", "--op-by-op")
    for valid, test in ((valid_f, test_f), (valid_o, test_o)):
        assert len(valid) >= 3 and valid[0] < valid[-1] and valid[-1] > 4 * CHANCE and test > 4 * CHANCE, (valid, test)
        assert abs(valid[-1] - test) < 0.1  # validation and test sets: the same distribution
    assert abs(test_f - test_o) < 0.08, (test_f, test_o)


def test_arch5_training_learns(dataset):
    """the factored pipeline (sampler process -> queue -> trainer process with a pre-sample cache, both on this GPU)"""
    valid, test = _run(os.path.join("multi_gpu", "train_fgnn.py"), dataset, "--single-gpu", "--cache-percentage", "0.2")
    assert len(valid) >= 3 and valid[-1] > 4 * CHANCE and test > 4 * CHANCE, (valid, test)
