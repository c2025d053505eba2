"""samgraph.common must reproduce the reference's enum block (samgraph/common/__init__.py:47-265)."""
import json
import os


def test_constants_match_reference(golden_dir):
    import samgraph.common as sc
    with open(os.path.join(golden_dir, "py_constants.json")) as f:
        ref = json.load(f)
    assert len(ref["constants"]) == 130
    for name, val in ref["constants"].items():
        assert getattr(sc, name) == val, name
    assert sc.sample_types == ref["sample_types"]
    assert sc.cache_policies == ref["cache_policies"]
    assert sc.builtin_archs == ref["builtin_archs"]
    assert sc.cpu(1) == "cpu:1" and sc.gpu(3) == "cuda:3"
