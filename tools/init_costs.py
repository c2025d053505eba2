#!/usr/bin/env python3
"""Init-time costs of the engine on a dataset directory (the reference's Table 6 columns: DRAM -> GPU topology load,
pre-sampling, cache build): runs samgraph init for arch3 with a cache and prints the kLogInit* values.
usage: SAMGRAPH_EMPTY_FEAT=24 init_costs.py <dataset dir> [cache_percentage] [pre_sample|presample_static|degree]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
import samgraph.torch as sam  # noqa: E402

path = sys.argv[1]
cache = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
policy = sys.argv[3] if len(sys.argv) > 3 else "pre_sample"
sam.config(dict(dataset_path=path, _arch=sam.kArch3, _sample_type=sam.sample_types["khop2"], batch_size=8000,
                num_epoch=1, _cache_policy=sam.cache_policies[policy], presample_epoch=1, cache_percentage=cache,
                max_sampling_jobs=10, max_copying_jobs=2, omp_thread_num=8, sampler_ctx="cuda:0", trainer_ctx="cuda:0",
                num_fanout=2, fanout=[25, 10]))
t0 = time.time()
sam.init()
dt = time.time() - t0
names = ["kLogInitL1Common", "kLogInitL1Sampler", "kLogInitL1Trainer", "kLogInitL2LoadDataset", "kLogInitL2Presample",
         "kLogInitL2BuildCache"]
print("policy %s, cache %.2f: samgraph_init %.2f s; " % (policy, cache, dt) +
      ", ".join("%s %.3f" % (n[8:], sam.get_log_init_value(getattr(sam, n))) for n in names))
sam.shutdown()
