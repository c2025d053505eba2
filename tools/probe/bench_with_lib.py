#!/usr/bin/env python3
"""bench.py against another build of libfgnn_hip.so (A/B of two library builds on one box):
bench_with_lib.py <path/to/libfgnn_hip.so> [bench.py arguments]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fgnn-artifacts_amd"))
sys.path.insert(0, ROOT)
from fgnn_hip import lib  # noqa: E402

lib.use_library(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
