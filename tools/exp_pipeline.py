#!/usr/bin/env python3
"""The bench's file pipeline leg alone (nine 12 MP JPEGs -> one PNG), for profiling.  usage: python tools/exp_pipeline.py [reps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import imagestitching_amd as ist  # noqa: E402

print(json.dumps(bench.file_pipeline_leg(ist, reps=int(sys.argv[1]) if len(sys.argv) > 1 else 3)), flush=True)
