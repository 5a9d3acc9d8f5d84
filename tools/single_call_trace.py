#!/usr/bin/env python3
"""GPU box: BASELINE configs[1] as stated -- ONE WBFM / MFM / FM call per buffer -- for a kernel trace
(rocprofv3 --kernel-trace --stats -- python tools/single_call_trace.py): which launches the call consists of and
how long each one takes when it runs alone on the chip."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "radio-core_amd"))
import bench  # noqa: E402

print(json.dumps(bench.measure_cfg2_single(reps=int(sys.argv[1]) if len(sys.argv) > 1 else 100)))
