#!/usr/bin/env python3
"""The fastest route of BASELINE configs 2 and 3 only (what bench.py reports as `secondary`), for runs under
rocprofv3 where the other routes' kernels would only dilute the statistics:
  rocprofv3 --kernel-trace --stats -- python tools/config_best.py
  tools/pmc_kernel.sh "<kernel substring>" python tools/config_best.py"""
import json
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from phaneron_amd import capi  # noqa: E402
import config_bench  # noqa: E402

ctx = capi.Context(0)
for r in config_bench.measure(ctx, torch, np, capi, "best", reps=300):
    print(json.dumps(r))
ctx.close()
