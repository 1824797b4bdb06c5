import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import numpy as np, torch
from phaneron_amd import capi
import config_bench
ctx = capi.Context(0)
for r in config_bench.measure(ctx, torch, np, capi, "best", reps=300): print(json.dumps(r))
ctx.close()
