"""GPU test (-m gpu) of BASELINE config 5's shape on ONE GPU: two processes (one per "GPU", both on
cuda:0, gloo staged through host memory because RCCL needs distinct devices), two channels each,
every channel's fourth layer routed from a channel of the OTHER process.  tools/route_bench.py --check
verifies that what arrives is the source channel's combiner output."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def run(cmd):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_routes_alias_locally_on_one_rank():
    out = run([sys.executable, "tools/route_bench.py", "--check", "--steps", "3", "--warmup", "1", "--width", "1920",
               "--height", "270"])
    assert "route check ok: 2 channels on 1 rank(s)" in out
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["routes_crossing_ranks_per_rank"] == 0 and line["route_bytes_per_rank_per_step"] == 0


def test_routes_cross_ranks_world_2():
    out = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", "29541", "tools/route_bench.py", "--backend", "gloo", "--same-gpu", "--check",
               "--steps", "3", "--warmup", "1", "--width", "1920", "--height", "270"])
    assert "route check ok: 4 channels on 2 rank(s)" in out
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    # each rank sends its two channels' outputs and receives two: 4 frames of 1920x270 f32 RGBA per step
    assert line["routes_crossing_ranks_per_rank"] == 2
    assert line["route_bytes_per_rank_per_step"] == 4 * 1920 * 270 * 16
