"""GPU tests (-m gpu) of BASELINE config 5's shape on ONE GPU.  tools/route_bench.py --check verifies, bit for
bit (position-sensitive fingerprint computed on the device), that what arrives as a channel's routed layer is the
source channel's combiner output.
  - one rank, routes alias local buffers (the reference's own case: addRef on the same buffer);
  - one rank, --loopback: the same routes through the library's ROUTE path (ph_route_*: RCCL send / recv to the own
    rank on the communication stream, event-ordered against the process queue) - RCCL refuses two ranks on one
    device, so this is how the C path, its stream ordering and librccl itself run on a single-GPU box;
  - two processes on cuda:0 with gloo staged through host memory: the cross-rank plan (who sends what to whom)."""
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


def test_routes_through_the_library_route_path_rccl_loopback():
    out = run([sys.executable, "tools/route_bench.py", "--loopback", "--check", "--steps", "5", "--warmup", "2", "--width", "1920",
               "--height", "270"])
    assert "route check ok: 2 channels on 1 rank(s), ph_route (RCCL)" in out
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["path"].startswith("ph_route")
    # both channels' outputs travel: 2 sends + 2 receives of 1920x270 f32 RGBA per step
    assert line["routes_crossing_ranks_per_rank"] == 2 and line["route_bytes_per_rank_per_step"] == 4 * 1920 * 270 * 16


def test_route_results_do_not_depend_on_the_path(tmp_path):
    """The composited output of every channel after a few steps is the same whether routed frames alias or travel
    through RCCL (the routed layer is one step late either way)."""
    outs = []
    for extra in ([], ["--loopback"]):
        o = run([sys.executable, "tools/route_bench.py", "--steps", "4", "--warmup", "0", "--width", "1920", "--height", "270",
                 "--print-fingerprints"] + extra)
        outs.append([l for l in o.splitlines() if l.startswith("fingerprints")][-1])
    assert outs[0] == outs[1]


def test_routes_cross_ranks_world_2():
    out = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", "29541", "tools/route_bench.py", "--backend", "gloo", "--same-gpu", "--check",
               "--steps", "3", "--warmup", "1", "--width", "1920", "--height", "270"])
    assert "route check ok: 4 channels on 2 rank(s)" in out
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    # each rank sends its two channels' outputs and receives two: 4 frames of 1920x270 f32 RGBA per step
    assert line["routes_crossing_ranks_per_rank"] == 2
    assert line["route_bytes_per_rank_per_step"] == 4 * 1920 * 270 * 16


def test_release_right_after_send_keeps_the_routed_frame_intact():
    """The source rank's `send(frame); frame.release()`: the block goes back to the pool and the very next
    createBuffer of that size gets it - its first writer must still run AFTER RCCL has read the frame.  The pool orders
    the queues behind the transfers in flight before it hands a recycled block out."""
    import numpy as np
    from phaneron_amd import capi
    ctx = capi.Context(0)
    route = capi.Route(ctx, capi.route_unique_id(), 0, 1)
    assert route.comm_count() == 1
    n = 96 << 20
    for attempt in range(3):
        frame = (np.arange(n // 4, dtype=np.uint32) * np.uint32(2654435761 + attempt)).view(np.uint8)
        src = ctx.create_buffer(n)
        dst = ctx.create_buffer(n + 4096)             # another size: never the recycled block
        src.host_access("writeonly", capi.QUEUE_LOAD, frame)
        route.after_queue(capi.QUEUE_LOAD)
        with route.group():
            route.send(src.device_ptr(), 0, n)
            route.recv(dst.device_ptr(), 0, n)
        block = src.device_ptr()
        src.release()                                 # right after the send, as a channel does
        again = ctx.create_buffer(n)
        assert again.device_ptr() == block            # the pool handed the same block out ...
        again.host_access("writeonly", capi.QUEUE_LOAD, np.full(n, 0xEE, np.uint8))  # ... and its new owner overwrites it at once
        route.wait()
        ctx.wait(capi.QUEUE_LOAD)
        dst.host_access("readonly", capi.QUEUE_UNLOAD)
        ctx.wait(capi.QUEUE_UNLOAD)
        assert np.array_equal(dst.host()[:n], frame), "attempt %d: the routed frame was overwritten under the transfer" % attempt
        again.release(), dst.release()
    route.destroy()
    ctx.close()
