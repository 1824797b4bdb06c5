#!/usr/bin/env python3
"""The AMDGPU backend's scheduling strategies on the table kernels (A/B in one call): the default (maximise occupancy - which here is fixed at
four waves per SIMD by the 1024-lane workgroup and its LDS table, whatever the register count) against -amdgpu-sched-strategy=max-ilp and
max-memory-clause.  Builds a variant library per strategy and times the headline, config 2's channel kernel and config 3's two kernels.
  python tools/sched_ab.py [strategy ...]      (on the GPU box; one JSON line per build and pass)"""
import json
import os
import subprocess
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from phaneron_amd import build  # noqa: E402


def last_json(cmd, lib):
    env = dict(os.environ)
    if lib:
        env["PHANERON_HIP_LIB"] = lib
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else {"error": r.stderr[-300:]}


def main():
    strategies = sys.argv[1:] or ["max-ilp", "max-memory-clause"]
    libs = [("default", None)] + [(s, build.build(extra_flags=["-mllvm", "-amdgpu-sched-strategy=" + s], variant="sched_" + s.replace("-", "_"))) for s in strategies]
    for rep in range(2):
        for name, lib in libs:
            b = last_json([sys.executable, "bench.py", "--steps", "400", "--warmup", "20", "--cpu-seconds", "0", "--no-secondary", "--no-traffic"], lib)
            c = last_json([sys.executable, "tools/chan_bench.py", "400", "rgba", "wipe"], lib)
            u = last_json([sys.executable, "tools/up_bench.py", "150"], lib)
            print(json.dumps({"build": name, "pass": rep, "headline_us": round(1e3 * b["roofline"]["avg_launch_ms"], 2) if "roofline" in b else b,
                              "config2_chan_us": c.get("us_per_frame"), "deint_rgb_us_per_frame": u.get("deint_rgb_us_per_frame"),
                              "compose_up_rgb_pair_us_per_field": u.get("compose_up_rgb_pair_us_per_field"), "compose_up_rgb_us_per_field": u.get("compose_up_rgb_us_per_field")}), flush=True)


if __name__ == "__main__":
    main()
