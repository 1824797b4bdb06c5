#!/usr/bin/env python3
"""What a compact, co-resident table form would cost the table kernels (priced before it is built): a timing
build of the library (-DPH_ABLATE=16, never shipped: ph_ldslut.h adds the five instructions the compact decode needs to every
lookup, results unchanged) against the shipped one, on config 2's channel kernel and on the headline.
  python tools/compact_lut_price.py        (on the GPU box; one JSON line per measurement)"""
import json
import os
import subprocess
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from phaneron_amd import build  # noqa: E402


def run(cmd, lib):
    env = dict(os.environ)
    if lib:
        env["PHANERON_HIP_LIB"] = lib
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else {"error": r.stderr[-400:]}


def main():
    lib = build.build(extra_flags=["-DPH_ABLATE=16"], variant="compactprice")
    for name, l in (("shipped", None), ("five more VALU per lookup", lib), ("shipped", None), ("five more VALU per lookup", lib)):
        c = run([sys.executable, "tools/chan_bench.py", "400", "rgba", "wipe"], l)
        b = run([sys.executable, "bench.py", "--steps", "400", "--warmup", "20", "--cpu-seconds", "0", "--no-secondary", "--no-traffic"], l)
        print(json.dumps({"build": name, "config2_chan_us_per_frame": c.get("us_per_frame"), "headline_us_per_frame": round(1e3 * b["roofline"]["avg_launch_ms"], 2) if "roofline" in b else b}), flush=True)


if __name__ == "__main__":
    main()
