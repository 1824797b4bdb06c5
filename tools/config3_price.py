#!/usr/bin/env python3
"""VERDICT r3 item 3, first candidate, priced before it is built: the fields' KEPT lines (half of a de-interlaced field's lines are
copies of the source frame's) carried between config 3's two kernels as three 16-bit reader-table indices, only the interpolated
lines as floats.  Two timing builds of the library (never shipped; pixels wrong):
  -DPH_DEINT_PRICE_INDEX=1  the reader stores its kept lines as 6 bytes per pixel (what it would save: a quarter of its stores)
  -DPH_UP_PRICE_INDEX=1     the compositor runs the reader's three lookups + gamut matrix on the texels of kept lines (what it would pay)
against the shipped kernels, each timed alone (tools/up_bench.py).  python tools/config3_price.py   (on the GPU box)"""
import json
import os
import subprocess
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from phaneron_amd import build  # noqa: E402


def run(lib):
    env = dict(os.environ)
    if lib:
        env["PHANERON_HIP_LIB"] = lib
    r = subprocess.run([sys.executable, "tools/up_bench.py", "200"], cwd=ROOT, env=env, capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else {"error": r.stderr[-400:]}


def main():
    lib = build.build(extra_flags=["-DPH_DEINT_PRICE_INDEX=1", "-DPH_UP_PRICE_INDEX=1"], variant="config3price")
    for name, l in (("shipped", None), ("kept lines as 16-bit indices (timing build)", lib), ("shipped", None), ("kept lines as 16-bit indices (timing build)", lib)):
        r = run(l)
        print(json.dumps({"build": name, "reader_us_per_frame": r.get("deint_rgb_us_per_frame"), "compositor_us_per_field": r.get("compose_up_rgb_us_per_field"),
                          "compositor_pair_us_per_field": r.get("compose_up_rgb_pair_us_per_field"), "raw": r if "error" in r else None}), flush=True)


if __name__ == "__main__":
    main()
