#!/usr/bin/env python3
"""EXPERIMENT (round 6): a share of the headline kernel's table lookups through the vector-memory path (-DPH_FUSED_TA=1|2|3: that many of every quad's
eighteen lookups per layer read the table's plain f32 copy in global memory instead of its compressed form in LDS), against the shipped kernel,
A/B alternating in one call.  python tools/ta_ab.py [passes] [variant ...]   (variants built in the container:
python -c "from phaneron_amd import build; build.build(variant='ta2', extra_flags=['-DPH_FUSED_TA=2'])")"""
import json
import os
import subprocess
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from phaneron_amd import build  # noqa: E402


def last_json(cmd, lib, extra_env=None):
    env = dict(os.environ)
    if lib:
        env["PHANERON_HIP_LIB"] = lib
    env.update(extra_env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else {"error": r.stderr[-300:]}


def main():
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    variants = sys.argv[2:] or ["ta1", "ta2", "ta3"]
    libs = [("shipped", None)] + [(v, build.variant_path(v)) for v in variants]
    quick = os.environ.get("PH_AB_HEADLINE_ONLY") == "1"
    for rep in range(passes):
        for name, lib in libs:
            b = last_json([sys.executable, "bench.py", "--steps", "400", "--warmup", "20", "--cpu-seconds", "0", "--no-secondary", "--no-traffic"], lib)
            rec = {"build": name, "pass": rep, "headline_us": round(1e3 * b["roofline"]["avg_launch_ms"], 2) if "roofline" in b else b}
            if not quick:
                c = last_json([sys.executable, "tools/chan_bench.py", "400", "rgba", "wipe"], lib)
                c4 = last_json([sys.executable, "tools/chan_bench.py", "400", "rgba", "wipe"], lib, {"PH_CHAN_BENCH_JOBS": "4"})
                u = last_json([sys.executable, "tools/up_bench.py", "150"], lib)
                rec.update({"config2_chan_us": c.get("us_per_frame"), "config2_chan_x4_us": c4.get("us_per_frame"),
                            "deint_rgb_us_per_frame": u.get("deint_rgb_us_per_frame"),
                            "compose_up_rgb_pair_us_per_field": u.get("compose_up_rgb_pair_us_per_field")})
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
