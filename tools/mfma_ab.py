#!/usr/bin/env python3
"""The CSC matrices on the matrix pipe (v_mfma_f32_4x4x1_16b_f32, -DPH_MFMA=1) against the shipped VALU form, A/B in one call:
the headline kernel, config 2's channel kernel (alone and four to a launch) and config 3's two kernels, alternating builds
within each pass.  (timing first, stop rule 3 us on the headline.)
  python tools/mfma_ab.py [passes] [variant ...]      (on the GPU box; the variant libraries were built in the container:
  python -c "from phaneron_amd import build; build.build(variant='mfma', extra_flags=['-DPH_MFMA=1'])")"""
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
    variants = sys.argv[2:] or ["mfma"]
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
