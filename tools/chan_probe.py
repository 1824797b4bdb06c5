#!/usr/bin/env python3
"""In-kernel clock probe of ph_chan_compose_v210 (PH_CHAN_PROBE build of the library, never shipped): where the cycles of a
turn of phase 1 go.  python tools/chan_probe.py [variant]   (variants of tools/chan_bench.py)"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else "wipe"
    from phaneron_amd import build
    lib = build.build(extra_flags=["-DPH_CHAN_PROBE=1"], variant="chanprobe")  # objects are rebuilt only when a source changed
    env = dict(os.environ, PHANERON_HIP_LIB=lib)
    if os.environ.get("PH_CHAN_PROBE_CHILD") != "1":
        env["PH_CHAN_PROBE_CHILD"] = "1"
        sys.exit(subprocess.run([sys.executable, os.path.abspath(__file__), variant], env=env).returncode)
    sys.argv = [sys.argv[0], "20", "rgba", variant]
    import runpy
    runpy.run_path(os.path.join(ROOT, "tools", "chan_bench.py"), run_name="__main__")
    from phaneron_amd import capi
    buf = (C.c_ulonglong * (8 * 64))()
    capi.lib().ph_debug_chan_probe(buf, 8 * 64)
    names = ["next step / active ops", "issue (addresses + loads)", "finish (unpack, tables, filter)", "apply + between + store"]
    tot = [0] * 4
    turns = 0
    for t in range(1, 63):
        st = [buf[t * 8 + i] for i in range(5)]
        if not all(st) or st[4] <= st[0]:
            continue
        turns += 1
        for i in range(4):
            tot[i] += st[i + 1] - st[i]
        if t < 14:
            print(json.dumps({"turn": t, "cycles": [st[i + 1] - st[i] for i in range(4)], "gap_to_next": buf[(t + 1) * 8] - st[4] if buf[(t + 1) * 8] else None}))
    ph = (C.c_ulonglong * (256 * 16))()
    capi.lib().ph_debug_chan_phase(ph, 256 * 16)
    rows = [[ph[b * 16 + i] for i in range(16)] for b in range(256) if ph[b * 16 + 5]]
    if rows:
        t0 = min(r[0] for r in rows)
        names_p = ["reader table load + barrier", "phase 1 (wave 0)", "wait for the workgroup", "writer table load + barrier", "phase 2"]
        mean = lambda f: round(sum(f(r) for r in rows) / len(rows) / 100.0, 2)  # s_memrealtime ticks at 100 MHz -> us
        rec = {"workgroups": len(rows), "us_mean": {n: mean(lambda r, i=i: r[i + 1] - r[i]) for i, n in enumerate(names_p)},
               "start_skew_us": mean(lambda r: r[0] - t0), "end_us_after_first_start": {"mean": mean(lambda r: r[5] - t0),
               "max": round(max(r[5] - t0 for r in rows) / 100.0, 2)}}
        if all(r[7] for r in rows):  # the batch kernel stamps two more points: halo columns' words asked for (7), halo tables made (6)
            rec["us_mean"] = {"jobs' tables + halo words asked for": mean(lambda r: r[7] - r[0]), "reader table load + barrier": mean(lambda r: r[1] - r[7]),
                              "halo tables + barrier": mean(lambda r: r[6] - r[1]), "phase 1 (wave 0)": mean(lambda r: r[2] - r[6]),
                              **{n: rec["us_mean"][n] for n in names_p[2:]}}
        print(json.dumps(rec))
    print(json.dumps({"variant": variant, "turns": turns, "mean_cycles": {n: round(tot[i] / max(turns, 1)) for i, n in enumerate(names)}}))


if __name__ == "__main__":
    main()
