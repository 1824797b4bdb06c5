import os, subprocess, sys, json
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
prev = os.path.join(ROOT, "tools", "_variants", "libphaneron_hip_prev.so")
for rep in range(3):
    for name, l in (("new", None), ("prev", prev)):
        out = []
        for v in ("wipe", "nowipe", "insets", "layer0"):
            env = dict(os.environ)
            if l: env["PHANERON_HIP_LIB"] = l
            r = subprocess.run([sys.executable, "tools/chan_bench.py", "400", "rgba", v], cwd=ROOT, env=env, capture_output=True, text=True)
            line = [x for x in r.stdout.splitlines() if x.startswith("{")][-1]
            out.append((v, json.loads(line)["us_per_frame"]))
        print(name, out, flush=True)
