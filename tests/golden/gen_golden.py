#!/usr/bin/env python3
"""Generate tests/golden/{host_maths.json,kernels.npz} by RUNNING THE REFERENCE in the build
container (needs /root/reference; never runs on the GPU box):

  1. oracle/refbuild/build_ref.sh   - the reference's OpenCL C kernel text, compiled unmodified
                                      for x86 and linked with oracle/refbuild/ocl_shim.cpp
  2. oracle/refbuild/ts_strip.py    - the reference's TypeScript host maths made runnable on node 12
  3. tests/golden/ref_host_dump.js  - colourMaths / Transform / fillBuf results -> host_maths.json
  4. every case of tests/golden/cases.py through the reference kernels     -> kernels.npz

Only data (inputs are seeds; outputs are arrays / hashes) is written to the repo.
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import orc  # noqa: E402  (ctypes binding of oracle/_ref)
import cases  # noqa: E402
import frames  # noqa: E402


def f32_from_hex(words):
    return np.array([int(w, 16) for w in words], np.uint32).view(np.float32)


def main():
    if not os.path.isdir("/root/reference/src/process"):
        raise SystemExit("reference checkout missing: goldens can only be generated in the build container")
    subprocess.run([os.path.join(ROOT, "oracle", "refbuild", "build_ref.sh")], check=True)
    subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "refbuild", "ts_strip.py")], check=True)
    lut_dir = os.path.join(ROOT, "oracle", "_ref", "work", "luts")
    os.makedirs(lut_dir, exist_ok=True)
    env = dict(os.environ, REF_LUT_DIR=lut_dir)
    host = subprocess.run(["node", os.path.join(HERE, "ref_host_dump.js")], check=True, capture_output=True,
                          text=True, env=env).stdout
    hm = json.loads(host)
    with open(os.path.join(HERE, "host_maths.json"), "w") as f:
        json.dump(hm, f, indent=1, sort_keys=True)
        f.write("\n")

    def lut(kind, spec):
        return np.fromfile(os.path.join(lut_dir, "%s_%s.bin" % (kind, spec)), np.float32)

    ref = orc.ref()
    out = {}
    for c in cases.CASES:
        inp = cases.inputs(c)
        op = c["op"]
        if op == "v210_read":
            cm = f32_from_hex(hm["ycbcr2rgb"][c["spec"] + "/10"])
            gm = f32_from_hex(hm["rgb2rgb"]["%s->%s" % (c["spec"], c["out_spec"])])
            o = np.zeros(c["w"] * c["h"] * 4, np.float32)
            ref.ref_v210_read(inp["words"], o, c["w"], c["h"], cm, lut("g2l", c["spec"]), gm)
        elif op == "v210_write":
            cm = f32_from_hex(hm["rgb2ycbcr"][c["spec"] + "/10"])
            o = inp["dst"].copy()
            ref.ref_v210_write(inp["rgba"].reshape(-1), o, c["w"], c["h"], c["interlace"], cm, lut("l2g", c["spec"]))
        elif op == "pack_read":
            rng = orc.FORMAT_RANGE[c["fmt"]]
            cm = None if rng is None else f32_from_hex(hm["ycbcr2rgb"]["%s/%d" % (c["spec"], rng[0])])
            gm = f32_from_hex(hm["rgb2rgb"]["%s->%s" % (c["spec"], c["out_spec"])])
            o = np.zeros(c["w"] * c["h"] * 4, np.float32)
            pl = [np.ascontiguousarray(p) for p in inp["planes"]]
            ptrs = [p.ctypes.data for p in pl] + [None] * (3 - len(pl))
            assert 0 == ref.ref_pack_read(orc.FORMATS[c["fmt"]], ptrs[0], ptrs[1], ptrs[2], o, c["w"], c["h"],
                                          None if cm is None else cm.ctypes.data, lut("g2l", c["spec"]), gm)
        elif op == "pack_write":
            rng = orc.FORMAT_RANGE[c["fmt"]]
            cm = None if rng is None else f32_from_hex(hm["rgb2ycbcr"]["%s/%d" % (c["spec"], rng[0])])
            pl = [p.copy() for p in inp["dst"]]
            ptrs = [p.ctypes.data for p in pl] + [None] * (3 - len(pl))
            assert 0 == ref.ref_pack_write(orc.FORMATS[c["fmt"]], inp["rgba"].reshape(-1), ptrs[0], ptrs[1], ptrs[2],
                                           c["w"], c["h"], c["interlace"], None if cm is None else cm.ctypes.data,
                                           lut("l2g", c["spec"]))
            o = np.concatenate(pl)
        elif op == "yadif":
            o = np.zeros(c["w"] * c["h"] * 4, np.float32)
            ref.ref_yadif(inp["prev"].reshape(-1), inp["cur"].reshape(-1), inp["next"].reshape(-1), c["w"], c["h"],
                          c["parity"], c["tff"], c["skip"], o)
        elif op == "transform":
            t = hm["transform"][c["tp"]]
            assert (t["width"], t["height"]) == (c["mw"], c["mh"])
            o = np.zeros(c["ow"] * c["oh"] * 4, np.float32)
            ref.ref_transform(inp["img"].reshape(-1), c["iw"], c["ih"], f32_from_hex(t["matrix"]), o, c["ow"], c["oh"])
        elif op == "resize":
            flip = np.array([1.0 if c["fh"] else 0.0, -1.0 if c["fh"] else 1.0, 1.0 if c["fv"] else 0.0,
                             -1.0 if c["fv"] else 1.0], np.float32)
            o = np.zeros(c["ow"] * c["oh"] * 4, np.float32)
            ref.ref_resize(inp["img"].reshape(-1), c["iw"], c["ih"], c["scale"], c["ox"], c["oy"], flip, o,
                           c["ow"], c["oh"])
        elif op == "combine":
            ls = [l.reshape(-1) for l in inp["layers"]]
            o = np.zeros(c["w"] * c["h"] * 4, np.float32)
            assert 0 == ref.ref_combine(len(ls), orc._ptr_array(ls), c["w"], c["h"], o)
        elif op in ("dissolve", "mixer", "wipe"):
            fn = dict(dissolve=ref.ref_transition_dissolve, mixer=ref.ref_mixer, wipe=ref.ref_wipe)[op]
            o = np.zeros(c["w"] * c["h"] * 4, np.float32)
            fn(inp["in0"].reshape(-1), inp["in1"].reshape(-1), c.get("mix", c.get("wipe")), c["w"], c["h"], o)
        elif op == "twipe":
            o = np.zeros(c["w"] * c["h"] * 4, np.float32)
            ref.ref_transition_wipe(inp["in0"].reshape(-1), inp["in1"].reshape(-1), inp["mask"].reshape(-1),
                                    c["w"], c["h"], o)
        else:
            raise KeyError(op)
        out[c["name"]] = o

    # full-frame known-answer test implied by the reference's round-trip scripts:
    # 1080p ramp -> read(709->709) -> write(709) must reproduce the input bytes.
    w, h = 1920, 1080
    ramp = frames.v210_ramp(w, h)
    rgba = np.zeros(w * h * 4, np.float32)
    ref.ref_v210_read(ramp, rgba, w, h, f32_from_hex(hm["ycbcr2rgb"]["709/10"]), lut("g2l", "709"),
                      f32_from_hex(hm["rgb2rgb"]["709->709"]))
    back = np.zeros_like(ramp)
    ref.ref_v210_write(rgba, back, w, h, 0, f32_from_hex(hm["rgb2ycbcr"]["709/10"]), lut("l2g", "709"))
    kat = {
        "ramp_1080p_read709_rgba_sha256": hashlib.sha256(rgba.tobytes()).hexdigest(),
        "ramp_1080p_roundtrip_identical": bool(np.array_equal(back, ramp)),
        "ramp_1080p_sha256": hashlib.sha256(ramp.tobytes()).hexdigest(),
    }
    assert kat["ramp_1080p_roundtrip_identical"], "reference round trip is expected to be lossless"
    assert kat["ramp_1080p_sha256"] == hm["ramp"]["1920x1080"], "frames.v210_ramp != reference fillBuf"
    # the reference's own round-trip scripts (src/process/test/*.ts): test pattern -> read -> write;
    # they print Buffer.compare() without asserting it - record what the reference kernels give.
    for fmt, (w, h), spec in (("yuv422p10", (1920, 1080), "709"), ("yuv420p", (1920, 1080), "709"),
                              ("nv12", (1920, 1080), "709"), ("yuv422p8", (718, 480), "709"),
                              ("rgba8", (1920, 1080), "sRGB"), ("bgra8", (1920, 1080), "sRGB")):
        planes = frames.pack_ramp(fmt, w, h)
        assert hashlib.sha256(np.concatenate(planes).tobytes()).hexdigest() == hm["ramp_fmt"]["%s/%dx%d" % (fmt, w, h)], \
            "frames.pack_ramp(%s) != reference fillBuf" % fmt
        rng = orc.FORMAT_RANGE[fmt]
        rcm = None if rng is None else f32_from_hex(hm["ycbcr2rgb"]["%s/%d" % (spec, rng[0])])
        wcm = None if rng is None else f32_from_hex(hm["rgb2ycbcr"]["%s/%d" % (spec, rng[0])])
        rgba = np.zeros(w * h * 4, np.float32)
        ptrs = [p.ctypes.data for p in planes] + [None] * (3 - len(planes))
        ref.ref_pack_read(orc.FORMATS[fmt], ptrs[0], ptrs[1], ptrs[2], rgba, w, h, None if rcm is None else rcm.ctypes.data,
                          lut("g2l", spec), f32_from_hex(hm["rgb2rgb"]["%s->%s" % (spec, spec)]))
        back = [np.zeros_like(p) for p in planes]
        bptrs = [p.ctypes.data for p in back] + [None] * (3 - len(back))
        ref.ref_pack_write(orc.FORMATS[fmt], rgba, bptrs[0], bptrs[1], bptrs[2], w, h, 0,
                           None if wcm is None else wcm.ctypes.data, lut("l2g", spec))
        kat["%s_%dx%d_rgba_sha256" % (fmt, w, h)] = hashlib.sha256(rgba.tobytes()).hexdigest()
        kat["%s_%dx%d_back_sha256" % (fmt, w, h)] = hashlib.sha256(np.concatenate(back).tobytes()).hexdigest()
        kat["%s_%dx%d_roundtrip_identical" % (fmt, w, h)] = bool(all(np.array_equal(a, b) for a, b in zip(planes, back)))
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1, sort_keys=True)
        f.write("\n")
    # host-logic trace: the reference's own operator + dispatcher code against the recording mock
    tr = subprocess.run(["node", os.path.join(ROOT, "node", "test", "scenario.js"),
                         os.path.join(ROOT, "oracle", "_ref", "work", "js")], check=True, capture_output=True, text=True).stdout
    with open(os.path.join(HERE, "host_trace.json"), "w") as f:
        json.dump(json.loads(tr), f, indent=0, sort_keys=True)
        f.write("\n")
    # the reference's own video valves (mixer.ts, transitioner.ts, combiner.ts, blackSilence.ts: oracle/refbuild/ts_erase.py)
    # driven frame by frame against the same recording mock
    subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "refbuild", "ts_erase.py")], check=True, capture_output=True)
    tr = subprocess.run(["node", os.path.join(ROOT, "node", "test", "valve_scenario.js"),
                         os.path.join(ROOT, "oracle", "_ref", "work", "js")], check=True, capture_output=True, text=True).stdout
    with open(os.path.join(HERE, "valve_trace.json"), "w") as f:
        json.dump(json.loads(tr), f, indent=0, sort_keys=True)
        f.write("\n")
    # the same schedule with the reference's ToRGBA in front (v210 sources, one Loader per producer) and its FromRGBA + saveFrame
    # behind: the whole per-frame video path of a channel, v210 in, v210 out
    tr = subprocess.run(["node", os.path.join(ROOT, "node", "test", "valve_scenario.js"),
                         os.path.join(ROOT, "oracle", "_ref", "work", "js"), "--v210"], check=True, capture_output=True, text=True).stdout
    with open(os.path.join(HERE, "channel_trace.json"), "w") as f:
        json.dump(json.loads(tr), f, indent=0, sort_keys=True)
        f.write("\n")
    # which pack format each `read` / `write` program of that trace belongs to, keyed by the fingerprint the
    # mock recorded of the kernel text (a hash, so the trace can be replayed where the text cannot go)
    shas = {}
    for fmt in ("v210", "yuv422p10", "yuv422p8", "yuv420p", "nv12", "rgba8", "bgra8"):
        txt = open(os.path.join(ROOT, "oracle", "_ref", "work", "cl", fmt + ".cl")).read()
        shas[hashlib.sha256(txt.encode()).hexdigest()[:16]] = fmt
    with open(os.path.join(HERE, "kernel_text_sha.json"), "w") as f:
        json.dump(shas, f, indent=1, sort_keys=True)
        f.write("\n")
    np.savez_compressed(os.path.join(HERE, "kernels.npz"), **out)
    print("wrote %d kernel cases, host_maths.json, kat.json" % len(out))


if __name__ == "__main__":
    main()
