"""The drop-in boundary in the reference's own host language (node/):
  CPU: the JS operator layer + dispatcher produce the SAME trace against a recording mock as the
       reference's own (type-stripped) operator code did when the goldens were generated;
       the N-API addon builds and loads and exposes the nodencl-shaped entry points.
  GPU: the JS layer on the real addon reproduces the oracle bit for bit."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
NODE = shutil.which("node")
needs_node = pytest.mark.skipif(NODE is None, reason="node is not installed")


@needs_node
def test_js_operator_layer_matches_reference_trace():
    out = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "scenario.js"), os.path.join(ROOT, "node")],
                         check=True, capture_output=True, text=True).stdout
    got = json.loads(out)
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "host_trace.json")))
    assert len(got) == len(want)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, "trace event %d differs:\n got  %s\n want %s" % (i, json.dumps(g)[:400], json.dumps(w)[:400])


@needs_node
def test_reference_trace_still_reproducible_here():
    """Where the reference checkout exists the golden trace must regenerate identically."""
    ref_js = os.path.join(ROOT, "oracle", "_ref", "work", "js", "clJobQueue.js")
    if not os.path.exists(ref_js):
        pytest.skip("oracle/_ref/js not built (reference checkout absent)")
    out = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "scenario.js"), os.path.dirname(ref_js)],
                         check=True, capture_output=True, text=True).stdout
    assert json.loads(out) == json.load(open(os.path.join(ROOT, "tests", "golden", "host_trace.json")))


@needs_node
def test_napi_addon_builds_loads_and_refuses_without_gpu():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "node"))
    from phaneron_amd import build as hipbuild
    hipbuild.build()
    subprocess.run([sys.executable, os.path.join(ROOT, "node", "build.py")], check=True)
    js = ("const a=require('%s');"
          "const want=['abiVersion','createContext','contextInfo','createBuffer','bufAddRef','bufRelease','bufRefCount',"
          "'hostAccess','waitFinish','createProgram','runProgram','bufferStats','queueWaitQueue','downloadAsync',"
          "'eventRecord','eventWait','eventDone','waitFinishSpin'];"
          "for (const k of want) if (typeof a[k] !== 'function') { console.log('missing', k); process.exit(2) }"
          "console.log(a.abiVersion())") % os.path.join(ROOT, "node", "phaneron_napi.node")
    r = subprocess.run([NODE, "-e", js], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip() == "1"
    import torch
    if not torch.cuda.is_available():
        js = ("const {clContext}=require('%s'); const c=new clContext({deviceIndex:0});"
              "c.initialise().then(()=>{console.log('unexpected');process.exit(3)},e=>{console.log(e.message);})") % \
            os.path.join(ROOT, "node", "index.js")
        r = subprocess.run([NODE, "-e", js], capture_output=True, text=True)
        assert "no HIP device" in r.stdout, r.stdout + r.stderr


# the round trips gen_golden.py recorded from the reference kernels (kat.json "<fmt>_<w>x<h>_*")
FORMAT_KATS = [("yuv422p10", 1920, 1080, "709"), ("yuv420p", 1920, 1080, "709"), ("nv12", 1920, 1080, "709"),
               ("yuv422p8", 718, 480, "709"), ("rgba8", 1920, 1080, "sRGB"), ("bgra8", 1920, 1080, "sRGB")]


@needs_node
@pytest.mark.gpu
def test_node_layer_end_to_end_on_gpu(tmp_path):
    import frames
    from oracle import orc
    w, h, n = 1920, 96, 4
    layers = [frames.v210_random(w, h, frames.layer_seed(3, i)) for i in range(n)]
    for i, l in enumerate(layers):
        l.tofile(tmp_path / ("layer%d.bin" % i))
    pip = dict(flipH=False, flipV=False, anchorX=0.0, anchorY=0.0, scaleX=0.5, scaleY=0.5, rotate=-0.0,
               offsetX=-0.25, offsetY=0.25)
    yw, yh = 320, 48
    fields = [frames.rgba_random(yw, yh, 700 + i) for i in range(4)]
    for i, f in enumerate(fields):
        f.tofile(tmp_path / ("field%d.bin" % i))
    job = dict(channel=dict(width=w, height=h, layers=["layer%d.bin" % i for i in range(n)], readSpec="709",
                            writeSpec="2020", pip=pip),
               yadif=dict(width=yw, height=yh, frames=["field%d.bin" % i for i in range(4)], tff=True),
               formats=[dict(fmt=f, width=fw, height=fh, spec=sp) for f, fw, fh, sp in FORMAT_KATS],
               staged=dict(width=1920, height=24, layers=3, frames=5, readSpec="709", writeSpec="2020"))
    staged_src = [[frames.v210_random(1920, 24, frames.layer_seed(8 + f, l)) for l in range(3)] for f in range(5)]
    for f, ls in enumerate(staged_src):
        for l, words in enumerate(ls):
            words.tofile(tmp_path / ("staged_f%d_l%d.bin" % (f, l)))
    (tmp_path / "job.json").write_text(json.dumps(job))
    r = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "gpu_run.js"), str(tmp_path)], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.loads((tmp_path / "result.json").read_text())
    assert res["rampCompare"] == 0  # the reference scripts' "Compare returned 0"
    assert "gfx950" in res["platform"]["devices"][0]["name"]

    rd = (orc.ycbcr2rgb_matrix("709"), orc.gamma2linear_lut("709"), orc.rgb2rgb_matrix("709", "2020"))
    rgba = [orc.v210_read(l, w, h, *rd) for l in layers]
    m = orc.transform_matrix(w, h, False, False, 0.0, 0.0, 0.5, 0.5, -0.25, 0.25, -0.0)
    rgba[1] = orc.transform(rgba[1], m, w, h)
    want = orc.v210_write(orc.combine(rgba), w, h, 0, orc.rgb2ycbcr_matrix("2020"), orc.linear2gamma_lut("2020"))
    got = np.fromfile(tmp_path / "channel_out.bin", np.uint32)
    assert np.array_equal(got, want)

    # yadif send_field, tff: outputs for cur = field1 and field2, two each (yadif.ts:125-145)
    assert res["yadifTimestamps"] == [2, 3, 4, 5]
    k = 0
    for cur in (1, 2):
        for second in (False, True):
            parity = 1 ^ (0 if second else 1)
            want = orc.yadif(fields[cur - 1], fields[cur], fields[cur + 1], parity, True, False)
            got = np.fromfile(tmp_path / ("yadif_out%d.bin" % k), np.float32)
            assert np.array_equal(got.view(np.uint32), want.reshape(-1).view(np.uint32)), k
            k += 1

    # the other formats' round-trip scripts: same bytes back, and RGBA / output hashes equal to what the
    # reference's own kernels produced for the same pattern
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "kat.json")))
    for f, fw, fh, _ in FORMAT_KATS:
        got = res["formats"][f]
        key = "%s_%dx%d" % (f, fw, fh)
        assert got["rgbaSha256"] == kat[key + "_rgba_sha256"], f
        assert got["backSha256"] == kat[key + "_back_sha256"], f
        assert (got["compare"] == 0) == kat[key + "_roundtrip_identical"], f

    # staged ring + fused channel program: each frame's output equals the oracle chain for that frame
    assert res["stagedOrder"] == [0, 1, 2, 3, 4]
    wr = (orc.rgb2ycbcr_matrix("2020"), orc.linear2gamma_lut("2020"))
    for f, ls in enumerate(staged_src):
        want = orc.pipeline_v210_combine(ls, 1920, 24, *rd, *wr)
        assert np.array_equal(np.fromfile(tmp_path / ("staged_out%d.bin" % f), np.uint32), want), f
    assert res["fusedCallbackFired"] is True
    assert np.array_equal(np.fromfile(tmp_path / "fused_queue_out.bin", np.uint32),
                          orc.pipeline_v210_combine(staged_src[0], 1920, 24, *rd, *wr))


@needs_node
def test_valve_graph_host_logic_on_mock():
    """node/valves (Mixer -> Transitioner -> Combiner) on the recording mock: what it asks
    the device to do follows mixer.ts:209-223, transitioner.ts:143-176,269 and combiner.ts:211-254."""
    out = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "valves_scenario.js")], capture_output=True, text=True,
                         check=True).stdout
    d = json.loads(out)
    # the combiner renumbers its output 0.. and hands out exactly one reference (no route forks)
    assert [(o["ts"], o["refs"]) for o in d["outputs"]] == [(i, 1) for i in range(7)]
    names = [k["name"] for k in d["kernels"]]
    assert names.count("combine_3") == 7 and names.count("transition_dissolve") == 4
    # dissolve of 4 frames: numFrames = 3, mix = 1 - cur/3, keyed on the incoming source's timestamp
    dis = [k for k in d["kernels"] if k["name"] == "transition_dissolve"]
    assert [k["ts"] for k in dis] == [300, 301, 302, 303]
    want = [1.0, float(np.float32(1.0 - 1 / 3)), float(np.float32(1.0 - 2 / 3)), 0.0]
    assert [k["mix"] for k in dis] == want
    # every transform runs under its producer's timestamp; the PiP layer uploads a different matrix
    xf = [k for k in d["kernels"] if k["name"] == "transform"]
    assert len({k["matrix"] for k in xf}) == 2
    assert sorted(k["ts"] for k in xf if k["ts"] < 200) == list(range(100, 107))
    # the empty layer contributes its black frame (the same buffer every time) as the third input
    third = {k["inputs"][2] for k in d["kernels"] if k["name"] == "combine_3"}
    assert len(third) == 1
    # the transitioner reports the source timestamps per step (layerUpdate), [] for the empty layer
    assert {"layer": "L2", "ts": [202, 300]} in d["layerEvents"] and {"layer": "L3", "ts": []} in d["layerEvents"]
    # nothing leaks: every source frame was released; only the black frames and matrices stay alive
    assert d["leakedFrames"] == []
    assert sorted(o["owner"].split("-")[0] for o in d["liveOwners"]) == ["black"] * 4 + ["transformMatrix"] * 3

    # second graph: a source that ends, a wipe with a mask source, one route fork
    e = d["second"]
    assert [(o["ts"], o["refs"]) for o in e["outputs"]] == [(i, 1) for i in range(6)]   # one fork: one reference
    names = [k["name"] for k in e["kernels"]]
    assert names.count("transition_wipe") == 6 and names.count("combine_2") == 6
    assert all(k["inputs"] == ["input0", "input1", "maskIn"] for k in e["kernels"] if k["name"] == "transition_wipe")
    # the ended layer keeps its place in the composite as the transitioner's black frame (transitioner.ts:186-192)
    assert e["combineFirstInputs"][3:] == [e["blackId"]] * 3 and e["blackId"] not in e["combineFirstInputs"][:3]
    assert names.count("transform") == 3 * 4 + 3 * 3          # the ended source is no longer transformed
    assert e["leakedFrames"] == [] and e["forksAfterRelease"] == 0


def mixer_matrix(w, h, p):
    """Mixer.mixVidValve's parameter mapping (mixer.ts:209-223) into the Transform matrix"""
    return orc_mod().transform_matrix(w, h, False, False, p["anchor"]["x"] - 0.5, p["anchor"]["y"] - 0.5, p["fill"]["xScale"],
                                      p["fill"]["yScale"], -p["fill"]["xOffset"], -p["fill"]["yOffset"], -p["rotation"] / 360.0)


def orc_mod():
    from oracle import orc
    return orc


@needs_node
@pytest.mark.gpu
def test_valve_graph_on_gpu(tmp_path):
    """The same graph on the real addon: every output frame equals the oracle's chain for that frame."""
    import frames
    orc = orc_mod()
    w, h, nf = 192, 64, 7
    default = dict(anchor=dict(x=0, y=0), rotation=0, fill=dict(xOffset=0, yOffset=0, xScale=1, yScale=1), volume=1)
    pip = dict(anchor=dict(x=0.25, y=0.75), rotation=30, fill=dict(xOffset=0.25, yOffset=-0.125, xScale=0.5, yScale=0.5),
               volume=1)
    src = {n: [frames.rgba_random(w, h, 9000 + 100 * k + i) for i in range(nf)] for k, n in enumerate(("A", "B0", "B1"))}
    for n, fs_ in src.items():
        for i, f in enumerate(fs_):
            f.tofile(tmp_path / ("%s_%d.bin" % (n, i)))
    job = dict(width=w, height=h, frames=nf, pip=pip, dissolveAt=2, dissolveLen=4, cutAt=6)
    (tmp_path / "job.json").write_text(json.dumps(job))
    r = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "valves_run.js"), str(tmp_path)], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.loads((tmp_path / "result.json").read_text())
    assert res["stamps"] == list(range(nf))
    md, mp = mixer_matrix(w, h, default), mixer_matrix(w, h, pip)
    black = np.zeros((h, w, 4), np.float32)
    b1 = 0  # frames of B1 consumed so far
    for f in range(nf):
        a = orc.transform(src["A"][f], md, w, h)
        if f < 2:
            b = orc.transform(src["B0"][f], mp, w, h)
        elif f < 6:
            mix = np.float32(1.0 - (f - 2) / 3)
            b = orc.transition_dissolve(orc.transform(src["B0"][f], mp, w, h), orc.transform(src["B1"][b1], md, w, h), float(mix))
            b1 += 1
        else:
            b = orc.transform(src["B1"][b1], md, w, h)
            b1 += 1
        want = orc.combine([a, b, black])
        got = np.fromfile(tmp_path / ("out_%d.bin" % f), np.float32)
        assert np.array_equal(got.view(np.uint32), want.reshape(-1).view(np.uint32)), f
