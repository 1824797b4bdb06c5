"""The drop-in boundary in the reference's own host language (node/).
In a deployment the layers above the boundary ARE the reference's files (src/process/*, src/clJobQueue.ts,
the valves); this repository ships the addon (ph_napi.c), the nodencl-shaped surface (index.js) and its own
small front end for tests and benchmarks (device.js, jobs.js, channel.js, staging.js).
  build container: the reference's own operator + dispatcher code, type-stripped, runs against a recording
       clContext; the kernel text it passes to createProgram resolves to the right precompiled kernel;
       the trace it leaves is the committed golden (tests/golden/host_trace.json).
  CPU: the addon builds, loads, exposes every entry point, refuses to run without a GPU; the JobBoard keeps
       the dispatcher contract.
  GPU: the golden trace replayed call by call on the real addon (reference counts included) gives the
       reference's known answers; the own front end reproduces the oracle bit for bit."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
NODE = shutil.which("node")
needs_node = pytest.mark.skipif(NODE is None, reason="node is not installed")


REF_JS = os.path.join(ROOT, "oracle", "_ref", "work", "js")
needs_ref_js = pytest.mark.skipif(not os.path.exists(os.path.join(REF_JS, "clJobQueue.js")),
                                  reason="the type-stripped reference exists only in the build container")
TRACE = os.path.join(ROOT, "tests", "golden", "host_trace.json")
TEXT_SHA = os.path.join(ROOT, "tests", "golden", "kernel_text_sha.json")
PACK_FORMATS = ["yuv422p10", "yuv422p8", "yuv420p", "nv12", "rgba8", "bgra8"]


def _build_addon():
    import sys
    from phaneron_amd import build as hipbuild
    hipbuild.build()
    subprocess.run([sys.executable, os.path.join(ROOT, "node", "build.py")], check=True)


@needs_node
@needs_ref_js
def test_reference_operator_code_still_leaves_the_golden_trace():
    """Where the reference checkout exists the golden trace must regenerate identically."""
    out = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "scenario.js"), REF_JS], check=True, capture_output=True,
                         text=True).stdout
    assert json.loads(out) == json.load(open(TRACE))


@needs_node
@needs_ref_js
def test_reference_operator_code_selects_the_right_kernels_through_the_addon():
    """The reference's unchanged Packer / ImageProcess (packer.ts:97-103, imageProcess.ts:69-72) hand their
    OpenCL text to createProgram; the addon's device-free resolver must pick the matching precompiled kernel for
    every one of them - the seven pack formats by their exact text."""
    _build_addon()
    out = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "scenario.js"), REF_JS, "--resolve"], check=True,
                         capture_output=True, text=True).stdout
    trace = json.loads(out)
    progs = [e for e in trace if e["op"] == "createProgram"]
    assert len(progs) == 27
    text_sha = json.load(open(TEXT_SHA))
    for e in progs:
        r = e["resolved"]
        if e["name"] in ("read", "write"):
            fmt = text_sha[e["srcSha"]]
            assert r == {"kernel": "%s_%s" % (fmt, e["name"]), "format": fmt, "how": "text"}, e
        else:
            assert r == {"kernel": e["name"], "format": None, "how": "name"}, e
    assert {e["resolved"]["format"] for e in progs if e["name"] == "read"} == {"v210"} | set(PACK_FORMATS)
    # apart from the resolver's verdicts this is the golden trace
    for e in trace:
        e.pop("resolved", None)
    assert trace == json.load(open(TRACE))


@needs_node
def test_job_board_keeps_the_dispatcher_contract():
    """node/jobs.js against the recording clContext (clJobQueue.ts:53-141): per-key order, first come
    first served including late arrivals, callbacks after the drain, unknown key throws, cancel fires callbacks and
    keeps entries - with and without coalescing of waiting flushes into one drain."""
    d = json.loads(subprocess.run([NODE, os.path.join(ROOT, "node", "test", "board_check.js")], check=True,
                                  capture_output=True, text=True).stdout)
    f32 = lambda x: float(np.float32(x))
    for mode in ("coalesced", "oneByOne"):
        r = d[mode]
        assert r["order"] == ["B2", "A1a", "A1b", "A3", "C4"]
        assert r["unknown"] == {"isError": True, "message": "Failed to run queue for id nobody ts 5"}
        assert r["pendingAfterCancel"] == 1
        assert [x for x in r["device"] if x != "wait"] == [f32(0.2), f32(0.1), f32(0.1), f32(0.3)]
        assert r["device"][-1] == "wait" and r["stats"]["kernels"] == 4 and r["stats"]["flushes"] == 3
    assert d["oneByOne"]["device"] == [f32(0.2), "wait", f32(0.1), f32(0.1), "wait", f32(0.3), "wait"]
    assert d["coalesced"]["device"] == [f32(0.2), f32(0.1), f32(0.1), "wait", f32(0.3), "wait"]  # two flushes, one drain
    # a rejecting waitFinish fails every flush of the drain, a failing job fails its own flush only (the job behind it is
    # skipped); in both cases every completion callback fires and the board serves the next flush
    assert d["failures"] == dict(outcome=["device lost", "device lost"], outcome2=["bad launch", "ok"], after="ok",
                                 fired=["A1", "B2", "C3a", "C3b", "D4", "E5"], pumpIdle=True)
    # the channel compositor's parameter mapping (mixer.ts:209-223, transitioner.ts:170,269)
    assert d["placement"] == dict(flipH=False, flipV=False, anchorX=-0.25, anchorY=0.25, scaleX=0.5, scaleY=0.5,
                                  rotate=-30 / 360.0, offsetX=-0.25, offsetY=0.125)
    assert d["dissolve"] == [1.0, 1.0 - 1 / 3, 1.0 - 2 / 3, 0.0, 0.0, 0.0]


@needs_node
def test_napi_addon_builds_loads_and_refuses_without_gpu():
    _build_addon()
    js = ("const a=require('%s');"
          "const want=['abiVersion','setOption','createContext','contextInfo','createBuffer','bufAddRef','bufRelease','bufRefCount',"
          "'hostAccess','waitFinish','createProgram','runProgram','bufferStats','queueWaitQueue','downloadAsync',"
          "'eventRecord','eventWait','eventDone','waitFinishSpin','resolveProgram','gammaLut','colourMatrix',"
          "'transformMatrix','planeBytes','routeUniqueId','routeInit','routeOp','runPrograms','runProgramsProgress','traceBegin','traceEnd'];"
          "for (const k of want) if (typeof a[k] !== 'function') { console.log('missing', k); process.exit(2) }"
          "console.log(a.abiVersion())") % os.path.join(ROOT, "node", "phaneron_napi.node")
    r = subprocess.run([NODE, "-e", js], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip() == "8"
    import torch
    if not torch.cuda.is_available():
        js = ("const {clContext}=require('%s'); const c=new clContext({deviceIndex:0});"
              "c.initialise().then(()=>{console.log('unexpected');process.exit(3)},e=>{console.log(e.message);})") % \
            os.path.join(ROOT, "node", "index.js")
        r = subprocess.run([NODE, "-e", js], capture_output=True, text=True)
        assert "no HIP device" in r.stdout, r.stdout + r.stderr


@needs_node
def test_addon_host_maths_equals_the_reference_goldens():
    """index.js `colour` (the library's ph_colour_* through N-API) against tests/golden/host_maths.json - the
    numbers the reference's colourMaths.ts / transform.ts produced under node."""
    _build_addon()
    hm = json.load(open(os.path.join(ROOT, "tests", "golden", "host_maths.json")))
    js = ("const {colour,planeBytes}=require('%s'); const crypto=require('crypto');"
          "const hex=(a)=>Array.from(new Uint32Array(a.buffer,a.byteOffset,a.length)).map(x=>x.toString(16).padStart(8,'0'));"
          "const sha=(a)=>crypto.createHash('sha256').update(Buffer.from(a.buffer,a.byteOffset,a.byteLength)).digest('hex');"
          "const o={y2r:{},r2y:{},g:{},lut:{},xf:[]};"
          "for (const s of ['601-625','601_525','709','2020','sRGB','bogus']) {"
          " o.y2r[s+'/10']=hex(colour.ycbcr2rgbMatrix(s)); o.y2r[s+'/8']=hex(colour.ycbcr2rgbMatrix(s,8,16,235,224));"
          " o.r2y[s+'/10']=hex(colour.rgb2ycbcrMatrix(s)); o.r2y[s+'/8']=hex(colour.rgb2ycbcrMatrix(s,8,16,235,224));"
          " for (const d of ['601-625','601_525','709','2020','sRGB','bogus']) o.g[s+'->'+d]=hex(colour.rgb2rgbMatrix(s,d));"
          " o.lut[s]=[sha(colour.gamma2linearLUT(s)),sha(colour.linear2gammaLUT(s))]; }"
          "for (const t of %s) o.xf.push(hex(colour.transformMatrix(t.width,t.height,t.params)));"
          "o.planes=planeBytes('yuv420p',1920,1080); console.log(JSON.stringify(o))") % (
              os.path.join(ROOT, "node", "index.js"), json.dumps([dict(width=t["width"], height=t["height"], params=t["params"]) for t in hm["transform"]]))
    r = subprocess.run([NODE, "-e", js], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    o = json.loads(r.stdout)
    assert o["y2r"] == hm["ycbcr2rgb"] and o["r2y"] == hm["rgb2ycbcr"] and o["g"] == hm["rgb2rgb"]
    for s, (g2l, l2g) in o["lut"].items():
        assert g2l == hm["lut"][s]["g2l_sha256"] and l2g == hm["lut"][s]["l2g_sha256"], s
    assert o["xf"] == [t["matrix"] for t in hm["transform"]]
    assert o["planes"] == [1920 * 1080, 960 * 540, 960 * 540]


@needs_node
@pytest.mark.gpu
@pytest.mark.parametrize("deferred", [False, True], ids=["launch_as_posted", "deferred"])
def test_golden_trace_replays_on_the_real_addon(tmp_path, deferred):
    """tests/golden/host_trace.json - every nodencl call the reference's own operators and dispatcher made in the
    scenario - executed call by call on index.js + ph_napi.c + libphaneron_hip.so (node/test/replay.js); a second time
    through the recording context (node/defer.js, PHANERON_DEFERRED=1): same counts, same frames, fewer launches."""
    import hashlib
    import frames
    trace = json.load(open(TRACE))
    # the frames the reference loaded were its own test patterns (fillBuf of each format): regenerate them
    blobs = [frames.v210_ramp(1920, 1080)]
    for fmt in PACK_FORMATS:
        blobs += frames.pack_ramp(fmt, 1920, 1080)
    have = set()
    for b in blobs:
        raw = np.ascontiguousarray(b).view(np.uint8).tobytes()
        name = hashlib.sha256(raw).hexdigest()[:16]
        have.add(name)
        (tmp_path / (name + ".bin")).write_bytes(raw)
    big = {e["src"]["sha"] for e in trace if e["op"] == "hostAccess" and e["src"] and e["src"]["bytes"] > 64}
    assert big <= have, "the trace loads a frame this test cannot regenerate"
    r = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "replay.js"), TRACE, TEXT_SHA, str(tmp_path)],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, PHANERON_DEFERRED="1" if deferred else "0"))
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.loads((tmp_path / "replay.json").read_text())
    assert "gfx950" in res["device"]
    assert res["problems"] == [], res["problems"][:5]
    if deferred:  # the v210 round trip (read -> write, io.ts) is one fused launch (images the scenario never maps stay recipes)
        d = res["deferred"]
        assert d["fused"] >= 1 and d["launched"] < d["recorded"] and d["fallbacks"] == 0, d
    else:
        assert res["deferred"] is None
    n_calls = sum(1 for e in trace if e["op"] in ("createBuffer", "hostAccess", "createProgram", "runProgram", "waitFinish", "addRef", "release"))
    assert res["calls"] == n_calls and n_calls > 480
    # what the reference mapped for reading (saveFrame): the v210 round trip gives the ramp back ("Compare returned
    # 0"), and every other format's pattern -> read(inSpec -> outSpec) -> write(outSpec, both fields) equals the oracle
    from oracle import orc
    dumps = res["dumps"]
    assert len(dumps) == 1 + sum(len(frames.pack_ramp(f, 1920, 1080)) for f in PACK_FORMATS)
    got = [np.fromfile(tmp_path / d["file"], np.uint8) for d in dumps]
    assert np.array_equal(got[0], np.ascontiguousarray(blobs[0]).view(np.uint8))
    colours = {"yuv422p10": ("709", "709"), "yuv422p8": ("601-625", "709"), "yuv420p": ("709", "2020"), "nv12": ("709", "709"),
               "rgba8": ("sRGB", "709"), "bgra8": ("sRGB", "sRGB")}  # node/test/scenario.js fmtColours
    k = 1
    for fmt in PACK_FORMATS:
        spec, ospec = colours[fmt]
        rng = orc.FORMAT_RANGE[fmt]
        planes = [np.ascontiguousarray(p).view(np.uint8) for p in frames.pack_ramp(fmt, 1920, 1080)]
        rgba = orc.pack_read(fmt, planes, 1920, 1080, None if rng is None else orc.ycbcr2rgb_matrix(spec, *rng),
                             orc.gamma2linear_lut(spec), orc.rgb2rgb_matrix(spec, ospec))
        wcm = None if rng is None else orc.rgb2ycbcr_matrix(ospec, *rng)
        want = orc.pack_write(fmt, rgba, 1920, 1080, 1, wcm, orc.linear2gamma_lut(ospec))
        want = orc.pack_write(fmt, rgba, 1920, 1080, 3, wcm, orc.linear2gamma_lut(ospec), planes=want)
        for i, wnt in enumerate(want):
            assert np.array_equal(got[k + i], wnt), "%s plane %d" % (fmt, i)
        if spec == ospec:  # same colourspace in and out: the reference scripts' round trip
            assert all(np.array_equal(g, p) for g, p in zip(got[k:k + len(planes)], planes)), fmt
        k += len(planes)


# the round trips gen_golden.py recorded from the reference kernels (kat.json "<fmt>_<w>x<h>_*")
FORMAT_KATS = [("yuv422p10", 1920, 1080, "709"), ("yuv420p", 1920, 1080, "709"), ("nv12", 1920, 1080, "709"),
               ("yuv422p8", 718, 480, "709"), ("rgba8", 1920, 1080, "sRGB"), ("bgra8", 1920, 1080, "sRGB")]


@needs_node
@pytest.mark.gpu
@pytest.mark.parametrize("deferred", [False, True], ids=["launch_as_posted", "deferred"])
def test_node_layer_end_to_end_on_gpu(tmp_path, deferred):
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
    frames.v210_ramp(1920, 1080).tofile(tmp_path / "ramp.bin")
    for f, fw, fh, _ in FORMAT_KATS:  # each format's reference test pattern (its fillBuf), all planes in one file
        np.concatenate([np.ascontiguousarray(p).view(np.uint8) for p in frames.pack_ramp(f, fw, fh)]).tofile(tmp_path / ("pattern_%s.bin" % f))
    job = dict(channel=dict(width=w, height=h, layers=["layer%d.bin" % i for i in range(n)], readSpec="709",
                            writeSpec="2020", pip=pip),
               ramp="ramp.bin",
               yadif=dict(width=yw, height=yh, frames=["field%d.bin" % i for i in range(4)], tff=True),
               formats=[dict(fmt=f, width=fw, height=fh, spec=sp, file="pattern_%s.bin" % f) for f, fw, fh, sp in FORMAT_KATS],
               staged=dict(width=1920, height=24, layers=3, frames=5, readSpec="709", writeSpec="2020"),
               deint=dict(width=384, height=22, tff=True, readSpec="709", writeSpec="2020",
                          layers=[["deint_l%d_f%d.bin" % (l, i) for i in range(3)] for l in range(2)]))
    deint_src = [[frames.v210_random(384, 22, 9300 + 10 * l + i) for i in range(3)] for l in range(2)]
    for l, win in enumerate(deint_src):
        for i, words in enumerate(win):
            words.tofile(tmp_path / ("deint_l%d_f%d.bin" % (l, i)))
    staged_src = [[frames.v210_random(1920, 24, frames.layer_seed(8 + f, l)) for l in range(3)] for f in range(5)]
    for f, ls in enumerate(staged_src):
        for l, words in enumerate(ls):
            words.tofile(tmp_path / ("staged_f%d_l%d.bin" % (f, l)))
    (tmp_path / "job.json").write_text(json.dumps(job))
    # (deferred: the whole script on a recording context - same frames, same hashes, the same errors where they are raised)
    r = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "gpu_run.js"), str(tmp_path)], capture_output=True,
                       text=True, timeout=300, env=dict(os.environ, PHANERON_DEFERRED="1" if deferred else "0"))
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

    # the same channel with its tail as one launch (program 'compose_write_v210_<n>'), and with a wipe inside
    assert np.array_equal(np.fromfile(tmp_path / "compose_out.bin", np.uint32), want)
    top = orc.transition_wipe(rgba[n - 1], orc.v210_read(layers[0], w, h, *rd), orc.v210_read(layers[2], w, h, *rd))
    want_wipe = orc.v210_write(orc.combine(rgba[:n - 1] + [top]), w, h, 0, orc.rgb2ycbcr_matrix("2020"), orc.linear2gamma_lut("2020"))
    assert np.array_equal(np.fromfile(tmp_path / "compose_wipe_out.bin", np.uint32), want_wipe)
    # and as ONE launch straight from the v210 sources (program 'chan_compose_v210_<n>': no f32 frame at all)
    assert np.array_equal(np.fromfile(tmp_path / "chan_out.bin", np.uint32), want)
    assert np.array_equal(np.fromfile(tmp_path / "chan_wipe_out.bin", np.uint32), want_wipe)

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

    # the de-interlacing fusions under their program names: both fields in one pass, and ToRGBA + both fields
    for parity in (0, 1):
        want = orc.yadif(fields[0], fields[1], fields[2], parity, True, False)
        got = np.fromfile(tmp_path / ("yadif_pair_p%d.bin" % parity), np.float32)
        assert np.array_equal(got.view(np.uint32), want.reshape(-1).view(np.uint32)), parity
    for l, win in enumerate(deint_src):
        p, c, nx = (orc.v210_read(f, 384, 22, *rd) for f in win)
        for parity in (0, 1):
            got = np.fromfile(tmp_path / ("deint_l%d_p%d.bin" % (l, parity)), np.float32)
            assert np.array_equal(got.view(np.uint32), orc.yadif(p, c, nx, parity, True, False).reshape(-1).view(np.uint32)), (l, parity)

    for i, f in enumerate(deint_src[0]):   # the batched reader under its program name
        got = np.fromfile(tmp_path / ("batch_read_%d.bin" % i), np.float32)
        assert np.array_equal(got.view(np.uint32), orc.v210_read(f, 384, 22, *rd).reshape(-1).view(np.uint32)), i

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
    # failures reach JS as Errors carrying the library's message
    e = res["errors"]
    assert e["unknownKey"] == "Failed to run queue for id nobody ts 5"
    assert "unknown kernel 'sharpen'" in e["unknownKernel"]
    assert "kernel argument 'input' (buffer) missing" in e["missingArgument"]
    assert "at least 2 layers" in e["combineOne"]
    assert "unknown option 'stream_everything'" in e["unknownOption"] and e["knownOption"] is None
    # five flushes requested together: served in order, one drain (jobs.js)
    assert res["boardStats"] == {"flushes": 5, "drains": 1, "kernels": 7}
    assert res["liveAfter"] == 0
    assert res["routeLoopbackCompare"] == 0     # a frame through ph_route (RCCL, own rank) and back to v210: the ramp again


def mixer_matrix(w, h, p):
    """Mixer.mixVidValve's parameter mapping (mixer.ts:209-223) into the Transform matrix"""
    return orc_mod().transform_matrix(w, h, False, False, p["anchor"]["x"] - 0.5, p["anchor"]["y"] - 0.5, p["fill"]["xScale"],
                                      p["fill"]["yScale"], -p["fill"]["xOffset"], -p["fill"]["yOffset"], -p["rotation"] / 360.0)


def orc_mod():
    from oracle import orc
    return orc


@needs_node
@pytest.mark.gpu
@pytest.mark.parametrize("deferred", [False, True], ids=["launch_as_posted", "deferred"])
def test_channel_compositor_on_gpu(tmp_path, deferred):
    """node/channel.js on the real addon - a full-frame layer, a PiP layer that dissolves into another clip, an
    empty layer: every output frame equals the oracle's chain for that frame under the reference's valve rules
    (mixer.ts:209-223, transitioner.ts:143-176, combiner.ts:211-254), and no buffer leaks."""
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
    r = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "channel_run.js"), str(tmp_path)], capture_output=True,
                       text=True, timeout=300, env=dict(os.environ, PHANERON_DEFERRED="1" if deferred else "0"))
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.loads((tmp_path / "result.json").read_text())
    assert res["stamps"] == list(range(nf))     # the combiner renumbers its output 0, 1, 2 ...
    assert res["leaked"] == 0                   # every intermediate went back to the pool
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


def amcp_noise(nbytes, seed, k):
    """node/amcp.js noiseFrame, restated: a 32-bit LCG stream, three 10-bit legal-range fields per word"""
    def imul(a, b):
        return (a * b) & 0xFFFFFFFF
    s = (imul(seed, 0x9E3779B1) ^ imul(k + 1, 0x85EBCA6B)) & 0xFFFFFFFF
    out = np.empty(nbytes // 4, np.uint32)
    for i in range(out.size):
        w = 0
        for f in range(3):
            s = (imul(s, 1664525) + 1013904223) & 0xFFFFFFFF
            w |= (64 + ((s >> 8) % 877)) << (10 * f)
        out[i] = w
    return out


@needs_node
@pytest.mark.gpu
@pytest.mark.parametrize("deferred", [False, True], ids=["launch_as_posted", "deferred"])
def test_control_plane_smoke_on_gpu(tmp_path, deferred):
    """AMCP command lines (PLAY / LOADBG ... MIX / MIXER FILL / ROTATION / STOP / CLEAR) drive a channel of
    synthetic v210 sources on the real addon; every output frame equals the oracle chain read -> place -> dissolve ->
    combine -> write for the state the commands left, responses have the server's shape, nothing leaks."""
    import frames
    orc = orc_mod()
    w, h = 192, 16
    script = ["PLAY 1-10 NOISE:1", "PLAY 1-20 NOISE:2", "MIXER 1-20 FILL 0.25 0.25 0.5 0.5", dict(tick=2),
              "LOADBG 1-20 NOISE:3 MIX 4", "PLAY 1-20", dict(tick=5), "MIXER 1-10 ROTATION 15", dict(tick=1),
              "STOP 1-10", dict(tick=1), "CLEAR 1", dict(tick=1),
              "SWAP 1-10 1-20", "PLAY 9-1 NOISE:1", "PLAY 1-30 NOFILE", "MIXER 1-10 FILL 1 2"]
    (tmp_path / "job.json").write_text(json.dumps(dict(width=w, height=h, readSpec="709", writeSpec="2020", script=script)))
    # (deferred: the same script on a recording context - every frame it packs must still be the oracle's)
    r = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "amcp_run.js"), str(tmp_path)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PHANERON_DEFERRED="1" if deferred else "0"))
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.loads((tmp_path / "result.json").read_text())
    assert res["responses"] == ["202 PLAY OK", "202 PLAY OK", "202 MIXER OK", "202 LOADBG OK", "202 PLAY OK", "202 MIXER OK",
                                "202 STOP OK", "202 CLEAR OK", "400 ERROR\r\nSWAP 1-10 1-20 NOT IMPLEMENTED", "404 PLAY ERROR",
                                "404 PLAY ERROR", "400 ERROR\r\nMIXER 1-10 FILL 1 2 NOT IMPLEMENTED"]
    assert res["frames"] == 10 and res["leaked"] == 0

    nbytes = frames.v210_pitch_bytes(w) * h
    rd = (orc.ycbcr2rgb_matrix("709"), orc.gamma2linear_lut("709"), orc.rgb2rgb_matrix("709", "2020"))
    wr = (orc.rgb2ycbcr_matrix("2020"), orc.linear2gamma_lut("2020"))
    src = lambda seed, k: orc.v210_read(amcp_noise(nbytes, seed, k), w, h, *rd)
    place = lambda anchor=(0, 0), rot=0, fill=(0, 0, 1, 1): mixer_matrix(w, h, dict(anchor=dict(x=anchor[0], y=anchor[1]), rotation=rot,
                                                                             fill=dict(xOffset=fill[0], yOffset=fill[1], xScale=fill[2], yScale=fill[3])))
    full, pip, rot = place(), place(fill=(0.25, 0.25, 0.5, 0.5)), place(rot=15)
    black = np.zeros((h, w, 4), np.float32)
    for f in range(10):
        if f < 2:
            img = orc.combine([orc.transform(src(1, f), full, w, h), orc.transform(src(2, f), pip, w, h)])
        elif f < 6:   # NOISE:2 keeps playing under a 4-frame dissolve into NOISE:3 (transitioner.ts:170)
            k = f - 2
            top = orc.transition_dissolve(orc.transform(src(2, f), pip, w, h), orc.transform(src(3, k), pip, w, h), float(np.float32(1.0 - k / 3)))
            img = orc.combine([orc.transform(src(1, f), full, w, h), top])
        elif f == 6:
            img = orc.combine([orc.transform(src(1, f), full, w, h), orc.transform(src(3, f - 2), pip, w, h)])
        elif f == 7:
            img = orc.combine([orc.transform(src(1, f), rot, w, h), orc.transform(src(3, f - 2), pip, w, h)])
        elif f == 8:  # layer 10 stopped: it contributes transparent black (combiner.ts:211-254)
            img = orc.combine([black, orc.transform(src(3, f - 2), pip, w, h)])
        else:         # cleared: black
            img = black
        want = orc.v210_write(img, w, h, 0, *wr)
        got = np.fromfile(tmp_path / ("out_%d.bin" % f), np.uint32)
        assert np.array_equal(got, want), f


# ---------------------------------------------------------------------------------------------------------------------
# the reference's OWN video valves (mixer.ts, transitioner.ts, combiner.ts, blackSilence.ts), type-erased and run
# against stand-ins for redioactive / beamcoder in the build container (node/test/valve_scenario.js); their trace is the
# golden, and it is replayed call by call on the real addon
# ---------------------------------------------------------------------------------------------------------------------
VALVE_TRACE = os.path.join(ROOT, "tests", "golden", "valve_trace.json")
VALVE_CLIPS = {"A": 11, "B0": 22, "B1": 33, "C": 44, "M": 55}  # node/test/valve_scenario.js CLIPS
VALVE_PIP = dict(anchor=dict(x=0.25, y=0.75), rotation=30, fill=dict(xOffset=0.25, yOffset=-0.125, xScale=0.5, yScale=0.5), volume=1)
VALVE_DEFAULT = dict(anchor=dict(x=0, y=0), rotation=0, fill=dict(xOffset=0, yOffset=0, xScale=1, yScale=1), volume=1)


def valve_clip_frame(w, h, seed, k):
    """node/test/valve_scenario.js clipFrame, restated: f32 RGBA noise in [0, 1) from a 32-bit hash of (element index, seed)"""
    def imul(a, b):
        return (np.asarray(a, np.uint64) * np.uint64(b)) & np.uint64(0xFFFFFFFF)
    s = int(imul(seed, 0x9E3779B1)) ^ int(imul(k + 1, 0x85EBCA6B))
    i = np.arange(1, w * h * 4 + 1, dtype=np.uint64)
    x = imul(i ^ np.uint64(s), 0x9E3779B1)
    x = x ^ (x >> np.uint64(15))
    x = imul(x, 0x85EBCA6B)
    x = x ^ (x >> np.uint64(13))
    return ((x >> np.uint64(8)).astype(np.float64) / 16777216.0).astype(np.float32).reshape(h, w, 4)


@needs_node
@needs_ref_js
def test_reference_valves_still_leave_the_golden_trace():
    """Where the reference checkout exists: its Mixer / Transitioner / Combiner / Black valves, driven through the 11-frame
    schedule, make exactly the nodencl calls of tests/golden/valve_trace.json"""
    out = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "valve_scenario.js"), REF_JS], check=True, capture_output=True, text=True).stdout
    trace = json.loads(out)
    assert trace == json.load(open(VALVE_TRACE))
    runs = [e for e in trace if e["op"] == "runProgram"]
    assert [e["name"] for e in runs].count("transition_dissolve") == 4 and [e["name"] for e in runs].count("transition_wipe") == 4
    f32 = lambda x: float(np.float32(x))
    assert [e["params"]["mix"] for e in runs if e["name"] == "transition_dissolve"] == [1.0, f32(2 / 3), f32(1 / 3), 0.0]  # transitioner.ts:170
    assert [e["timestamp"] for e in trace if e["op"] == "note" and e["what"] == "output"] == list(range(11))  # combiner.ts:211


def valve_clip_frame_v210(w, h, seed, k):
    """node/test/valve_scenario.js clipFrameV210, restated: word i packs three legal-range codes from the hash of elements 3i .. 3i + 2"""
    def imul(a, b):
        return (np.asarray(a, np.uint64) * np.uint64(b)) & np.uint64(0xFFFFFFFF)
    words = (w + 47) // 48 * 128 * h // 4
    s = int(imul(seed, 0x9E3779B1)) ^ int(imul(k + 1, 0x85EBCA6B))
    i = np.arange(1, 3 * words + 1, dtype=np.uint64)
    x = imul(i ^ np.uint64(s), 0x9E3779B1)
    x = x ^ (x >> np.uint64(15))
    x = imul(x, 0x85EBCA6B)
    x = x ^ (x >> np.uint64(13))
    code = (np.uint64(64) + (x >> np.uint64(8)) % np.uint64(877)).reshape(words, 3)
    return (code[:, 0] | (code[:, 1] << np.uint64(10)) | (code[:, 2] << np.uint64(20))).astype(np.uint32)


def valve_oracle_frames(w, h, v210=False):
    """what the 11 output frames must be: the schedule of node/test/valve_scenario.js restated on the oracle's kernels
    (v210: the clips are v210 frames read 709 -> 709, the outputs packed v210 frames)"""
    orc = orc_mod()
    md, mp = mixer_matrix(w, h, VALVE_DEFAULT), mixer_matrix(w, h, VALVE_PIP)
    used = {c: 0 for c in VALVE_CLIPS}
    rd = (orc.ycbcr2rgb_matrix("709"), orc.gamma2linear_lut("709"), orc.rgb2rgb_matrix("709", "709"))
    wr = (orc.rgb2ycbcr_matrix("709"), orc.linear2gamma_lut("709"))

    def mixed(clip, m):
        k = used[clip]
        used[clip] += 1
        if v210:
            return orc.transform(orc.v210_read(valve_clip_frame_v210(w, h, VALVE_CLIPS[clip], k), w, h, *rd), m, w, h)
        return orc.transform(valve_clip_frame(w, h, VALVE_CLIPS[clip], k), m, w, h)
    black = np.zeros((h, w, 4), np.float32)
    out = []
    for f in range(11):
        l1 = mixed("A", md)
        if f >= 7:  # transitioner.ts:165-176: inputs = srcFrames.slice(0, 2), mask = srcFrames[2]
            l1 = orc.transition_wipe(l1, mixed("C", md), mixed("M", md))
        if f < 2:
            l2 = mixed("B0", mp)
        elif f < 6:
            l2 = orc.transition_dissolve(mixed("B0", mp), mixed("B1", md), float(np.float32(1.0 - (f - 2) / 3)))
        else:
            l2 = mixed("B1", md)
        out.append(orc.combine([l1, l2, black]))
    if v210:
        return [np.asarray(orc.v210_write(o, w, h, 0, *wr)).reshape(-1) for o in out]
    return out


@needs_node
@pytest.mark.gpu
@pytest.mark.parametrize("deferred", [False, True], ids=["launch_as_posted", "deferred"])
def test_reference_valve_trace_replays_on_the_real_addon(tmp_path, deferred):
    """Every nodencl call the reference's valves made - buffers by owner, Transform / Transition / Combine programs from their
    kernel names, parameters by OpenCL argument name, the black frame written through a mapped mirror, every addRef / release -
    goes to index.js + ph_napi.c + libphaneron_hip.so; reference counts match the recorded ones at every step, nothing leaks,
    and all 11 output frames equal the oracle's chain for the schedule."""
    import hashlib
    trace = json.load(open(VALVE_TRACE))
    w, h = 192, 64
    have = set()
    for clip, seed in VALVE_CLIPS.items():
        for k in range(11):
            raw = valve_clip_frame(w, h, seed, k).tobytes()
            name = hashlib.sha256(raw).hexdigest()[:16]
            have.add(name)
            (tmp_path / (name + ".bin")).write_bytes(raw)
    big = {e["src"]["sha"] for e in trace if e["op"] == "hostAccess" and e["src"] and e["src"]["bytes"] > 64}
    assert big <= have, "the trace loads a frame this test cannot regenerate"
    r = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "replay.js"), VALVE_TRACE, TEXT_SHA, str(tmp_path)],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, PHANERON_DEFERRED="1" if deferred else "0"))
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.loads((tmp_path / "replay.json").read_text())
    assert res["problems"] == [], res["problems"][:5]
    if deferred:  # (the valves' outputs are f32 images mapped by the test, so every job runs as recorded, when its image is asked for)
        assert res["deferred"]["recorded"] > 0 and res["deferred"]["fallbacks"] == 0, res["deferred"]
    assert len(res["dumps"]) == 11
    want = valve_oracle_frames(w, h)
    for f, d in enumerate(res["dumps"]):
        got = np.fromfile(tmp_path / d["file"], np.float32)
        assert np.array_equal(got.view(np.uint32), want[f].reshape(-1).view(np.uint32)), "output frame %d" % f


@needs_node
@pytest.mark.gpu
def test_channel_js_equals_the_reference_valves_frame_for_frame(tmp_path):
    """this repository's own channel compositor (node/channel.js) on the first seven frames of the same schedule (full-frame
    layer, PiP layer dissolving into another clip, empty layer): the frames the reference's valves produce"""
    w, h, nf = 192, 64, 7
    for name, seed in (("A", 11), ("B0", 22), ("B1", 33)):
        for k in range(nf):
            valve_clip_frame(w, h, seed, k).tofile(tmp_path / ("%s_%d.bin" % (name, k)))
    (tmp_path / "job.json").write_text(json.dumps(dict(width=w, height=h, frames=nf, pip=VALVE_PIP, dissolveAt=2, dissolveLen=4, cutAt=6)))
    r = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "channel_run.js"), str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    want = valve_oracle_frames(w, h)  # == the replayed reference valves (test above)
    for f in range(nf):
        got = np.fromfile(tmp_path / ("out_%d.bin" % f), np.float32)
        assert np.array_equal(got.view(np.uint32), want[f].reshape(-1).view(np.uint32)), "frame %d" % f


@needs_node
def test_recording_context_is_the_default():
    """`new clContext()` - what src/index.ts:94-107 writes - records; `deferred: false` or PHANERON_DEFERRED=0 gives launch-as-posted"""
    js = ("const { clContext } = require('%s'); delete process.env.PHANERON_DEFERRED;"
          "const out = [new clContext().deferred, new clContext({ deferred: false }).deferred];"
          "process.env.PHANERON_DEFERRED = '0'; out.push(new clContext().deferred, new clContext({ deferred: true }).deferred);"
          "process.env.PHANERON_DEFERRED = '1'; out.push(new clContext().deferred);"
          "console.log(JSON.stringify(out))") % os.path.join(ROOT, "node", "index.js")
    r = subprocess.run([NODE, "-e", js], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout) == [True, False, False, True, True]


@needs_node
@pytest.mark.gpu
def test_recording_context_gives_the_same_frames_with_one_launch_per_frame():
    """node/defer.js (new clContext({deferred: true})): operator-by-operator job streams shaped like the valves' - fresh
    destinations released in the job callbacks, several frames in flight, sources and placement matrices overwritten while
    recorded, field writes, a de-interlaced layer, intermediates asked for afterwards - give byte for byte the frames of the
    launch-as-posted context, the chains with a fused form as ONE launch per frame, and leave no buffer behind."""
    _build_addon()
    for size in (("384", "108"), ("1920", "64"), ("1920", "1080")):  # (the last: every scenario at the reference's own frame size)
        r = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "defer_run.js"), *size], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        res = json.loads(r.stdout.strip().splitlines()[-1])
        assert res["problems"] == [], res["problems"][:3]
        assert len(res["scenarios"]) >= 15 and all(s["frames"] >= 1 for s in res["scenarios"])
        first = res["scenarios"][0]["deferred"]
        # (three frames recorded, asked for together: one call, ph_fused_v210_combine_batch)
        assert first["recorded"] == 18 and first["launched"] == 1 and first["fused"] == 3 and first["batched"] == 3, first


# ---------------------------------------------------------------------------------------------------------------------
# The whole per-frame video path with the reference's code at BOTH ends: ToRGBA (one Loader per producer) -> Mixer ->
# Transitioner -> Combiner -> FromRGBA -> saveFrame (node/test/valve_scenario.js --v210), v210 in, v210 out
# ---------------------------------------------------------------------------------------------------------------------
CHANNEL_TRACE = os.path.join(ROOT, "tests", "golden", "channel_trace.json")


@needs_node
@needs_ref_js
def test_reference_channel_path_still_leaves_the_golden_trace():
    out = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "valve_scenario.js"), REF_JS, "--v210"], check=True, capture_output=True, text=True).stdout
    trace = json.loads(out)
    assert trace == json.load(open(CHANNEL_TRACE))
    names = [e["name"] for e in trace if e["op"] == "runProgram"]
    assert names.count("read") == 34 and names.count("transform") == 34 and names.count("write") == 11 and names.count("combine_3") == 11


@needs_node
@pytest.mark.gpu
@pytest.mark.parametrize("deferred", [False, True], ids=["launch_as_posted", "deferred"])
def test_reference_channel_trace_replays_on_the_real_addon(tmp_path, deferred):
    """The nodencl calls of the reference's complete video path (98 kernel jobs for 11 frames) on the real addon: reference
    counts as recorded, nothing leaks, every packed output frame equals the oracle's chain.  Through the recording context
    (PHANERON_DEFERRED=1) the same calls give the same frames with ONE launch per output frame: the Loaders of the five
    producers are different buffers with the same contents, the third layer is the Transitioner's black frame."""
    import hashlib
    trace = json.load(open(CHANNEL_TRACE))
    w, h = 192, 64
    have = set()
    for clip, seed in VALVE_CLIPS.items():
        for k in range(11):
            raw = valve_clip_frame_v210(w, h, seed, k).tobytes()
            name = hashlib.sha256(raw).hexdigest()[:16]
            have.add(name)
            (tmp_path / (name + ".bin")).write_bytes(raw)
    big = {e["src"]["sha"] for e in trace if e["op"] == "hostAccess" and e["src"] and e["src"]["bytes"] > 64}
    assert big <= have, "the trace loads a frame this test cannot regenerate"
    r = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "replay.js"), CHANNEL_TRACE, TEXT_SHA, str(tmp_path)],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, PHANERON_DEFERRED="1" if deferred else "0"))
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.loads((tmp_path / "replay.json").read_text())
    assert res["problems"] == [], res["problems"][:5]
    assert len(res["dumps"]) == 11
    want = valve_oracle_frames(w, h, v210=True)
    for f, d in enumerate(res["dumps"]):
        got = np.fromfile(tmp_path / d["file"], np.uint32)
        assert np.array_equal(got, want[f]), "output frame %d" % f
    if deferred:
        d = res["deferred"]
        assert d["recorded"] == 98 and d["fused"] == 11 and d["launched"] == 11 and d["plain"] == 0 and d["fallbacks"] == 0 and d["pending"] == 0, d


@needs_node
def test_recording_graph_logic_without_a_device():
    """node/defer.js against a counting stand-in for the addon (node/test/defer_check.js): what is launched and in which order,
    what is dropped, who keeps which buffer alive, hazards (a source overwritten while recorded, a destination reused, field
    writes, in-place jobs), Loaders matched by content, refused fused launches falling back to the recorded jobs"""
    r = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "defer_check.js")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.loads(r.stdout)
    assert res["problems"] == [] and res["checks"] >= 34, res["problems"]


@needs_node
@pytest.mark.gpu
def test_random_job_streams_give_the_same_bytes_through_the_recording_context():
    """node/test/defer_fuzz.js: seeded random streams of reads, transforms, transitions, combines, de-interlaces, frame and
    field writes, source and matrix overwrites, early releases and host reads in the middle of chains - nothing shaped like a
    channel - through the plain context and through the recording one: every host read sees the same bytes, nothing is left
    alive or recorded at the end, and fused launches did take part"""
    _build_addon()
    r = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "defer_fuzz.js"), "7000", "80", "120"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["problems"] == [], res["problems"][:2]
    assert res["fusedLaunches"] > 100 and res["launchesSaved"] > 1000, res
    # the same at a width that is not a multiple of 48 (a tail quad and cleared slots in every v210 line; 4:2:0 frames among the outputs)
    r = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "defer_fuzz.js"), "7500", "40", "120"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PHANERON_FUZZ_SIZE="176x12"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["problems"] == [], res["problems"][:2]
    assert res["fusedLaunches"] > 50, res


@needs_node
@pytest.mark.gpu
def test_random_multi_channel_ticks_equal_the_launch_as_posted_context():
    """node/test/channels_fuzz.js: per tick 1 - 6 channels post a frame each - plain reads, clips under the default fill (v210 and decoders'
    planar frames), clips smaller than the channel, picture-in-picture, graphics with alpha - and the tick goes to the device as one
    runPrograms call: batch kernel, headline batch, read + compositor route (alone and grouped), jobs in their turn, in every order the
    seeds produce.  Every consumer's frame equals the launch-as-posted context's; nothing refused, nothing left.  Progressive shapes only:
    every frame folds (no job runs as recorded).  Then with 1080i channels among them (1 - 3 Yadif windows, both fields' frames: the
    de-interlacing reader + the 2 x 2-block compositor where the fields agree, the recorded jobs where they do not), in both orders
    node/defer.js knows for them (channel by channel; PHANERON_FIELD_BATCH=1: the channels' windows and frames in shared launches)"""
    _build_addon()
    fuzz = [NODE, os.path.join(ROOT, "node", "test", "channels_fuzz.js")]
    r = subprocess.run(fuzz + ["1", "24", "10"], capture_output=True, text=True, timeout=900, env=dict(os.environ, PH_FUZZ_NO_DEINT="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["problems"] == [], res["problems"][:4]
    assert res["deferred"]["fallbacks"] == 0 and res["deferred"]["plain"] == 0 and res["deferred"]["batched"] > res["deferred"]["launched"]
    for extra in ({"PH_FUZZ_ROUTES": "1"}, {"PH_FUZZ_ROUTES": "1", "PHANERON_FIELD_BATCH": "1"}):
        r = subprocess.run(fuzz + ["101", "12", "10"], capture_output=True, text=True, timeout=900, env=dict(os.environ, **extra))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        res = json.loads(r.stdout.strip().splitlines()[-1])
        assert res["problems"] == [] and res["deferred"]["fallbacks"] == 0, res
        assert res["routes"].get("v210_yadif_pair", 0) > 10 and res["routes"].get("compose_up_write_v210", 0) > 10, res["routes"]
    # the reference's own frame size (copies and kernels long enough for an ordering slip between the queues to show)
    r = subprocess.run(fuzz + ["9000", "3", "5", "1920", "1080"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["problems"] == [] and res["deferred"]["fallbacks"] == 0, res


@needs_node
@pytest.mark.gpu
def test_recording_context_soak_fault_and_timings():
    """The default (recording) context over 10^5 frames with a format change every 25 000 (1080 -> 720 -> 2160 -> 1080): buffer and
    pinned-memory counters flat, nothing pinned in steady state, no fused launch refused; launches made to fail while one frame's
    consumer maps it (context option fail_launches): that consumer is told, every job callback fires, the frames after it are the
    launch-as-posted context's bytes, nothing leaks; and with `profile` the terminal write of a frame returns the fused launch's
    device time shared out over the frame's jobs - every row of what src/clJobQueue.ts:159-215 prints non-zero, the rows summing to the launch; four
    channels posting a frame per tick for 20 000 ticks with a format change: one call per tick, every frame through a batch launch, counters flat."""
    _build_addon()
    r = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "soak_run.js"), "100000"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["problems"] == [], res["problems"][:4]
    assert res["soak"]["frames"] == 100000 and res["soak"]["deferred"]["fallbacks"] == 0
    assert res["channels"]["ticks"] == 20000 and res["channels"]["deferred"]["batched"] == 80000 and res["channels"]["deferred"]["launched"] == 20000
    assert "injected" in res["fault"]["deferred"]["rejected"] and res["fault"]["plain"]["rejected"] is None
    assert all(t["write"]["kernelExec"] > 0 and all(x > 0 for x in t["folded"]) and 5 < t["sum"] < 5000 for t in res["timings"])


@needs_node
@pytest.mark.gpu
def test_parked_buffers_are_settled_before_their_next_owner():
    """node/index.js parks released frames whole (handle, device block, pinned mirror).  The next owner of one whose mirror or block may
    still be busy - released right after downloadAsync, mapped for writing and never handed back - gets it settled by the library
    (ph_buf_reuse): it reads back what IT wrote, the previous owner's abandoned host data never reaches the device, and a buffer that
    was merely created and released still costs no call.  With `strictHandles` the next owner gets another JS object over
    the parked block and the reference its previous owner kept is refused, as a released nodencl buffer is."""
    _build_addon()
    r = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "park_run.js")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["problems"] == [] and res["checks"] >= 25, res


@needs_node
@pytest.mark.gpu
def test_recorded_job_streams_reach_the_device_as_the_pinned_kernels():
    """which kernels the recording context's frames become (node/defer.js folds, the library routes): the headline chain, config 2's
    shape, file playback, a 1080i source's field pair, four 1080i sources with picture-in-picture placements, four channels in a tick - recorded under a dry trace and pinned
    (node/test/routes_run.js; the C-side pins are tests/test_routes_gpu.py)"""
    _build_addon()
    r = subprocess.run([NODE, os.path.join(ROOT, "node", "test", "routes_run.js")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["problems"] == [] and res["checks"] == 6, res
