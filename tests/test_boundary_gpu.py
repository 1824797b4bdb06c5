"""GPU tests (-m gpu) of the nodencl-shaped C-ABI surface itself, driven from Python exactly as the
N-API addon drives it: pooled ref-counted buffers with pinned mirrors, hostAccess in its three
directions, createProgram by kernel name / source, runProgram with arguments keyed by OpenCL argument
name, queue ordering primitives and the staged producer -> GPU -> consumer ring.
Results are checked against the oracle, bit for bit."""
import os

import numpy as np
import pytest

import frames
from oracle import orc
from phaneron_amd import capi, staging

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def upload(ctx, arr, access="readonly", svm="none", owner="test", dims=None):
    a = np.ascontiguousarray(arr)
    b = ctx.create_buffer(a.nbytes, access, svm, dims=dims, owner=owner)
    b.host_access("writeonly", capi.QUEUE_LOAD, a)
    return b


class Colour:
    """Loader / Saver parameter buffers as loadSave.ts:50-99,139-160 creates them."""

    def __init__(self, ctx, spec, out_spec):
        self.rd_cm = upload(ctx, capi.ycbcr2rgb_matrix(spec))
        self.rd_lut = upload(ctx, capi.gamma2linear_lut(spec), svm="coarse")
        self.rd_gm = upload(ctx, capi.rgb2rgb_matrix(spec, out_spec))          # 36 bytes, like the reference
        self.wr_cm = upload(ctx, capi.rgb2ycbcr_matrix(out_spec))
        self.wr_lut = upload(ctx, capi.linear2gamma_lut(out_spec), svm="coarse")
        ctx.wait(capi.QUEUE_LOAD)
        self.oracle_rd = (orc.ycbcr2rgb_matrix(spec), orc.gamma2linear_lut(spec), orc.rgb2rgb_matrix(spec, out_spec))
        self.oracle_wr = (orc.rgb2ycbcr_matrix(out_spec), orc.linear2gamma_lut(out_spec))

    def release(self):
        for b in (self.rd_cm, self.rd_lut, self.rd_gm, self.wr_cm, self.wr_lut):
            b.release()


def test_buffers_are_pooled_and_refcounted(ctx):
    base = ctx.buffer_stats()
    a = ctx.create_buffer(1 << 20, owner="a")
    assert a.refcount() == 1
    assert ctx.buffer_stats()["live_buffers"] == base["live_buffers"] + 1
    ptr = a.device_ptr()
    a.add_ref()
    assert a.refcount() == 2
    a.release()
    a.release()                                   # last release: storage goes back to the pool
    after = ctx.buffer_stats()
    assert after["live_buffers"] == base["live_buffers"]
    assert after["pooled_bytes"] >= base["pooled_bytes"] + (1 << 20)
    b = ctx.create_buffer(1 << 20, owner="b")     # same size: the pooled block is handed out again
    assert b.device_ptr() == ptr
    b.release()


def test_pinned_mirrors_are_recycled_behind_their_copies_and_evicted_oldest_first(ctx):
    """A buffer released right after downloadAsync (before its waitFinish) hands its pinned mirror to the pool with the copy
    possibly still in flight: the next buffer of that size must not see the stale copy land in what it uploads.  And a pool that
    is over its budget lets the OLDEST blocks go instead of refusing the newcomer (a format change does not pin dead sizes for ever)."""
    n = 64 << 20
    old = np.full(n, 0xA5, np.uint8)
    new = (np.arange(n, dtype=np.uint32) * 2654435761 >> 24).astype(np.uint8)
    for _ in range(4):
        a = ctx.create_buffer(n, owner="old")
        a.host_access("writeonly", capi.QUEUE_LOAD, old)
        ctx.wait(capi.QUEUE_LOAD)
        a.download_async(capi.QUEUE_UNLOAD)   # 64 MiB device -> mirror, asynchronous
        a.release()                           # ... and the mirror is pooled while that copy may still be running
        b = ctx.create_buffer(n, owner="new")
        b.host_access("writeonly", capi.QUEUE_LOAD, new)  # fills the recycled mirror, then uploads from it
        ctx.wait(capi.QUEUE_LOAD)
        ctx.wait(capi.QUEUE_UNLOAD)
        b.host()[:] = 0
        b.host_access("readonly", capi.QUEUE_UNLOAD)
        assert np.array_equal(b.host(), new)
        b.release()
    # eviction: a budget of 8 MiB, blocks of 3 MiB in three sizes - the pool keeps serving, and a block beyond the budget is just freed
    ctx.set_option("host_pool_mb", 8)
    try:
        for size in (3 << 20, (3 << 20) + 4096, (3 << 20) + 8192, 3 << 20, 16 << 20, (3 << 20) + 4096):
            c = ctx.create_buffer(size)
            data = (frames.splitmix64(size, 1024) & np.uint64(0xFF)).astype(np.uint8)
            c.host_access("writeonly", capi.QUEUE_LOAD, data)
            ctx.wait(capi.QUEUE_LOAD)
            c.host()[:1024] = 0
            c.host_access("readonly", capi.QUEUE_UNLOAD)
            assert np.array_equal(c.host()[:1024], data)
            c.release()
    finally:
        ctx.set_option("host_pool_mb", 1024)


def test_host_access_round_trip_and_range_check(ctx):
    data = (frames.splitmix64(77, 4096) & np.uint64(0xFF)).astype(np.uint8)
    b = ctx.create_buffer(4096)
    b.host_access("writeonly", capi.QUEUE_LOAD, data)
    ctx.wait(capi.QUEUE_LOAD)
    b.host()[:] = 0                                # scribble over the mirror: READONLY must restore it
    b.host_access("readonly", capi.QUEUE_UNLOAD)
    assert np.array_equal(b.host(), data)
    with pytest.raises(capi.PhaneronError):
        b.host_access("writeonly", capi.QUEUE_LOAD, np.zeros(8192, np.uint8))
    # expose -> fill -> 'none' (Loader.init, loadSave.ts:76-99)
    b.host_access("writeonly", capi.QUEUE_LOAD)
    b.host()[:] = data[::-1]
    b.host_access("none", capi.QUEUE_LOAD)
    ctx.wait(capi.QUEUE_LOAD)
    b.host()[:] = 0
    b.host_access("readonly", capi.QUEUE_UNLOAD)
    assert np.array_equal(b.host(), data[::-1])
    b.release()


def test_run_program_by_argument_name_matches_oracle(ctx):
    """The reference's job queue shape: createProgram(name) + runProgram({argName: value})."""
    w, h = 1920, 8
    col = Colour(ctx, "709", "2020")
    src = frames.v210_random(w, h, 4242)
    wipg = frames.v210_pitch_pixels(w) // 48
    rd = ctx.create_program("phaneron:v210", "read", wipg * h, wipg)
    wr = ctx.create_program("phaneron:v210", "write", wipg * h, wipg)
    assert rd.kernel() == "v210_read" and wr.kernel() == "v210_write"
    vin = upload(ctx, src, svm="coarse")
    ctx.wait(capi.QUEUE_LOAD)
    rgba = ctx.create_buffer(w * h * 16, dims=(w, h))
    vout = ctx.create_buffer(src.nbytes, "writeonly")
    t = ctx.run_program(rd, {"input": vin, "output": rgba, "width": w, "colMatrix": col.rd_cm, "gammaLut": col.rd_lut,
                             "gamutMatrix": col.rd_gm})
    assert set(t) == {"dataToKernel", "kernelExec", "totalTime"} and t["totalTime"] >= t["kernelExec"]
    # argument order is irrelevant, names are what binds
    ctx.run_program(wr, {"gammaLut": col.wr_lut, "colMatrix": col.wr_cm, "interlace": 0, "width": w, "output": vout,
                         "input": rgba})
    ctx.wait()
    rgba.host_access("readonly", capi.QUEUE_UNLOAD)
    want_rgba = orc.v210_read(src, w, h, *col.oracle_rd)
    assert np.array_equal(rgba.host(np.uint32), want_rgba.reshape(-1).view(np.uint32))
    vout.host_access("readonly", capi.QUEUE_UNLOAD)
    assert np.array_equal(vout.host(np.uint32), orc.v210_write(want_rgba, w, h, 0, *col.oracle_wr))
    # ph_check_program: the same job examined without a launch - accepted, and the output is not touched
    vout.host_access("writeonly", capi.QUEUE_LOAD, np.zeros(src.size, np.uint32))
    ctx.wait(capi.QUEUE_LOAD)
    assert ctx.run_program(wr, {"gammaLut": col.wr_lut, "colMatrix": col.wr_cm, "interlace": 0, "width": w, "output": vout, "input": rgba},
                           check_only=True) is None
    ctx.wait()
    vout.host_access("readonly", capi.QUEUE_UNLOAD)
    assert not vout.host(np.uint32).any()
    with pytest.raises(capi.PhaneronError, match="'gammaLut'"):  # and refused with the launch's own words when an argument is missing
        ctx.run_program(wr, {"colMatrix": col.wr_cm, "interlace": 0, "width": w, "output": vout, "input": rgba}, check_only=True)
    # errors: a missing argument and an unknown kernel are reported, not ignored
    with pytest.raises(capi.PhaneronError, match="colMatrix"):
        ctx.run_program(rd, {"input": vin, "output": rgba, "width": w, "gammaLut": col.rd_lut, "gamutMatrix": col.rd_gm})
    with pytest.raises(capi.PhaneronError):
        ctx.create_program("__kernel void sharpen() {}", "sharpen", [w, h])
    for b in (vin, rgba, vout):
        b.release()
    rd.destroy(), wr.destroy()
    col.release()


def test_image_programs_by_name(ctx):
    w, h = 64, 32
    imgs = [frames.rgba_random(w, h, 900 + i) for i in range(3)]
    bufs = [upload(ctx, im, "readwrite", "coarse") for im in imgs]
    ctx.wait(capi.QUEUE_LOAD)
    out = ctx.create_buffer(w * h * 16, dims=(w, h))
    comb = ctx.create_program("phaneron:combine", "combine_3", [w, h])
    ctx.run_program(comb, {"l0In": bufs[0], "l1In": bufs[1], "l2In": bufs[2], "output": out})
    out.host_access("readonly", capi.QUEUE_UNLOAD)
    assert np.array_equal(out.host(np.uint32), orc.combine(imgs).reshape(-1).view(np.uint32))
    dis = ctx.create_program("phaneron:transition", "transition_dissolve", [w, h])
    ctx.run_program(dis, {"input0": bufs[0], "input1": bufs[1], "mix": 0.3, "output": out})
    out.host_access("readonly", capi.QUEUE_UNLOAD)
    assert np.array_equal(out.host(np.uint32), orc.transition_dissolve(imgs[0], imgs[1], 0.3).reshape(-1).view(np.uint32))
    yad = ctx.create_program("phaneron:yadif", "yadif", [w, h])
    ctx.run_program(yad, {"prev": bufs[0], "cur": bufs[1], "next": bufs[2], "parity": 1, "tff": 1, "skipSpatial": 0,
                          "output": out})
    out.host_access("readonly", capi.QUEUE_UNLOAD)
    assert np.array_equal(out.host(np.uint32), orc.yadif(imgs[0], imgs[1], imgs[2], 1, True, False).reshape(-1).view(np.uint32))
    for b in bufs + [out]:
        b.release()


def test_events_and_queue_ordering(ctx):
    """A kernel on PROCESS ordered behind an upload on LOAD by ph_queue_wait_queue, and a download on
    UNLOAD ordered behind the kernel, with no host wait in between."""
    w, h = 1920, 64
    a, b = frames.rgba_random(w, h, 1), frames.rgba_random(w, h, 2)
    ba, bb = ctx.create_buffer(a.nbytes, dims=(w, h)), ctx.create_buffer(a.nbytes, dims=(w, h))
    out = ctx.create_buffer(a.nbytes, dims=(w, h))
    ctx.wait(capi.QUEUE_LOAD)
    for buf, src in ((ba, a), (bb, b)):
        buf.host_access("writeonly", capi.QUEUE_LOAD)
        buf.host(np.float32)[:] = src.reshape(-1)
        buf.host_access("none", capi.QUEUE_LOAD)
    ctx.queue_wait_queue(capi.QUEUE_PROCESS, capi.QUEUE_LOAD)
    ctx.mixer(ba.device_ptr(), bb.device_ptr(), 0.25, out.device_ptr(), w, h)
    ctx.queue_wait_queue(capi.QUEUE_UNLOAD, capi.QUEUE_PROCESS)
    out.download_async(capi.QUEUE_UNLOAD)
    ev = ctx.record_event(capi.QUEUE_UNLOAD)
    ev.wait()
    assert ev.done()
    ev.destroy()
    assert np.array_equal(out.host(np.uint32), orc.mixer(a, b, 0.25).reshape(-1).view(np.uint32))
    for x in (ba, bb, out):
        x.release()


@pytest.mark.parametrize("depth", [1, 3])
def test_staged_channel_ring_reuse(ctx, depth):
    """7 different frames through a ring of `depth` slots: every output equals the oracle's chain
    for THAT frame (no slot is overwritten while its frame is still in flight)."""
    w, h, n, nframes = 1920, 12, 4, 7
    col = Colour(ctx, "709", "2020")
    vbytes = frames.v210_pitch_bytes(w) * h
    src = [[frames.v210_random(w, h, frames.layer_seed(f, i)) for i in range(n)] for f in range(nframes)]

    def process(c, sources, output):
        c.fused_v210_combine([s.device_ptr() for s in sources], output.device_ptr(), w, h, col.rd_cm.device_ptr(),
                             col.rd_lut.device_ptr(), col.rd_gm.device_ptr(), col.wr_cm.device_ptr(),
                             col.wr_lut.device_ptr())

    def fill(f, mirrors):
        for m, words in zip(mirrors, src[f]):
            m.view(np.uint32)[:] = words

    got = {}
    chan = staging.StagedChannel(ctx, [vbytes] * n, vbytes, process, depth=depth)
    for f in range(nframes):
        chan.submit(fill, lambda fr, mirror: got.__setitem__(fr, mirror.view(np.uint32).copy()))
    chan.drain(lambda fr, mirror: got.__setitem__(fr, mirror.view(np.uint32).copy()))
    assert sorted(got) == list(range(nframes))
    for f in range(nframes):
        want = orc.pipeline_v210_combine(src[f], w, h, *col.oracle_rd, *col.oracle_wr)
        assert np.array_equal(got[f], want), f
    chan.close()
    col.release()


def test_recorded_batch_replays_with_new_contents(ctx):
    """ph_graph_*: the reference-shaped batch read x2 -> combine_2 -> write recorded once and replayed on
    the same buffers with new frame contents each time equals the oracle chain for those contents."""
    w, h = 1920, 16
    col = Colour(ctx, "709", "2020")
    vbytes = frames.v210_pitch_bytes(w) * h
    srcs = [ctx.create_buffer(vbytes, "readonly", "coarse") for _ in range(2)]
    rgba = [ctx.create_buffer(w * h * 16, dims=(w, h)) for _ in range(2)]
    comb = ctx.create_buffer(w * h * 16, dims=(w, h))
    out = ctx.create_buffer(vbytes, "writeonly")
    for b in srcs:  # something to read while recording (a capture launches nothing)
        b.host_access("writeonly", capi.QUEUE_LOAD, np.zeros(vbytes, np.uint8))
    ctx.wait(capi.QUEUE_LOAD)
    ctx.register_lut(col.rd_lut.device_ptr(), capi.gamma2linear_lut("709"))
    ctx.register_lut(col.wr_lut.device_ptr(), capi.linear2gamma_lut("2020"))

    def batch():
        for s, r in zip(srcs, rgba):
            ctx.v210_read(s.device_ptr(), r.device_ptr(), w, h, col.rd_cm.device_ptr(), col.rd_lut.device_ptr(),
                          col.rd_gm.device_ptr())
        ctx.combine([r.device_ptr() for r in rgba], comb.device_ptr(), w, h)
        ctx.v210_write(comb.device_ptr(), out.device_ptr(), w, h, 0, col.wr_cm.device_ptr(), col.wr_lut.device_ptr())

    graph = ctx.record(batch)
    for f in range(3):
        layers = [frames.v210_random(w, h, frames.layer_seed(20 + f, i)) for i in range(2)]
        for b, l in zip(srcs, layers):
            b.host_access("writeonly", capi.QUEUE_LOAD, l)
        ctx.wait(capi.QUEUE_LOAD)
        graph.launch()
        ctx.wait()
        out.host_access("readonly", capi.QUEUE_UNLOAD)
        want = orc.pipeline_v210_combine(layers, w, h, *col.oracle_rd, *col.oracle_wr)
        assert np.array_equal(out.host(np.uint32), want), f
    graph.destroy()
    for b in srcs + rgba + [comb, out]:
        b.release()
    col.release()


def test_handles_outlive_their_context_in_any_order():
    """A garbage collector finalises a clContext and its OpenCLBuffers / programs / events in no particular
    order (node/ph_napi.c).  Destroying the context first must leave every handle releasable, refuse new
    work with a message, and free the device state with the last handle."""
    import ctypes as C
    l = capi.lib()
    c = capi.Context(0)
    buf = c.create_buffer(1 << 20, owner="orphan")
    lut = upload(c, capi.gamma2linear_lut("709"), svm="coarse")
    prog = c.create_program("phaneron:v210", "read", 40 * 8, 40)
    ev = c.record_event(capi.QUEUE_PROCESS)
    c.wait(capi.QUEUE_LOAD)
    h = c.h
    c.close()                                       # context first ...
    assert l.ph_wait_finish(h, 1) < 0 and b"destroyed" in l.ph_last_error(None)
    out = C.c_void_p()
    assert l.ph_buf_create(h, 64, 0, 0, 0, 0, b"late", C.byref(out)) < 0
    assert b"destroyed" in l.ph_last_error(None)
    assert buf.refcount() == 1
    buf.add_ref()
    assert buf.release() == 0 and buf.release() == 0  # ... then its buffers (the pool it returns to still exists)
    assert lut.release() == 0
    assert l.ph_program_destroy(prog.h) == 0
    ev.destroy()                                    # the last handle tears the device state down
    c2 = capi.Context(0)                            # and the device is usable again
    b2 = c2.create_buffer(1 << 20)
    b2.release()
    c2.close()


def test_out_of_range_queue_is_refused(ctx):
    """a queue index that is not load / process / unload is an error everywhere, never a silent alias of `process`"""
    with pytest.raises(capi.PhaneronError, match="queue 7"):
        ctx.wait(7)
    with pytest.raises(capi.PhaneronError, match="queue -1"):
        ctx.queue_idle(-1)
    assert not capi.lib().ph_ctx_stream(ctx.h, 3)
    b = ctx.create_buffer(64)
    with pytest.raises(capi.PhaneronError, match="queue 3"):
        b.host_access("writeonly", 3, np.zeros(64, np.uint8))
    with pytest.raises(capi.PhaneronError, match="queue 5"):
        ctx.transition_dissolve(b.device_ptr(), b.device_ptr(), 0.5, b.device_ptr(), 2, 2, queue=5)
    b.release()


def test_channel_program_by_name_with_a_dissolve(ctx):
    """'chan_compose_v210_<n>' through runProgram's argument names (what node/defer.js folds a recorded chain into): a plain
    layer under a placed one that dissolves against a smaller, rotated source - against the oracle's chain of operators"""
    import frames
    ow, oh = 384, 108
    a = frames.v210_random(ow, oh, frames.layer_seed(96, 0))
    b = frames.v210_random(192, 54, frames.layer_seed(96, 1))
    col = Colour(ctx, "709", "709")
    va, vb = upload(ctx, a, svm="coarse"), upload(ctx, b, svm="coarse")
    m_id = np.zeros(12, np.float32)
    m_id[:9] = capi.transform_matrix(ow, oh)
    m_in = np.zeros(12, np.float32)
    m_in[:9] = capi.transform_matrix(ow, oh, scale_x=0.8, scale_y=0.8, rotate=0.05)
    bm_id, bm_in = upload(ctx, m_id), upload(ctx, m_in)
    vout = ctx.create_buffer(capi.v210_pitch_bytes(ow) * oh, "writeonly", "coarse")
    ctx.wait(capi.QUEUE_LOAD)
    prog = ctx.create_program("phaneron:chan", "chan_compose_v210_2", [ow, oh])
    ctx.run_program(prog, {"output": vout, "colMatrix": col.rd_cm, "gammaLut": col.rd_lut, "gamutMatrix": col.rd_gm, "outColMatrix": col.wr_cm,
                           "outGammaLut": col.wr_lut, "interlace": 0, "l0In": va, "l0Width": ow, "l0Height": oh, "l1In": va, "l1Matrix": bm_id,
                           "l1Width": ow, "l1Height": oh, "l1Transition": 1, "l1Mix": 0.75, "l1IncomingIn": vb, "l1IncomingMatrix": bm_in,
                           "l1IncomingWidth": 192, "l1IncomingHeight": 54})
    ctx.wait()
    vout.host_access("readonly", capi.QUEUE_UNLOAD)
    got = vout.host(np.uint32).copy()
    ra = orc.v210_read(a, ow, oh, *col.oracle_rd)
    rb = orc.v210_read(b, 192, 54, *col.oracle_rd)
    d = orc.transition_dissolve(orc.transform(ra, m_id[:9], ow, oh), orc.transform(rb, m_in[:9], ow, oh), 0.75)
    want = orc.v210_write(orc.combine([ra, d]), ow, oh, 0, *col.oracle_wr)
    assert np.array_equal(got, np.asarray(want).reshape(-1))
    for x in (va, vb, bm_id, bm_in, vout):
        x.release()
    col.release()


def test_run_programs_puts_plain_read_channels_into_one_launch(ctx):
    """ph_run_programs with fused_v210_combine_<n> jobs (what node/defer.js makes of channels whose layers are plain reads): four frames of
    three layers, one of two layers, and a frame that READS the first frame's output (it has to wait for it: call order) - every output
    equals the oracle's read -> combine -> write of its own sources"""
    w, h = 384, 54
    col = Colour(ctx, "709", "709")
    src = [frames.v210_random(w, h, frames.layer_seed(120, i)) for i in range(7)]
    dev = [upload(ctx, a, svm="coarse") for a in src]
    nbytes = capi.v210_pitch_bytes(w) * h
    outs = [ctx.create_buffer(nbytes, "writeonly", "coarse") for _ in range(6)]
    ctx.wait(capi.QUEUE_LOAD)
    recipe = {"colMatrix": col.rd_cm, "gammaLut": col.rd_lut, "gamutMatrix": col.rd_gm, "outColMatrix": col.wr_cm, "outGammaLut": col.wr_lut}
    p3 = ctx.create_program("phaneron:fused", "fused_v210_combine_3", [w, h])
    p2 = ctx.create_program("phaneron:fused", "fused_v210_combine_2", [w, h])
    picks = [(0, 1, 2), (1, 2, 3), (3, 4, 5), (6, 0, 4), (5, 6), None]
    jobs = []
    for j, pick in enumerate(picks):
        if pick is None:  # layers: frame 0's OUTPUT under two sources
            params = dict(recipe, output=outs[j], l0In=outs[0], l1In=dev[1], l2In=dev[2])
        else:
            params = dict(recipe, output=outs[j], **{"l%dIn" % l: dev[i] for l, i in enumerate(pick)})
        jobs.append((p3 if len(params) - 6 == 3 else p2, params))
    ctx.run_programs(jobs)
    ctx.wait()
    rd = [orc.v210_read(a, w, h, *col.oracle_rd) for a in src]
    want = []
    for pick in picks:
        layers = [rd[i] for i in pick] if pick is not None else [orc.v210_read(want[0].reshape(-1), w, h, *col.oracle_rd), rd[1], rd[2]]
        want.append(np.asarray(orc.v210_write(orc.combine(layers), w, h, 0, *col.oracle_wr)).reshape(-1))
    for j, o in enumerate(outs):
        o.host_access("readonly", capi.QUEUE_UNLOAD)
        assert np.array_equal(o.host(np.uint32), want[j].view(np.uint32) if want[j].dtype != np.uint32 else want[j]), "frame %d" % j
    for x in dev + outs:
        x.release()
    col.release()


def test_run_programs_random_calls_with_frames_feeding_frames(ctx):
    """seeded random ph_run_programs calls of fused_v210_combine_<n> and chan_compose_v210_<n> jobs in which a frame's layers may be EARLIER
    frames of the same call (a channel routed into another: call order must hold however the library groups its launches), outputs may
    be written twice, layer counts vary: every output equals the same jobs posted one ph_run_program each"""
    w, h = 384, 54
    r = np.random.default_rng(int(os.environ.get("PH_FUZZ_SEED", "77")))
    col = Colour(ctx, "709", "709")
    nbytes = capi.v210_pitch_bytes(w) * h
    dev = [upload(ctx, frames.v210_random(w, h, frames.layer_seed(130, i)), svm="coarse") for i in range(6)]
    mat = np.zeros(12, np.float32)
    mat[:9] = capi.transform_matrix(w, h, scale_x=0.5, scale_y=0.5, offset_x=0.2)
    fill = np.zeros(12, np.float32)
    fill[:9] = capi.transform_matrix(w, h)
    bm, bf = upload(ctx, mat), upload(ctx, fill)
    ctx.wait(capi.QUEUE_LOAD)
    recipe = {"colMatrix": col.rd_cm, "gammaLut": col.rd_lut, "gamutMatrix": col.rd_gm, "outColMatrix": col.wr_cm, "outGammaLut": col.wr_lut}
    fused = {n: ctx.create_program("phaneron:fused", "fused_v210_combine_%d" % n, [w, h]) for n in (1, 2, 3)}
    chan = {n: ctx.create_program("phaneron:chan", "chan_compose_v210_%d" % n, [w, h]) for n in (1, 2)}
    for case in range(int(os.environ.get("PH_FUZZ_CASES", "25"))):
        # [one call, separate calls]: the frames start out equal on both sides (a layer may be a frame nobody of the call has written yet)
        outs = [[upload(ctx, frames.v210_random(w, h, frames.layer_seed(140, i)), svm="coarse") for i in range(5)] for _ in range(2)]
        ctx.wait(capi.QUEUE_LOAD)
        spec = []
        for j in range(int(r.integers(2, 9))):
            n = int(r.integers(1, 4))
            kind = "fused" if r.random() < 0.6 else "chan"
            if kind == "chan":
                n = min(n, 2)
            o = int(r.integers(0, 5))
            layers = [("out", int(r.integers(0, 5))) if r.random() < 0.3 else ("src", int(r.integers(0, 6))) for _ in range(n)]
            layers = [("src", i) if which == "out" and i == o else (which, i) for which, i in layers]  # (no frame made from itself: a race in any context)
            spec.append((kind, n, o, layers, [bool(r.random() < 0.5) for _ in range(n)]))
        for side in (0, 1):
            jobs = []
            for kind, n, o, layers, small in spec:
                buf = lambda which, i: outs[side][i] if which == "out" else dev[i]
                params = dict(recipe, output=outs[side][o])
                for l, (which, i) in enumerate(layers):
                    params["l%dIn" % l] = buf(which, i)
                    if kind == "chan":
                        params.update({"l%dWidth" % l: w, "l%dHeight" % l: h, "l%dMatrix" % l: bm if small[l] else bf})
                if kind == "chan":
                    params["interlace"] = 0
                jobs.append(((fused if kind == "fused" else chan)[n], params))
            if side == 0:
                ctx.run_programs(jobs)
            else:
                for prog, params in jobs:
                    ctx.run_program(prog, params)
            ctx.wait()
        for i in range(5):
            a, b = outs[0][i], outs[1][i]
            a.host_access("readonly", capi.QUEUE_UNLOAD)
            b.host_access("readonly", capi.QUEUE_UNLOAD)
            assert np.array_equal(a.host(np.uint32), b.host(np.uint32)), "case %d: frame %d of the one call differs from the separate calls: %r" % (case, i, spec)
        for x in outs[0] + outs[1]:
            x.release()
    for x in dev + [bm, bf]:
        x.release()
    col.release()


def test_host_mirror_pool_covers_the_working_set():
    """the pool of pinned mirrors keeps what was in use at once even when that is more than `host_pool_mb` - a steady stream of
    create / map / release rounds pins nothing after the first round (36 images a tick was 40 ms a tick under a fixed 1 GiB budget)"""
    from phaneron_amd import capi
    with capi.Context(0) as ctx:
        ctx.set_option("host_pool_mb", 8)          # far less than a round's 64 x 1 MiB
        pins = []
        for rnd in range(4):
            bufs = [ctx.create_buffer(1 << 20) for _ in range(64)]
            for b in bufs:
                b.host()                            # attaches the mirror (the node binding does this for every buffer)
            for b in bufs:
                b.release()
            pins.append(ctx.host_pool_stats()["pins"])
        st = ctx.host_pool_stats()
        assert pins[0] == 64 and pins[1:] == [64, 64, 64], pins
        assert st["in_use"] == 0 and st["peak_in_use"] == 64 << 20 and st["pooled"] == 64 << 20, st
        # another size: the old blocks make room oldest first, the budget is still the peak
        bufs = [ctx.create_buffer(2 << 20) for _ in range(32)]
        for b in bufs:
            b.host()
        for b in bufs:
            b.release()
        st2 = ctx.host_pool_stats()
        assert st2["pooled"] <= 64 << 20 and st2["pins"] == 64 + 32, st2


def test_packed_fields_unpack_in_place():
    """ph_image_unpack_rgb (program "rgb_unpack"): the de-interlacing reader's packed-RGB fields written into RGBA-sized buffers and expanded in
    place are, bit for bit, the RGBA fields the same launch writes directly (alpha 1) - what node/defer.js hands to anybody but the 2 x 2-block
    compositor; at a size whose image does not fit the scratch area's first allocation, twice in a row (the scratch is reused)"""
    import torch
    import frames
    import hip_harness as hh
    k = hh.ctx()
    rd = hh.ColourParams.reader("709", "2020")
    for w, h in ((192, 54), (1920, 1080), (1920, 1080)):
        win = [hh.dev(frames.v210_random(w, h, frames.layer_seed(1400 + w, t)).reshape(-1)) for t in range(3)]
        rgba = [torch.zeros(w * h * 4, dtype=torch.float32, device="cuda") for _ in range(2)]
        packed = [torch.full((w * h * 4,), 7.0, dtype=torch.float32, device="cuda") for _ in range(2)]
        k.v210_yadif_pair([(win[0], win[1], win[2], rgba[0], rgba[1])], w, h, 1, False, *rd)
        k.v210_yadif_pair([(win[0], win[1], win[2], packed[0], packed[1])], w, h, 1, False, *rd, rgb=True)
        for p in packed:
            k.image_unpack_rgb(p, w, h)
        for a, b in zip(rgba, packed):
            assert np.array_equal(hh.host(a).view(np.uint32), hh.host(b).view(np.uint32)), (w, h)


def test_run_programs_puts_like_compositor_frames_into_one_launch(ctx):
    """ph_run_programs with compose_up_write_v210_<n> jobs (what node/defer.js makes of channels showing de-interlaced fields): two channels' jobs,
    each BOTH fields of its frame (output2 / l<i>In2), are one launch of four frames; a third job of another placement starts the next; a job
    that writes a frame the group already writes starts the next too - every frame equals the same jobs posted one ph_run_program each, and
    the trace says how the call was launched"""
    w, h, sw, sh = 384, 54, 192, 30
    col = Colour(ctx, "709", "709")
    imgs = [upload(ctx, frames.rgba_random(sw, sh, 1500 + i, -0.05, 1.05).reshape(-1), svm="coarse", dims=(sw, sh)) for i in range(6)]
    fill, small = np.zeros(12, np.float32), np.zeros(12, np.float32)
    fill[:9] = capi.transform_matrix(w, h)
    small[:9] = capi.transform_matrix(w, h, scale_x=0.8, scale_y=0.8)
    bf, bs = upload(ctx, fill), upload(ctx, small)
    ctx.wait(capi.QUEUE_LOAD)
    nbytes = capi.v210_pitch_bytes(w) * h
    prog = ctx.create_program("phaneron:up", "compose_up_write_v210_1", [w, h])
    saver = {"outColMatrix": col.wr_cm, "outGammaLut": col.wr_lut, "interlace": 0}
    sides = []
    for side in (0, 1):
        outs = [ctx.create_buffer(nbytes, "writeonly", "coarse") for _ in range(6)]
        jobs = [(prog, dict(saver, l0In=imgs[0], l0In2=imgs[1], l0Matrix=bf, output=outs[0], output2=outs[1])),   # channel A: both fields
                (prog, dict(saver, l0In=imgs[2], l0In2=imgs[3], l0Matrix=bf, output=outs[2], output2=outs[3])),   # channel B: both fields -> the same launch
                (prog, dict(saver, l0In=imgs[4], l0Matrix=bs, output=outs[4])),                                   # another placement: a launch of its own
                (prog, dict(saver, l0In=imgs[5], l0Matrix=bs, output=outs[5])),                                   # ... shared with this one
                (prog, dict(saver, l0In=imgs[0], l0Matrix=bs, output=outs[4]))]                                   # writes a frame of the group: the next launch
        if side == 0:
            with capi.trace() as t:
                ctx.run_programs(jobs)
            assert t.route == "compose_up_write_v210+compose_up_write_v210+compose_up_write_v210", t.route
        else:
            for p, params in jobs:
                ctx.run_program(p, params)
        ctx.wait()
        got = []
        for o in outs:
            o.host_access("readonly", capi.QUEUE_UNLOAD)
            got.append(o.host(np.uint32).copy())
        sides.append(got)
        for o in outs:
            o.release()
    for i, (a, b) in enumerate(zip(*sides)):
        assert np.array_equal(a, b), "frame %d of the one call differs from the separate calls" % i
    assert not np.array_equal(sides[0][0], sides[0][2])
    for x in imgs + [bf, bs]:
        x.release()
    col.release()


def test_run_programs_reports_how_far_a_failing_call_got(ctx):
    """ph_run_programs_progress: a call whose second group is refused at its launch (fault injection: option fail_launches = -1 lets one
    launch group through, then every one fails) has made the first group's frames - the binding continues behind them instead of rendering
    them twice; a call refused by its checks has made nothing"""
    import ctypes
    w, h = 384, 54
    col = Colour(ctx, "709", "709")
    dev = [upload(ctx, frames.v210_random(w, h, frames.layer_seed(160, i)), svm="coarse") for i in range(4)]
    nbytes = capi.v210_pitch_bytes(w) * h
    outs = [ctx.create_buffer(nbytes, "writeonly", "coarse") for _ in range(4)]
    ctx.wait(capi.QUEUE_LOAD)
    recipe = {"colMatrix": col.rd_cm, "gammaLut": col.rd_lut, "gamutMatrix": col.rd_gm, "outColMatrix": col.wr_cm, "outGammaLut": col.wr_lut}
    p2 = ctx.create_program("phaneron:fused", "fused_v210_combine_2", [w, h])
    p1 = ctx.create_program("phaneron:fused", "fused_v210_combine_1", [w, h])
    # two frames of two layers (one launch), then two of one layer (another launch: another layer count)
    jobs = [(p2, dict(recipe, output=outs[0], l0In=dev[0], l1In=dev[1])), (p2, dict(recipe, output=outs[1], l0In=dev[2], l1In=dev[3])),
            (p1, dict(recipe, output=outs[2], l0In=dev[0])), (p1, dict(recipe, output=outs[3], l0In=dev[1]))]
    done = ctypes.c_int(-1)
    ctx.set_option("fail_launches", -1)
    try:
        with pytest.raises(capi.PhaneronError, match="injected"):
            ctx.run_programs(jobs)
        capi.check(capi.lib().ph_run_programs_progress(ctypes.byref(done)))
        assert done.value == 2
    finally:
        ctx.set_option("fail_launches", 0)
    ctx.run_programs(jobs[done.value:])  # the binding's continuation
    capi.check(capi.lib().ph_run_programs_progress(ctypes.byref(done)))
    assert done.value == 2
    ctx.wait()
    rd = [orc.v210_read(frames.v210_random(w, h, frames.layer_seed(160, i)), w, h, *col.oracle_rd) for i in range(4)]
    want = [orc.combine([rd[0], rd[1]]), orc.combine([rd[2], rd[3]]), rd[0], rd[1]]
    for o, img in zip(outs, want):
        o.host_access("readonly", capi.QUEUE_UNLOAD)
        assert np.array_equal(o.host(np.uint32), np.asarray(orc.v210_write(img, w, h, 0, *col.oracle_wr)).reshape(-1).view(np.uint32))
    bad = dict(recipe, output=outs[0], l0In=dev[0])  # a two-layer program without its second layer: refused by the checks, nothing launched
    with pytest.raises(capi.PhaneronError):
        ctx.run_programs([jobs[0], (p2, bad)])
    capi.check(capi.lib().ph_run_programs_progress(ctypes.byref(done)))
    assert done.value == 0
    for x in dev + outs:
        x.release()
    col.release()
