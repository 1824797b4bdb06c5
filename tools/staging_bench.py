#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline pipeline (4 x 2160p v210 in -> 1 x 2160p v210 out) when the
boundary hands over HOST buffers, three ways (one JSON line each):

  serial    the reference's own sequence (io.ts loadFrame / processFrame / saveFrame): hostAccess
            copies each source in, waitFinish(load), kernels, waitFinish(process), hostAccess
            readonly - one frame at a time
  staged    phaneron_amd/staging.py ring (depth 3), device-side queue ordering; the producer's
            frame is in ordinary host memory and is copied into the pinned mirror (memcpy)
  staged0   same, the producer decodes straight into the pinned mirror (no host memcpy)

The bench.py `value` is the HBM-resident rate; DESIGN.md quotes these next to it."""
import json
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from phaneron_amd import capi, staging
import frames


def main():
    w, h, n = 3840, 2160, 4
    nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    ctx = capi.Context(0)
    vbytes = capi.v210_pitch_bytes(w) * h

    def up(a, svm="none"):
        a = np.ascontiguousarray(a)
        b = ctx.create_buffer(a.nbytes, "readonly", svm)
        b.host_access("writeonly", capi.QUEUE_LOAD, a)
        return b

    g2l, l2g = capi.gamma2linear_lut("709"), capi.linear2gamma_lut("2020")
    rd_cm, rd_lut, rd_gm = up(capi.ycbcr2rgb_matrix("709")), up(g2l, "coarse"), up(capi.rgb2rgb_matrix("709", "2020"))
    wr_cm, wr_lut = up(capi.rgb2ycbcr_matrix("2020")), up(l2g, "coarse")
    ctx.wait(capi.QUEUE_LOAD)
    ctx.register_lut(rd_lut.device_ptr(), g2l)
    ctx.register_lut(wr_lut.device_ptr(), l2g)
    src = [frames.v210_random(w, h, frames.layer_seed(0, i)) for i in range(n)]  # pageable host memory

    def process(c, sources, output):
        c.fused_v210_combine([s.device_ptr() for s in sources], output.device_ptr(), w, h, rd_cm.device_ptr(),
                             rd_lut.device_ptr(), rd_gm.device_ptr(), wr_cm.device_ptr(), wr_lut.device_ptr())

    def report(mode, el, extra=""):
        fps = nframes / el
        print(json.dumps({"mode": mode, "frames": nframes, "frames_per_sec": round(fps, 1),
                          "host_to_device_GBps": round(fps * n * vbytes / 1e9, 2),
                          "device_to_host_GBps": round(fps * vbytes / 1e9, 2), "note": extra}), flush=True)

    # ---- serial: the reference's sequence --------------------------------------------------------
    sources = [ctx.create_buffer(vbytes, "readonly", "coarse") for _ in range(n)]
    out = ctx.create_buffer(vbytes, "writeonly", "coarse")
    for warm in (True, False):
        t0 = time.perf_counter()
        for f in range(3 if warm else nframes):
            for b, s in zip(sources, src):
                b.host_access("writeonly", capi.QUEUE_LOAD, s)
            ctx.wait(capi.QUEUE_LOAD)
            process(ctx, sources, out)
            ctx.wait(capi.QUEUE_PROCESS)
            out.host_access("readonly", capi.QUEUE_UNLOAD)
        el = time.perf_counter() - t0
    report("serial", el, "hostAccess(src) + waitFinish per stage, one frame in flight")
    want = out.host(np.uint32).copy()
    for b in sources + [out]:
        b.release()

    # ---- staged ring ---------------------------------------------------------------------------------
    for mode in ("staged", "staged0"):
        chan = staging.StagedChannel(ctx, [vbytes] * n, vbytes, process, depth=3)
        if mode == "staged0":
            for slot in chan.slots:  # the "decoder" has written the mirrors already
                for b, s in zip(slot.sources, src):
                    b.host(np.uint32)[:] = s

        def fill(f, mirrors):
            if mode == "staged":
                for m, s in zip(mirrors, src):
                    m.view(np.uint32)[:] = s

        last = {}
        consume = lambda fr, mirror: last.__setitem__("frame", fr)
        for f in range(3):
            chan.submit(fill, consume)
        chan.drain(consume)
        t0 = time.perf_counter()
        for f in range(nframes):
            chan.submit(fill, consume)
        chan.drain(consume)
        el = time.perf_counter() - t0
        ok = bool(np.array_equal(chan.slots[(chan.submitted - 1) % 3].output.host(np.uint32), want))
        report(mode, el, "ring depth 3, device-side ordering; output identical to serial: %s" % ok)
        chan.close()
    ctx.close()


if __name__ == "__main__":
    main()
