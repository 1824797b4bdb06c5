#!/usr/bin/env python3
"""Secondary workloads of BASELINE.json (configs 2 and 3) through the C ABI, every route the library offers.

  config 2: 1 channel, 4 layers 1080p50: read x5 (4 layers + the transition's second source) -> transform
            (full frame / three quarter-size insets) -> transition_wipe on layer 4 (ramp mask) -> combine_4 -> write
  config 3: 1 channel, 4 layers -> 2160p50 from 1080i50: read(709->2020) (one new frame per layer per two fields) ->
            yadif send_field -> transform 1080->2160 -> combine_4 -> write(2020)

Algorithmic bytes: every v210 input once + the v210 output once; f32 intermediates count 0.
  python tools/config_bench.py            prints one JSON line per (config, route)
bench.py imports `measure(..., routes="best")` for its `secondary` field.
"""
import json
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def measure(ctx, torch, np, capi, routes="all", reps=200):
    """routes: "all" or "best" (the fastest route of each config).  Returns a list of records.  A record's `kernel_launches` names the
    kernels of its route and how many launches of each one unit (frame / field) takes: bench.py adds the counters of a profiled
    child run of this script (VALU instructions, HBM traffic) per unit from them."""
    stream = ctx.torch_stream()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def colour(rspec, wspec):
        rd = [dev(capi.ycbcr2rgb_matrix(rspec)), dev(capi.gamma2linear_lut(rspec)),
              dev(np.concatenate([capi.rgb2rgb_matrix(rspec, wspec), np.zeros(3, np.float32)]))]
        wr = [dev(capi.rgb2ycbcr_matrix(wspec)), dev(capi.linear2gamma_lut(wspec))]
        torch.cuda.synchronize()
        ctx.register_lut(rd[1], capi.gamma2linear_lut(rspec))
        ctx.register_lut(wr[1], capi.linear2gamma_lut(wspec))
        return rd, wr

    def v210(w, h, n):
        words = capi.v210_pitch_bytes(w) * h // 4
        return [torch.randint(0, 2 ** 30, (words,), dtype=torch.int32, device="cuda") & 0x3FFFFFFF for _ in range(n)]

    def img(w, h, n=1):
        return [torch.empty(w * h * 4, dtype=torch.float32, device="cuda") for _ in range(n)]

    def timeit(fn, n):
        if only and not any(current[0].startswith(x) for x in only):  # PH_CONFIG_BENCH_ONLY: this config's kernels are not run at all
            return float("nan")
        # the chip needs ~50 ms of load before its clocks settle (bench.py's FIXED_WARMUP, profiles/r02_bench_repeat.jsonl): a route is
        # run untimed for 0.15 s first - measured straight after an idle spell, the first route of a config came out 8 % slower than
        # the same route measured second (config 3: 113.9 against 105.2 us per field)
        import time
        i, t0 = 0, time.perf_counter()
        while i < 4 or time.perf_counter() - t0 < float(os.environ.get("PH_CONFIG_BENCH_WARM_S", "0.15")):
            fn(i)
            i += 1
            if i % 64 == 0:
                ctx.wait()
        i += i & 1  # routes alternate fields: start on an even index
        ctx.wait()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for k in range(i, i + n):
            fn(k)
        e1.record(stream)
        ctx.wait()
        return e0.elapsed_time(e1) / n

    out_records = []
    only = [x for x in os.environ.get("PH_CONFIG_BENCH_ONLY", "").split(",") if x]  # config-name prefixes (bench.py's counter passes take the groups one by one)
    current = [""]  # the config whose routes are being timed (set before each config's first timeit)

    def record(config, route, unit, ms, algo, kernels, kernel_launches=None, **extra):
        if ms != ms:  # not selected (timeit)
            return
        rec = {"config": config, "route": route, "kernels_per_%s" % unit: kernels, "ms_per_%s" % unit: round(ms, 4),
               "%ss_per_sec" % unit: round(1e3 / ms, 1), "algorithmic_bytes": algo,
               "roofline": {"bound": "hbm", "achieved": round(algo / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(algo / ms / 1e6 / HBM_PEAK_GBS, 4)},
               "x_realtime_50fps": round(1e3 / ms / 50, 1)}
        if kernel_launches:
            rec["kernel_launches"] = kernel_launches
        rec.update(extra)
        out_records.append(rec)

    R = 8
    # ---------------- config 2 -------------------------------------------------------------
    w, h = 1920, 1080
    rd, wr = colour("709", "709")
    src = [v210(w, h, 5) for _ in range(R)]          # 4 layers + the wipe's second source
    rgba = img(w, h, 5)
    xf = img(w, h, 4)
    trans, comb = img(w, h)[0], img(w, h)[0]
    out = torch.empty(capi.v210_pitch_bytes(w) * h // 4, dtype=torch.int32, device="cuda")
    mask = torch.zeros(h, w, 4, device="cuda")
    mask[..., 0] = torch.linspace(0, 1, w, device="cuda")[None, :]
    mask = mask.reshape(-1).contiguous()
    mats = [dev(capi.transform_matrix(w, h))] + [
        dev(capi.transform_matrix(w, h, scale_x=0.5, scale_y=0.5, offset_x=ox, offset_y=oy))
        for ox, oy in ((-0.25, -0.25), (0.25, -0.25), (0.25, 0.25))]
    torch.cuda.synchronize()

    def config2(i):
        s = src[i % R]
        for l in range(5):
            ctx.v210_read(s[l], rgba[l], w, h, *rd)
        for l in range(4):
            ctx.transform(rgba[l], w, h, mats[l], xf[l], w, h)
        ctx.transition_wipe(xf[3], rgba[4], mask, trans, w, h)
        ctx.combine([xf[0], xf[1], xf[2], trans], comb, w, h)
        ctx.v210_write(comb, out, w, h, 0, *wr)

    def config2_fused(i, batch=False):  # same frame with transform x3 + combine_4 + write as one kernel
        s = src[i % R]
        if batch:
            ctx.v210_read_batch(s, rgba, w, h, *rd)
        else:
            for l in range(5):
                ctx.v210_read(s[l], rgba[l], w, h, *rd)
        ctx.transform(rgba[3], w, h, mats[3], xf[3], w, h)
        ctx.transition_wipe(xf[3], rgba[4], mask, trans, w, h)
        ctx.compose_write_v210([(rgba[0], w, h, mats[0]), (rgba[1], w, h, mats[1]), (rgba[2], w, h, mats[2]),
                                (trans, w, h, None)], out, w, h, 0, *wr)

    def config2_wipe_inside(i):  # [read x5], then ONE compositor launch: transform x4 + transition_wipe + combine_4 + write
        s = src[i % R]
        ctx.v210_read_batch(s, rgba, w, h, *rd)
        ctx.compose_wipe_write_v210([(rgba[l], w, h, mats[l]) for l in range(4)], [None, None, None, (rgba[4], mask)],
                                    out, w, h, 0, *wr)

    mats_h = [capi.transform_matrix(w, h)] + [capi.transform_matrix(w, h, scale_x=0.5, scale_y=0.5, offset_x=ox, offset_y=oy)
                                              for ox, oy in ((-0.25, -0.25), (0.25, -0.25), (0.25, 0.25))]
    # ONE launch per frame, no f32 frame at all: the compositor samples the v210 words (ph_chan_compose_v210)
    chan_jobs = [ctx.chan_compose_v210(
        [dict(src=(s[l], w, h, mats_h[l])) for l in range(3)] +
        [dict(src=(s[3], w, h, mats_h[3]), transition="wipe", incoming=(s[4], w, h, None), mask=(mask, w, h, None, "rgba"))],
        out, w, h, 0, *rd, *wr, prepare_only=True) for s in src]

    def config2_chan(i):
        chan_jobs[i % R]()

    # 4 layers + second source + the mask counted as one more v210-sized input + 1 output (38 707 200 during a transition)
    algo2 = 7 * capi.v210_pitch_bytes(w) * h
    name2 = "2: 1 channel, 4-layer 1080p50, three quarter-size insets, wipe transition on the top layer"
    # what the benched kernel really reads and writes: five v210 frames in, one out, and the wipe's mask as an f32 RGBA image
    # (SURVEY's 38.7 MB counts the mask as a sixth v210-sized input)
    benched2 = 6 * capi.v210_pitch_bytes(w) * h + w * h * 16
    current[0] = name2
    record(name2, "channel compositor straight from v210 (ph_chan_compose_v210): [read x5 + transform x4 + transition_wipe + combine_4 + write] "
           "as one kernel, no f32 frame in HBM", "frame", timeit(config2_chan, reps), algo2, 1, {"chan_compose_v210_kernel<0, 0>": 1.0},
           parity_test="tests/test_chan_gpu.py::test_config2_full_size_with_wipe",
           bytes_as_benched=benched2, bytes_as_benched_note="5 v210 frames in + 1 out + the wipe's mask as the f32 RGBA image the kernel reads "
           "(33 MB); algorithmic_bytes is SURVEY 8d's figure, which counts the mask as a v210-sized input")
    # round 5: the reference runs FOUR channels of <= 1080p in one context through one queue (src/index.ts:45-71,156-160): their frames
    # of a tick in ONE launch (ph_chan_compose_batch) - per channel frame
    C2 = 4
    outs2 = [torch.empty_like(out) for _ in range(C2)]
    chan_layers = lambda s: ([dict(src=(s[l], w, h, mats_h[l])) for l in range(3)] +
                             [dict(src=(s[3], w, h, mats_h[3]), transition="wipe", incoming=(s[4], w, h, None), mask=(mask, w, h, None, "rgba"))])
    batch_jobs = [ctx.chan_compose_batch([(chan_layers(src[(i + j) % R]), outs2[j], 0) for j in range(C2)], w, h, *rd, *wr, prepare_only=True) for i in range(R)]
    record("2 x 4: four channels of config 2's shape in one context, their frames of a tick in one launch (per channel frame)",
           "ph_chan_compose_batch: the four frames share the workgroups of one launch - tables loaded once, the wave steps of all four taken "
           "by the waves as they come free", "frame", timeit(lambda i: batch_jobs[i % R](), reps) / C2, algo2, 1.0 / C2,
           {"chan_compose_batch_kernel<false, false>": 1.0 / C2}, bytes_as_benched=benched2, channels_per_launch=C2,
           parity_test="tests/test_chan_gpu.py::test_chan_batch_full_size_four_1080p_channels")
    if routes == "all":
        record(name2, "batched reads + compositor with the wipe inside: [read x5], [transform x4 + transition_wipe + combine_4 + write]", "frame",
               timeit(config2_wipe_inside, reps), algo2, 2)
        record(name2, "fused compositor, batched reads: [read x5], transform, transition_wipe, [transform x3 + combine_4 + write]", "frame",
               timeit(lambda i: config2_fused(i, True), reps), algo2, 4)
        record(name2, "fused compositor: read x5, transform, transition_wipe, [transform x3 + combine_4 + write]", "frame",
               timeit(config2_fused, reps), algo2, 8)
        record(name2, "one kernel per operator (the reference's job batch)", "frame", timeit(config2, reps), algo2, 13)
        graphs = [ctx.record(lambda i=i: config2_fused(i)) for i in range(R)]
        record(name2, "fused compositor, each frame's batch replayed as one hipGraph", "frame",
               timeit(lambda i: graphs[i % R].launch(), reps), algo2, 8)

    # ---------------- 720p50: the reference's third video format (src/config.ts:43-54) -------------------
    # 1280 % 48 = 32, 1280 % 6 = 2: every line ends in a tail quad and two cleared slots (v210.ts:84-110, 166-193).  Two figures:
    # the headline's shape (4 x 1:1 layers, ph_fused_v210_combine) and config 2's placements without the transition (ph_chan_compose_v210)
    w7, h7 = 1280, 720
    rd7, wr7 = colour("709", "709")
    src7 = [v210(w7, h7, 4) for _ in range(R)]
    out7 = torch.empty(capi.v210_pitch_bytes(w7) * h7 // 4, dtype=torch.int32, device="cuda")
    mats7 = [capi.transform_matrix(w7, h7)] + [capi.transform_matrix(w7, h7, scale_x=0.5, scale_y=0.5, offset_x=ox, offset_y=oy)
                                               for ox, oy in ((-0.25, -0.25), (0.25, -0.25), (0.25, 0.25))]
    chan7 = [ctx.chan_compose_v210([dict(src=(s[l], w7, h7, mats7[l])) for l in range(4)], out7, w7, h7, 0, *rd7, *wr7, prepare_only=True) for s in src7]
    torch.cuda.synchronize()
    algo7 = 5 * capi.v210_pitch_bytes(w7) * h7
    name7 = "720p50: 1 channel, 4 x 1280x720 v210 layers -> 1 v210 frame (lines with a tail quad: the reference's tail arithmetic)"
    current[0] = name7
    record(name7, "fused unpack / CSC / combine_4 / CSC / pack, 1:1 layers (ph_fused_v210_combine, tail instantiation)", "frame",
           timeit(lambda i: ctx.fused_v210_combine(src7[i % R], out7, w7, h7, *rd7, *wr7), reps), algo7, 1, {"fused_v210_combine_lds_kernel": 1.0},
           parity_test="tests/test_chains_gpu.py::test_720p_headline_shape_full_size")
    record(name7, "channel compositor straight from v210, a full-frame layer and three quarter-size insets (ph_chan_compose_v210, general instantiation)", "frame",
           timeit(lambda i: chan7[i % R](), reps), algo7, 1, {"chan_compose_v210_kernel<1, 0>": 1.0}, parity_test="tests/test_chains_gpu.py::test_720p_chain_full_size")
    C7 = 4
    outs7 = [torch.empty_like(out7) for _ in range(C7)]
    batch7 = [ctx.chan_compose_batch([([dict(src=(src7[(i + j) % R][l], w7, h7, mats7[l])) for l in range(4)], outs7[j], 0) for j in range(C7)],
                                     w7, h7, *rd7, *wr7, prepare_only=True) for i in range(R)]
    record("720p50 x 4: four such channels in one context, their frames of a tick in one launch (per channel frame)",
           "ph_chan_compose_batch, a full-frame layer and three quarter-size insets per channel", "frame",
           timeit(lambda i: batch7[i % R](), reps) / C7, algo7, 1.0 / C7, {"chan_compose_batch_kernel<true, false>": 1.0 / C7}, channels_per_launch=C7,
           parity_test="tests/test_fullsize_gpu.py::test_four_720p_channels_in_one_launch_full_size")

    # ---------------- config 3 -------------------------------------------------------------
    sw, sh, ow, oh = 1920, 1080, 3840, 2160
    rd, wr = colour("709", "2020")
    srcs = [v210(sw, sh, 4) for _ in range(R)]
    win = [img(sw, sh, 3) for _ in range(4)]         # prev / cur / next per layer
    deint = img(sw, sh, 4)
    out = torch.empty(capi.v210_pitch_bytes(ow) * oh // 4, dtype=torch.int32, device="cuda")
    mh = capi.transform_matrix(ow, oh)
    m = dev(mh)
    torch.cuda.synchronize()

    def new_frames(i, batch=False):  # a new source frame is unpacked every second field (send_field: two outputs per frame)
        s = srcs[(i // 2) % R]
        if not (i & 1):
            for l in range(4):
                win[l] = [win[l][1], win[l][2], win[l][0]]
                if not batch:
                    ctx.v210_read(s[l], win[l][2], sw, sh, *rd)
            if batch:
                ctx.v210_read_batch(s, [win[l][2] for l in range(4)], sw, sh, *rd)
        return 1 ^ (0 if (i & 1) else 1)             # parity = tff ^ !second (yadif.ts:104), tff = 1

    def config3_fused(i):  # yadif per layer, then upscale x4 + combine_4 + write as one kernel
        parity = new_frames(i)
        for l in range(4):
            ctx.yadif(win[l][0], win[l][1], win[l][2], deint[l], sw, sh, parity, 1, False)
        ctx.compose_write_v210([(deint[l], sw, sh, m) for l in range(4)], out, ow, oh, 0, *wr)

    deint2 = [img(sw, sh, 2) for _ in range(4)]       # both fields of the current frame, per layer

    def config3_pair(i):  # both fields of a frame de-interlaced in one pass per layer; one compositor launch per field
        if not (i & 1):
            new_frames(i, True)
            for l in range(4):
                ctx.yadif_pair(win[l][0], win[l][1], win[l][2], deint2[l][0], deint2[l][1], sw, sh, 1, False)
        parity = 1 ^ (0 if (i & 1) else 1)
        ctx.compose_write_v210([(deint2[l][parity], sw, sh, m) for l in range(4)], out, ow, oh, 0, *wr)

    vwin = [[srcs[k % R][l] for k in range(3)] for l in range(4)]  # v210 windows: prev / cur / next per layer

    def config3_deint(i):  # per frame ONE launch: unpack + yadif of both fields, all four layers; per field the compositor
        if not (i & 1):
            s = srcs[(i // 2) % R]
            for l in range(4):
                vwin[l] = [vwin[l][1], vwin[l][2], s[l]]
            ctx.v210_yadif_pair([(vwin[l][0], vwin[l][1], vwin[l][2], deint2[l][0], deint2[l][1]) for l in range(4)], sw, sh, 1, False, *rd)
        parity = 1 ^ (0 if (i & 1) else 1)
        ctx.compose_write_v210([(deint2[l][parity], sw, sh, m) for l in range(4)], out, ow, oh, 0, *wr)

    # packed-RGB fields (12 bytes per pixel) + the 2 x 2-block compositor (ph_compose_up_write_v210): 2.25 texel loads per layer and pixel
    rgb2 = [[torch.empty(sw * sh * 3, dtype=torch.float32, device="cuda") for _ in range(2)] for _ in range(4)]
    up_jobs = [ctx.compose_up_write_v210([(rgb2[l][p], sw, sh, mh) for l in range(4)], out, ow, oh, 0, *wr, rgb=True, prepare_only=True) for p in range(2)]
    vwin2 = [[srcs[k % R][l] for k in range(3)] for l in range(4)]

    def config3_up(i):
        if not (i & 1):
            s = srcs[(i // 2) % R]
            for l in range(4):
                vwin2[l] = [vwin2[l][1], vwin2[l][2], s[l]]
            ctx.v210_yadif_pair([(vwin2[l][0], vwin2[l][1], vwin2[l][2], rgb2[l][0], rgb2[l][1]) for l in range(4)], sw, sh, 1, False, *rd, rgb=True)
        up_jobs[1 ^ (0 if (i & 1) else 1)]()

    # the same with both fields of a frame composited in ONE launch (ph_compose_up_write_v210_pair): two 2160p frames out per launch
    out_b = torch.empty_like(out)
    rgb3 = [[torch.empty(sw * sh * 3, dtype=torch.float32, device="cuda") for _ in range(2)] for _ in range(4)]
    pair_job = ctx.compose_up_write_v210_pair([(rgb3[l][0], sw, sh, mh) for l in range(4)], [(rgb3[l][1], sw, sh, mh) for l in range(4)], out, out_b,
                                              ow, oh, 0, *wr, rgb=True, prepare_only=True)
    vwin3 = [[srcs[k % R][l] for k in range(3)] for l in range(4)]

    def config3_up_pair(i):
        if not (i & 1):  # a frame's two fields: one reader launch, one compositor launch
            s = srcs[(i // 2) % R]
            for l in range(4):
                vwin3[l] = [vwin3[l][1], vwin3[l][2], s[l]]
            ctx.v210_yadif_pair([(vwin3[l][0], vwin3[l][1], vwin3[l][2], rgb3[l][0], rgb3[l][1]) for l in range(4)], sw, sh, 1, False, *rd, rgb=True)
            pair_job()

    algo3 = 4 * 3 * capi.v210_pitch_bytes(sw) * sh + capi.v210_pitch_bytes(ow) * oh  # 88 473 600
    name3 = "3: 1 channel, 4 x 1080i50 -> yadif -> 2x up-scale -> 709->2020 -> combine_4 -> 2160p50 (per output field)"
    current[0] = name3
    record(name3, "fused de-interlacing reader (packed RGB fields) + 2x2-block compositor, two launches per frame: [unpack + yadif, both fields, x4 layers] "
           "(ph_v210_yadif_pair_fmt), [transform x4 + combine_4 + write, both fields] (ph_compose_up_write_v210_pair)", "field",
           timeit(config3_up_pair, reps), algo3, 1.0, {"v210_yadif_pair_kernel": 0.5, "compose_up_write_v210_kernel": 0.5},
           parity_test="tests/test_chains_gpu.py::test_config3_deinterlacing_reader_full_size")
    if routes == "all":
        record(name3, "fused de-interlacing reader (packed RGB fields) + 2x2-block compositor: per frame [unpack + yadif, both fields, x4 layers] "
               "(ph_v210_yadif_pair_fmt), per field [transform x4 + combine_4 + write] (ph_compose_up_write_v210)", "field",
               timeit(config3_up, reps), algo3, 1.5)
        record(name3, "fused de-interlacing reader + fused compositor: per frame [unpack + yadif, both fields, x4 layers] (ph_v210_yadif_pair), "
               "per field [transform x4 + combine_4 + write]", "field", timeit(config3_deint, reps), algo3, 1.5)
        record(name3, "fused compositor, field pairs: per frame [read x4] + yadif_pair x4 (both fields in one pass), per field "
               "[transform x4 + combine_4 + write]", "field", timeit(config3_pair, reps), algo3, 3.5)
        record(name3, "fused compositor: read x4 every other field, yadif x4, [transform x4 + combine_4 + write]", "field",
               timeit(config3_fused, reps), algo3, 7)
    # ---------------- file playback: what the reference's producers really hand over (ffmpegProducer.ts:395-442) ---------------------------------
    # a decoder's planar frames at the clip's own size, the Mixer's default fill; inside ph_chan_compose_v210 such frames are the reader of the
    # format + the 2 x 2-block compositor (DESIGN.md section 5 "Frames of ENLARGED clips")
    fw, fh = 1920, 1080
    frd, fwr = colour("709", "709")
    cm8 = dev(capi.ycbcr2rgb_matrix("709", 8, 16, 235, 224))
    fout = torch.empty(capi.v210_pitch_bytes(fw) * fh // 4, dtype=torch.int32, device="cuda")

    def yuv420p(w_, h_):
        pitch = (w_ + 7) // 8 * 8
        return tuple(torch.randint(0, 256, (k,), dtype=torch.uint8, device="cuda") for k in (pitch * h_, pitch * h_ // 4, pitch * h_ // 4))
    for cw, ch, what in ((1920, 1080, "f1: 1 channel, one 1080p yuv420p file clip under the default fill -> 1080p50 v210"),
                         (1280, 720, "f2: 1 channel, one 720p yuv420p file clip filling a 1080p50 channel -> v210")):
        clips = [yuv420p(cw, ch) for _ in range(R)]
        fm = capi.transform_matrix(fw, fh)
        jobs = [ctx.chan_compose_v210([dict(src=(c, cw, ch, fm, "yuv420p", cm8))], fout, fw, fh, 0, *frd, *fwr, prepare_only=True) for c in clips]
        torch.cuda.synchronize()
        algo_f = cw * ch * 3 // 2 + capi.v210_pitch_bytes(fw) * fh
        current[0] = what
        record(what, "ph_chan_compose_v210 on the decoder's planes: inside, ONE launch - a workgroup converts the source pixels under its tile of the frame once each (the yuv420p reader) "
               "into a scratch rectangle that stays in L2, swaps the writer's table in and composes from there (clip_up_write_v210_kernel)", "frame",
               timeit(lambda i: jobs[i % R](), reps), algo_f, 1, {"clip_up_write_v210_kernel": 1.0},
               parity_test="tests/test_fullsize_gpu.py::test_file_playback_as_benched")
    # config 3 in the reference's own formats: 4 x 1080i50 -> yadif -> own size on a 1080p50 channel (src/config.ts:43-78), per output field
    isrc = [v210(fw, fh, 4) for _ in range(R)]
    iwin = [[isrc[k % R][l] for k in range(3)] for l in range(4)]
    irgb = [[torch.empty(fw * fh * 3, dtype=torch.float32, device="cuda") for _ in range(2)] for _ in range(4)]
    iouts = [torch.empty(capi.v210_pitch_bytes(fw) * fh // 4, dtype=torch.int32, device="cuda") for _ in range(2)]
    ifill = capi.transform_matrix(fw, fh)
    ipair = ctx.compose_up_write_v210_pair([(irgb[l][0], fw, fh, ifill) for l in range(4)], [(irgb[l][1], fw, fh, ifill) for l in range(4)], iouts[0], iouts[1], fw, fh, 0, *fwr,
                                           rgb=True, prepare_only=True)
    torch.cuda.synchronize()

    def config3b(i):
        if not (i & 1):
            s = isrc[(i // 2) % R]
            for l in range(4):
                iwin[l] = [iwin[l][1], iwin[l][2], s[l]]
            ctx.v210_yadif_pair([(iwin[l][0], iwin[l][1], iwin[l][2], irgb[l][0], irgb[l][1]) for l in range(4)], fw, fh, 1, False, *frd, rgb=True)
            ipair()
    current[0] = "f3"
    record("f3: 1 channel, 4 x 1080i50 -> yadif -> own size -> combine_4 -> 1080p50 v210 (per output field)",
           "fused de-interlacing reader (packed RGB fields) + 2x2-block compositor under the default fill, two launches per frame", "field",
           timeit(config3b, reps), 4 * 3 * capi.v210_pitch_bytes(fw) * fh // 2 + capi.v210_pitch_bytes(fw) * fh, 1.0,
           {"v210_yadif_pair_kernel": 0.5, "compose_up_write_v210_kernel": 0.5},
           parity_test="tests/test_fullsize_gpu.py::test_interlaced_sources_on_a_1080p_channel_as_benched")
    current[0] = name3
    if routes == "all":
        up = img(ow, oh, 4)
        comb3 = img(ow, oh)[0]

        def config3(i):
            parity = new_frames(i)
            for l in range(4):
                ctx.yadif(win[l][0], win[l][1], win[l][2], deint[l], sw, sh, parity, 1, False)
                ctx.transform(deint[l], sw, sh, m, up[l], ow, oh)
            ctx.combine(up, comb3, ow, oh)
            ctx.v210_write(comb3, out, ow, oh, 0, *wr)

        record(name3, "one kernel per operator (the reference's job batch)", "field", timeit(config3, reps), algo3, 12)
    return out_records


def main():
    import numpy as np
    import torch
    from phaneron_amd import capi
    ctx = capi.Context(0)
    best = "--best" in sys.argv  # the fastest route of each config only (bench.py's profiled child runs)
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 200
    for r in measure(ctx, torch, np, capi, "best" if best else "all", reps):
        print(json.dumps(r), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
