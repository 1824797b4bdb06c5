"""GPU tests (-m gpu): this repository's pack kernels against the REFERENCE's own kernels running on the same MI355X.

The GPU box's OpenCL runtime lists the MI355X but without image support (profiles/r03_opencl_probe.txt), so of the reference's
path only the buffer kernels - `read` / `write` of the seven pack formats - can execute on it.  Their OpenCL C text is compiled
unmodified for gfx950 with AMD's OpenCL device library in the build container (oracle/refbuild/build_ref_gpu.sh ->
oracle/_ref/refgpu/*.co) and launched here with the geometry the reference's Reader / Writer classes compute (golden trace).
Every output word / float of ph_v210_read / ph_v210_write / ph_pack_read / ph_pack_write must equal theirs."""
import numpy as np
import pytest

import frames
import refgpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not refgpu.available(), reason="oracle/_ref/refgpu is built where the reference checkout exists")]
W, H = 1920, 1080
SPECS = {"v210": ("709", "2020"), "yuv422p10": ("709", "709"), "yuv422p8": ("601-625", "709"), "yuv420p": ("709", "2020"), "nv12": ("709", "709"),
         "rgba8": ("sRGB", "709"), "bgra8": ("sRGB", "sRGB")}


@pytest.fixture(scope="module")
def ref():
    return refgpu.RefGpu()


@pytest.mark.parametrize("fmt", refgpu.FORMATS)
def test_read_equals_the_reference_kernel_on_this_gpu(ref, fmt):
    import torch
    import hip_harness as hh
    spec, ospec = SPECS[fmt]
    (g, wg), = refgpu.geometry()[(fmt, "read")]
    planes = [frames.v210_random(W, H, 4242, legal=False)] if fmt == "v210" else frames.pack_random(fmt, W, H, 4242)
    d_planes = [hh.dev(np.ascontiguousarray(p).view(np.uint8) if fmt != "v210" else p) for p in planes]
    cm, lut, gm = hh.ColourParams.fmt_reader(fmt, spec, ospec)
    want = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ref.launch(fmt, "read", g, wg, d_planes + [want, W] + ([cm] if cm is not None else []) + [lut, gm])
    got = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda")
    k = hh.ctx()
    if fmt == "v210":
        k.v210_read(d_planes[0], got, W, H, cm, lut, gm)
    else:
        k.pack_read(fmt, d_planes, got, W, H, cm, lut, gm)
    a, b = hh.host(got).view(np.uint32), want.cpu().numpy().view(np.uint32)
    assert np.count_nonzero(b) > b.size // 2, "the reference kernel did not run"
    assert np.array_equal(a, b), "%s read: %d of %d floats differ from the reference kernel's" % (fmt, int((a != b).sum()), a.size)


@pytest.mark.parametrize("fmt", refgpu.FORMATS)
def test_write_equals_the_reference_kernel_on_this_gpu(ref, fmt):
    """v210: the progressive writer; the other formats: both fields through their interlaced writer (what the reference's test
    scripts and the golden trace do)"""
    import hip_harness as hh
    spec = SPECS[fmt][1]
    rgba = frames.rgba_random(W, H, 77, -0.05, 1.05)
    d_rgba = hh.dev(rgba)
    cm, lut = hh.ColourParams.fmt_writer(fmt, spec)
    sizes = [frames.v210_pitch_bytes(W) * H] if fmt == "v210" else frames.pack_plane_bytes(fmt, W, H)
    geo = refgpu.geometry()[(fmt, "write")]
    fields = [0] if fmt == "v210" else [1, 3]
    g, wg = geo[0] if fmt == "v210" else geo[-1]
    want = [hh.dev(np.full(n, 0x5A, np.uint8)) for n in sizes]
    got = [hh.dev(np.full(n, 0x5A, np.uint8)) for n in sizes]
    k = hh.ctx()
    for il in fields:
        ref.launch(fmt, "write", g, wg, [d_rgba] + want + [W, il] + ([cm] if cm is not None else []) + [lut])
        if fmt == "v210":
            k.v210_write(d_rgba, got[0], W, H, il, cm, lut)
        else:
            k.pack_write(fmt, d_rgba, got, W, H, il, cm, lut)
    for i, (a, b) in enumerate(zip(got, want)):
        a, b = hh.host(a), b.cpu().numpy()
        assert np.count_nonzero(b != 0x5A) > b.size // 2, "the reference kernel did not run"
        assert np.array_equal(a, b), "%s write plane %d: %d of %d bytes differ from the reference kernel's" % (fmt, i, int((a != b).sum()), a.size)


def test_reference_v210_kernels_timed_on_this_gpu(ref):
    """the reference's v210 kernels and this repository's, both at 3840 x 2160 on this device (HIP events, frame ring); the
    record goes to gpurun_out/ (copied to profiles/r03_ref_on_gpu.jsonl).  Only a sanity bound is asserted."""
    import json
    import os
    import numpy as np
    import torch
    from phaneron_amd import capi
    lines = []
    ctx = capi.Context(0)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    w, h, R, reps = 3840, 2160, 6, 60
    rd = [dev(capi.ycbcr2rgb_matrix("709")), dev(capi.gamma2linear_lut("709")), dev(np.concatenate([capi.rgb2rgb_matrix("709", "2020"), np.zeros(3, np.float32)]))]
    wr = [dev(capi.rgb2ycbcr_matrix("2020")), dev(capi.linear2gamma_lut("2020"))]
    torch.cuda.synchronize()
    ctx.register_lut(rd[1], capi.gamma2linear_lut("709"))
    ctx.register_lut(wr[1], capi.linear2gamma_lut("2020"))
    words = capi.v210_pitch_bytes(w) * h // 4
    v = [torch.randint(0, 2 ** 30, (words,), dtype=torch.int32, device="cuda") for _ in range(R)]
    img = [torch.rand(w * h * 4, device="cuda") for _ in range(R)]
    out_img = [torch.empty(w * h * 4, device="cuda") for _ in range(2)]
    out_v = [torch.empty(words, dtype=torch.int32, device="cuda") for _ in range(2)]
    wipg = capi.v210_pitch_bytes(w) // 128  # pitch / 48 pixels per work-item (v210.ts:303-305)
    stream = ctx.torch_stream()

    def time_ours(fn):
        for i in range(3):
            fn(i)
        ctx.wait()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(reps):
            fn(i)
        e1.record(stream)
        ctx.wait()
        return 1e3 * e0.elapsed_time(e1) / reps

    def time_ref(name, args_of):
        import ctypes as C
        hip = ref.hip
        for i in range(3):
            ref.launch("v210", name, wipg * h, wipg, args_of(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        calls = []
        for i in range(reps):  # marshal first, then launch back to back on the null stream
            vals = [C.c_uint32(a) if isinstance(a, int) else C.c_void_p(a.data_ptr()) for a in args_of(i)]
            calls.append((vals, (C.c_void_p * len(vals))(*[C.cast(C.pointer(x), C.c_void_p) for x in vals])))
        torch.cuda.synchronize()
        e0.record(torch.cuda.default_stream())
        for vals, params in calls:
            hip.hipModuleLaunchKernel(ref.fn[("v210", name)], h, 1, 1, wipg, 1, 1, 0, None, params, None)
        e1.record(torch.cuda.default_stream())
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps
    vb, ib = words * 4, w * h * 16
    for name, ours, theirs in (
            ("v210 read 2160p", lambda i: ctx.v210_read(v[i % R], out_img[i % 2], w, h, *rd), ("read", lambda i: [v[i % R], out_img[i % 2], w, rd[0], rd[1], rd[2]])),
            ("v210 write 2160p", lambda i: ctx.v210_write(img[i % R], out_v[i % 2], w, h, 0, *wr), ("write", lambda i: [img[i % R], out_v[i % 2], w, 0, wr[0], wr[1]]))):
        t_ours, t_ref = time_ours(ours), time_ref(*theirs)
        lines.append(json.dumps({"kernel": name, "bytes": vb + ib, "this_repository_us": round(t_ours, 1), "this_repository_GBps": round((vb + ib) / t_ours / 1e3, 1),
                          "reference_kernel_on_this_gpu_us": round(t_ref, 1), "reference_GBps": round((vb + ib) / t_ref / 1e3, 1),
                          "speedup": round(t_ref / t_ours, 2)}))
        assert t_ours < t_ref, lines[-1]
    ctx.close()
    out_dir = os.path.join(refgpu.ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "r03_ref_on_gpu.jsonl"), "w") as f:
            f.write("\n".join(lines) + "\n")
