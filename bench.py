#!/usr/bin/env python3
"""bench.py - frames/sec of the 4-layer 2160p composite hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]

One step = one output frame of the headline pipeline (BASELINE.json / SURVEY.md 8d):
4 x 2160p v210 layers -> unpack -> YCbCr->RGB (BT.709 matrix, gamma LUT, 709->2020 gamut) ->
combine_4 -> RGB->YCbCr (BT.2020) -> v210 pack, executed by ONE fused HIP kernel through the
C ABI (ph_fused_v210_combine).  Inputs are resident in HBM before the timed region; a ring of
distinct frame sets (> 256 MiB in total) defeats the Infinity Cache.  Channels are independent
(SURVEY.md 8e): with N GPUs every rank runs its own channel, no collective on the data path,
scaling = weak, value = total frames/sec over all ranks.

Prints ONE JSON line (rank 0).  The CPU baseline leg (rank 0, N = 1) times the oracle's
restatement of the same pipeline on the host cores - a reported baseline, never the product.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WIDTH, HEIGHT, LAYERS = 3840, 2160, 4
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--ring", type=int, default=8, help="distinct frame sets cycled through (cache defeat)")
    ap.add_argument("--content", choices=["noise", "picture"], default="noise",
                    help="synthetic frames: independent uniform code values per sample (default; the worst case for the gamma "
                         "tables in LDS: every lane of a wave hits a random bank) or picture-like (smooth gradients + a few "
                         "code values of noise: neighbouring pixels look up neighbouring table entries)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--channels", type=int, default=1,
                    help="channels per GPU, composited in ONE batched launch per step (default 1: the headline)")
    ap.add_argument("--frames-per-launch", type=int, default=1,
                    help="successive frames of the ONE channel composited per launch (a play-out that decodes ahead; default 1: every "
                         "frame its own launch).  A 1080p frame is 23 us of which launch and two table loads are ~8: K frames share them")
    ap.add_argument("--layers", type=int, default=LAYERS)
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--no-picture", action="store_true", help="skip the picture-content figure beside the headline (the `content_picture` field)")
    ap.add_argument("--no-secondary", action="store_true", help="skip BASELINE configs 2 and 3 (the `secondary` field)")
    ap.add_argument("--no-route", action="store_true", help="N > 1: skip BASELINE config 5 (the `route` field)")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two short rocprofv3 --pmc child runs); report the recorded figure")
    ap.add_argument("--plan", action="store_true",
                    help="print how --gpus N would be launched (one JSON line) and exit; needs no GPU")
    return ap.parse_args()


# Issue-rate ceiling of the VALU: one wave64 instruction per SIMD every 2 cycles at 2.4 GHz on 1024 SIMDs
# (MI355X_MICROARCH.md; tools/opbench3.hip sustains one per 2.2 - 2.4 cycles at 1.9 - 2.4 GHz).  The headline kernel's
# instruction count per launch comes from the committed SQ_INSTS_VALU pass (profiles/): it is a property of the binary.
VALU_PEAK_WAVE_INSTR_PER_S = 1024 * 2.4e9 / 2.0

# untimed launches before --warmup is honoured: a 20-step run must still measure settled clocks.  The chip needs
# ~50 ms of load to settle: after 200 launches (11 ms) the next 20 still ran 9 % slow, after 1000 they run at the
# steady 54.6 us (profiles/r02_bench_repeat.jsonl and DESIGN.md 5).  Round 6: 20 000 (one second).  The contract's sync between
# the warm-up and the timed steps idles the chip for a moment, and after only 50 ms of load its power state falls back
# far enough for the first of 20 timed launches to run slow (20 timed steps: 50.9 us per launch after 1 000, 48.7 - 49.7
# after 40 000; 2 000 timed steps: 49.6 - 49.9 either way - gpurun_out/r06).  `fixed_warmup` is in the line.
WATCHDOG_EXIT = 3  # exit code of a job whose config-5 leg was abandoned (the partial line is still printed)

FIXED_WARMUP = int(os.environ.get("PH_BENCH_FIXED_WARMUP", "20000"))


def launch_plan(args, argv):
    """How this invocation runs.  Started by torchrun (WORLD_SIZE set): one rank per GPU, as told.  Started bare with
    --gpus N > 1: re-launch itself as N ranks of ONE node through torch.distributed.run (one process per GPU,
    RCCL rendezvous on 127.0.0.1) - the contract `python bench.py --gpus N` must honour on its own."""
    if "WORLD_SIZE" in os.environ:
        world = int(os.environ["WORLD_SIZE"])
        return {"mode": "rank", "world": world, "rank": int(os.environ.get("RANK", "0")),
                "gpus_flag_matches": args.gpus in (1, world)}
    if args.gpus <= 1:
        return {"mode": "single", "world": 1}
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + [a for a in argv if a != "--plan"]
    return {"mode": "spawn", "world": args.gpus, "ranks": list(range(args.gpus)), "cmd": cmd}


def synth_v210(torch, width, height, seed, device, content="noise"):
    """Legal-range v210 frame generated on the GPU (Y 64..940, C 64..960): uniform noise, or a picture-like frame
    (diagonal luma gradient, chroma drifting down the frame, +-6 code values of noise)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    groups = height * (width // 6)
    if content == "picture":
        gx = torch.arange(width // 6, device=device, dtype=torch.float32).repeat(height)          # group column
        gy = torch.arange(height, device=device, dtype=torch.float32).repeat_interleave(width // 6)  # line
        ph = float(seed % 97) / 97.0
        px = gx[:, None] * 6 + torch.arange(6, device=device, dtype=torch.float32)[None, :]
        luma = 64 + 876 * (0.5 + 0.5 * torch.sin(6.2832 * (px / width * 1.5 + gy[:, None] / height + ph)))
        chb = 512 + 380 * torch.sin(6.2832 * (gy / height * 0.7 + ph))[:, None].expand(groups, 3)
        chr_ = 512 + 380 * torch.cos(6.2832 * (gx / (width // 6) * 0.9 + ph))[:, None].expand(groups, 3)
        n = lambda k: torch.randint(-6, 7, (groups, k), generator=g, device=device, dtype=torch.int32)
        y = (luma.to(torch.int32) + n(6)).clamp(64, 940)
        cb = (chb.to(torch.int32) + n(3)).clamp(64, 960)
        cr = (chr_.to(torch.int32) + n(3)).clamp(64, 960)
    else:
        y = torch.randint(64, 941, (groups, 6), generator=g, device=device, dtype=torch.int32)
        cb = torch.randint(64, 961, (groups, 3), generator=g, device=device, dtype=torch.int32)
        cr = torch.randint(64, 961, (groups, 3), generator=g, device=device, dtype=torch.int32)
    w = torch.empty((groups, 4), dtype=torch.int32, device=device)
    w[:, 0] = (cr[:, 0] << 20) | (y[:, 0] << 10) | cb[:, 0]
    w[:, 1] = (y[:, 2] << 20) | (cb[:, 1] << 10) | y[:, 1]
    w[:, 2] = (cb[:, 2] << 20) | (y[:, 3] << 10) | cr[:, 1]
    w[:, 3] = (y[:, 5] << 20) | (cr[:, 2] << 10) | y[:, 4]
    return w.reshape(-1).contiguous()


def cpu_baseline(args, budget_s):
    """The same pipeline (read x4 -> combine_4 -> write) on the host cores, on a bounded sample:
    whole frames until the budget is used.  kind "reference" = the reference's own kernel text
    compiled for x86 (oracle/_ref, built in the build container and shipped prebuilt), its work
    groups spread over the host threads; kind "port" = the oracle's restatement, if that build
    is absent or the host CPU lacks AVX2/FMA."""
    import numpy as np
    import frames
    from oracle import orc
    w, h, n = args.width, args.height, args.layers
    layers = [frames.v210_random(w, h, frames.layer_seed(0, i)) for i in range(n)]
    rd = (orc.ycbcr2rgb_matrix("709"), orc.gamma2linear_lut("709"), orc.rgb2rgb_matrix("709", "2020"))
    wr = (orc.rgb2ycbcr_matrix("2020"), orc.linear2gamma_lut("2020"))
    scratch = np.empty((n + 1) * w * h * 4, np.float32)
    cores = orc.effective_cpus()  # cgroup quota, not the 256 logical CPUs the box reports
    if orc.have_ref_fast():
        r = orc.ref_fast()
        set_threads = r.ref_set_num_threads
        set_threads(cores)
        kind, what = "reference", "the reference's OpenCL kernels compiled for x86 (oracle/_ref, -O3 -mavx2 -mfma; " \
                                  "read x%d, combine_%d, write as separate passes; work groups over %d threads)" % (n, n, cores)
        run = lambda: orc.ref_pipeline_v210_combine(r, layers, w, h, *rd, *wr, scratch=scratch)
    else:
        set_threads = orc.set_num_threads
        set_threads(cores)
        kind, what = "port", "oracle/ restatement (OpenMP over lines, %d threads)" % cores
        run = lambda: orc.pipeline_v210_combine(layers, w, h, *rd, *wr, scratch=scratch)
    run()  # warm-up (page faults)
    t0 = time.perf_counter()
    frames_done = 0
    while True:
        run()
        frames_done += 1
        el = time.perf_counter() - t0
        if el >= budget_s or frames_done >= 200:
            break
    # the same code on ONE host thread (SURVEY 8d asks for both): two frames
    set_threads(1)
    t1 = time.perf_counter()
    run(), run()
    one = 2.0 / (time.perf_counter() - t1)
    set_threads(cores)
    return {"value": round(frames_done / el, 3), "unit": "frames/sec", "cores": cores, "kind": kind,
            "one_core_value": round(one, 3),
            "sample": "%d whole %dx%d %d-layer frames in %.1f s through %s; one_core_value = 2 more frames on 1 thread"
                      % (frames_done, w, h, n, el, what)}


def recorded_valu_instructions():
    """wave64 VALU instructions per launch of the fused kernel (rocprofv3 SQ_INSTS_VALU pass), if recorded"""
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq.txt")), reverse=True):
        for line in open(p):
            f = line.split()
            if len(f) >= 2 and f[0] == "SQ_INSTS_VALU":
                return float(f[1]), os.path.basename(p)
    return None, None


def recorded_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), if present."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get("fused_v210_combine_4_2160p_bytes_per_launch")
        except Exception:
            return None
    return None


# gfx950 corrections of the two counters (MI355X_MICROARCH.md "HBM": FETCH_SIZE reports half the bytes of a wide coalesced
# streaming read; profiles/pmc_traffic.json calibrated both on a 1 GiB copy: 1.99996 and 1.00000).  Counter unit: KiB.
FETCH_CORRECTION, WRITE_CORRECTION = 2.0, 1.0


def measured_traffic(kernel_substring="fused_v210_combine", timeout_s=150):
    """HBM bytes (and VALU instructions) per launch of the headline kernel measured NOW: this script is run three more times,
    briefly, as a child of `rocprofv3 --kernel-trace --pmc <counter>` (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU: one counter per pass;
    --pmc goes with --kernel-trace only), and the per-dispatch counter values of the kernel are averaged.  Returns (bytes, detail) or
    (None, reason) - no rocprofv3 on PATH, a failing pass, nothing parsed."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 is not on PATH"
    means = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
        out = tempfile.mkdtemp(prefix="ph_bench_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable,
               os.path.abspath(__file__), "--steps", "24", "--warmup", "4", "--cpu-seconds", "0", "--no-secondary", "--no-traffic", "--no-picture"]
        env = dict(os.environ, PH_BENCH_FIXED_WARMUP="12", TMPDIR="/tmp")
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "PH_BENCH_FORCE_DIST"):
            env.pop(k, None)
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            vals = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == counter and kernel_substring in row.get("Kernel_Name", ""):
                        vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, "%s pass: no dispatch of the kernel in the counter file (exit %d)" % (counter, r.returncode)
            means[counter] = (sum(vals) / len(vals), len(vals))
        except Exception as e:
            return None, "%s pass failed: %s: %s" % (counter, type(e).__name__, e)
        finally:
            shutil.rmtree(out, ignore_errors=True)
    total = int(round((means["FETCH_SIZE"][0] * FETCH_CORRECTION + means["WRITE_SIZE"][0] * WRITE_CORRECTION) * 1024))
    return total, {"FETCH_SIZE_KiB_mean": round(means["FETCH_SIZE"][0], 1), "WRITE_SIZE_KiB_mean": round(means["WRITE_SIZE"][0], 1),
                   "dispatches": means["FETCH_SIZE"][1], "fetch_correction": FETCH_CORRECTION, "write_correction": WRITE_CORRECTION,
                   "SQ_INSTS_VALU_mean": round(means["SQ_INSTS_VALU"][0], 1)}


class ClockSampler:
    """The shader clock the SMU reports while the bench runs (sysfs pp_dpm_sclk of this GPU: the line marked '*' is the current
    frequency), sampled by a thread every ~2 ms from the second half of the untimed warm-up to the end of the timed steps.  Boxes
    differ by +-8 % in what they sustain under this kernel's VALU load: the line says which clock its numbers were measured at."""

    def __init__(self, torch, device_index):
        import glob
        self.path, self.samples, self.stop, self.thread = None, [], False, None
        try:
            bus = torch.cuda.get_device_properties(device_index).pci_bus_id
            dev = torch.cuda.get_device_properties(device_index).pci_device_id
            dom = getattr(torch.cuda.get_device_properties(device_index), "pci_domain_id", 0)
            want = "%04x:%02x:%02x.0" % (dom, bus, dev)
            for c in glob.glob("/sys/class/drm/card*/device"):
                if os.path.basename(os.path.realpath(c)) == want and os.path.exists(os.path.join(c, "pp_dpm_sclk")):
                    self.path = os.path.join(c, "pp_dpm_sclk")
        except Exception:
            self.path = None

    def _read(self):
        try:
            for line in open(self.path):
                if "*" in line:
                    return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except Exception:
            return None
        return None

    def start(self):
        if not self.path:
            return
        import threading

        def run():
            while not self.stop:
                v = self._read()
                if v:
                    self.samples.append((time.perf_counter(), v))
                time.sleep(0.002)
        self.thread = threading.Thread(target=run, daemon=True)
        self.thread.start()

    def finish(self, since):
        self.stop = True
        if self.thread:
            self.thread.join(timeout=1.0)
        v = [x for t, x in self.samples if t >= since]
        if not v:
            return {"mhz": None, "why": "no sysfs pp_dpm_sclk for this GPU" if not self.path else "no sample fell into the window"}
        return {"mhz_mean": round(sum(v) / len(v), 1), "mhz_min": min(v), "mhz_max": max(v), "samples": len(v),
                "source": "sysfs pp_dpm_sclk ('*' line) sampled every ~2 ms from the second half of the untimed warm-up to the end of the timed steps"}


def secondary_counters(timeout_s=240):
    """Counters of the secondary workloads' kernels, measured NOW: tools/config_bench.py --best is run again, briefly, config group by config group (PH_CONFIG_BENCH_ONLY), as a
    child of `rocprofv3 --kernel-trace --pmc <counter>` (one counter per pass: SQ_INSTS_VALU, FETCH_SIZE, WRITE_SIZE), and each
    kernel's per-dispatch values are averaged.  Returns ({counter: {kernel name: mean}}, None) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 is not on PATH"
    got = {}
    # the configs go through the passes in groups whose kernels do not share names: the file-playback entries (f1 - f3) launch the 2 x 2-block
    # compositor at three more shapes, and a per-name average over all of them would belong to none.  A group's counters are kept under its prefix
    for group, counter in [(g, c) for g in ("2,7,3", "f1", "f2", "f3") for c in ("SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE")]:
        out = tempfile.mkdtemp(prefix="ph_bench_pmc2_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable,
               os.path.join(ROOT, "tools", "config_bench.py"), "--best", "--reps", "12"]
        env = dict(os.environ, TMPDIR="/tmp", PH_CONFIG_BENCH_WARM_S="0.02", PH_CONFIG_BENCH_ONLY=group)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "PH_BENCH_FORCE_DIST"):
            env.pop(k, None)
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            vals = {}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == counter:
                        vals.setdefault(row.get("Kernel_Name", ""), []).append(float(row["Counter_Value"]))
            if not vals:
                return None, "%s pass of group %s: no counter values (exit %d)" % (counter, group, r.returncode)
            got.setdefault(group, {})[counter] = {k: sum(v) / len(v) for k, v in vals.items()}
        except Exception as e:
            return None, "%s pass failed: %s: %s" % (counter, type(e).__name__, e)
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return got, None


def name_binding_resources(records, counters):
    """Every secondary record's roofline gets what binds it: `valu` (wave64 instructions per unit from SQ_INSTS_VALU of its kernels,
    their rate against the issue peak) and `traffic` (HBM bytes per unit, FETCH_SIZE doubled as the guide prescribes) - so that a
    reader can recompute every fraction from the line alone and see that these kernels sit on VALU issue, not on HBM."""
    for rec in records:
        kl = rec.get("kernel_launches")
        if not kl or "roofline" not in rec:
            continue
        ms = [v for k, v in rec.items() if k.startswith("ms_per_")][0]
        per_unit = {}
        group = [g for g in counters if any(rec["config"].startswith(x) for x in g.split(","))]
        for counter, by_kernel in (counters[group[0]] if group else {}).items():
            total, found = 0.0, True
            for sub, n in kl.items():
                hit = [v for k, v in by_kernel.items() if sub in k]
                if not hit:
                    found = False
                    break
                total += n * sum(hit) / len(hit)
            per_unit[counter] = total if found else None
        rf = rec["roofline"]
        if per_unit.get("SQ_INSTS_VALU"):
            rate = per_unit["SQ_INSTS_VALU"] / (ms * 1e-3)
            rf["valu"] = {"instructions_per_unit": round(per_unit["SQ_INSTS_VALU"]), "achieved": round(rate / 1e12, 4),
                          "peak": round(VALU_PEAK_WAVE_INSTR_PER_S / 1e12, 4), "unit": "T wave64-instr/s", "frac": round(rate / VALU_PEAK_WAVE_INSTR_PER_S, 4)}
        if per_unit.get("FETCH_SIZE") is not None and per_unit.get("WRITE_SIZE") is not None:
            rf["traffic"] = int(round((per_unit["FETCH_SIZE"] * FETCH_CORRECTION + per_unit["WRITE_SIZE"] * WRITE_CORRECTION) * 1024))
            rf["traffic_over_algorithmic"] = round(rf["traffic"] / rec["algorithmic_bytes"], 3)
            if rec.get("bytes_as_benched"):
                rf["traffic_over_bytes_as_benched"] = round(rf["traffic"] / rec["bytes_as_benched"], 3)
        if "valu" in rf:
            rf["binding_resource"] = "valu issue" if rf["valu"]["frac"] > rf["frac"] else "hbm"
            rf["counters_source"] = "measured in this run: rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU | FETCH_SIZE | WRITE_SIZE (separate child passes of tools/config_bench.py --best)"


def main():
    args = parse()
    plan = launch_plan(args, sys.argv[1:])
    if args.plan:
        print(json.dumps(plan), flush=True)
        return 0
    if plan["mode"] == "rank" and not plan["gpus_flag_matches"]:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, plan["world"]))
    import torch
    if plan["mode"] == "spawn":
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit("bench.py: --gpus %d but only %d GPU(s) are visible" % (args.gpus, have))
        import subprocess
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        return subprocess.run(plan["cmd"], env=env).returncode  # rank 0 of the children prints the JSON line
    from phaneron_amd import capi
    import numpy as np

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libphaneron_hip has no CPU path")
    # test hooks (tests/test_bench_gpu.py; never set by the driver): PH_BENCH_SHARE_GPU=1 lets several ranks share the
    # one GPU of a test box (gloo process group: RCCL refuses two ranks on one device), PH_BENCH_FORCE_DIST=1 creates the
    # RCCL process group even for a single rank, so that the N > 1 code path runs on one GPU.
    share_gpu = os.environ.get("PH_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU %d (%d visible)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("PH_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)
    reduce_device = torch.device("cpu") if share_gpu else device

    w, h, n = args.width, args.height, args.layers
    ctx = capi.Context(local_rank)
    frame_words = capi.v210_pitch_bytes(w) * h // 4

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(device)

    rd = [dev(capi.ycbcr2rgb_matrix("709")), dev(capi.gamma2linear_lut("709")),
          dev(np.concatenate([capi.rgb2rgb_matrix("709", "2020"), np.zeros(3, np.float32)]))]
    wr = [dev(capi.rgb2ycbcr_matrix("2020")), dev(capi.linear2gamma_lut("2020"))]
    torch.cuda.synchronize()
    ctx.register_lut(rd[1], capi.gamma2linear_lut("709"))   # exact LDS form of the tables
    ctx.register_lut(wr[1], capi.linear2gamma_lut("2020"))
    if os.environ.get("PH_BENCH_GLOBAL_LUT"):
        ctx.set_option("lds_lut", 0)
    K = max(1, args.frames_per_launch)
    if K > 1 and args.channels > 1:
        raise SystemExit("bench.py: --frames-per-launch and --channels are two uses of the one batched launch; pick one")
    C = max(1, args.channels) if K == 1 else K  # jobs of one launch: C channels' frames, or K successive frames of one channel
    ring = []  # per slot: C channels' layer lists and outputs
    for r in range(args.ring):
        ins = [[synth_v210(torch, w, h, 0x5EED0000 + 16 * ((rank * C + c) * 64 + r) + l, device, args.content) for l in range(n)]
               for c in range(C)]
        ring.append((ins, [torch.empty(frame_words, dtype=torch.int32, device=device) for _ in range(C)]))
    torch.cuda.synchronize()

    stream = ctx.torch_stream(capi.QUEUE_PROCESS)

    # (the ctypes marshalling of a frame set's pointers is done once per ring slot, not per step: the binding's Python is not the product)
    jobs = [ctx.fused_v210_combine(ins[0], outs[0], w, h, *rd, *wr, prepare_only=True) for ins, outs in ring] if C == 1 else None

    def step(i):
        ins, outs = ring[i % args.ring]
        if C == 1:
            jobs[i % args.ring]()
        else:  # one batched launch for the GPU's channels (ph_fused_v210_combine_batch)
            ctx.fused_v210_combine_batch(ins, outs, w, h, *rd, *wr)

    from phaneron_amd import multigpu

    def sync():
        # The library's queue is polled (hipStreamQuery) until it is idle, THEN the blocking waits are made: a thread blocked in
        # hipStreamSynchronize is woken tens of microseconds after the last kernel has ended, and the driver's 20 timed steps are
        # one millisecond in all - the wake-up alone read as 5 % of `value` (BENCH_r03: 18 382 against 20 000 with 2000 steps).
        # The same work is waited for; only the host's reaction time goes out of the timed region.
        spin_until = time.perf_counter() + 5.0
        while not ctx.queue_idle() and time.perf_counter() < spin_until:
            pass
        ctx.wait()
        torch.cuda.synchronize()

    # HIP events on the library's own stream bracket the same K launches the wall clock times
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed_step(i):
        if i == args.warmup:
            ev0.record(stream)
        step(i)
        if i == args.warmup + args.steps - 1:
            ev1.record(stream)

    clock = ClockSampler(torch, local_rank)
    clock.start()
    clock_since = time.perf_counter()
    for i in range(FIXED_WARMUP):  # clocks, caches and the allocator settle before the contract's own warm-up
        step(i)
        if i == FIXED_WARMUP // 2:
            clock_since = time.perf_counter()
    sync()
    timed_local = {}
    elapsed = multigpu.timed_steps(timed_step, args.steps, args.warmup, sync, dist, reduce_device, local=timed_local)
    kernel_ms = ev0.elapsed_time(ev1) / max(args.steps, 1)  # average launch duration on the kernel's stream
    shader_clock = clock.finish(clock_since)
    # frames every rank really composited in the timed region, summed over ranks (not assumed equal)
    frames_done = C * args.steps
    per_rank = None
    if dist is not None:
        t = torch.tensor([frames_done], dtype=torch.int64, device=reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        frames_done = int(t.item())
        # every rank's own figures, so that the first run on N devices says which rank was slow and why: its own wall
        # clock over the timed steps (timed_steps keeps it), its frames, its kernel's average launch time
        mine = torch.tensor([C * args.steps, timed_local["elapsed"], kernel_ms], dtype=torch.float64, device=reduce_device)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [[float(x) for x in e.tolist()] for e in every]

    lds = ctx.lut_info(rd[1])["lds_bytes"] and ctx.lut_info(wr[1])["lds_bytes"] and not os.environ.get("PH_BENCH_GLOBAL_LUT")
    kernel_name = ("fused_v210_combine_lds_kernel<%d,...>" if lds else "fused_v210_combine_kernel<%d>") % n

    def report(route_rec, minimal=False):
        """rank 0 prints the one JSON line; `minimal` (the route watchdog's call) leaves out everything that takes time"""
        if rank == 0:
            fps = frames_done / elapsed
            algo_bytes = C * (n + 1) * frame_words * 4  # each input byte once + each output byte once
            achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
            line = {
                "metric": "frames/sec, 4-layer 2160p50 composite pipeline (v210 unpack->CSC->combine->CSC->v210 pack)",
                "value": round(fps, 2), "unit": "frames/sec", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "fixed_warmup": FIXED_WARMUP, "ms_per_step": round(1e3 * elapsed / max(args.steps, 1), 5),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                "data": "synthetic",
                "config": {"workload": "%s: %d channel%s per GPU, %d x %dx%d v210 layers -> fused unpack/CSC(709->2020)/"
                                       "combine_%d/CSC/pack -> 1 v210 frame%s"
                                       % ("headline" if C == 1 and (w, h, n) == (WIDTH, HEIGHT, LAYERS) else "variant", 1 if K > 1 else C,
                                          "" if C == 1 or K > 1 else "s", n, w, h, n,
                                          "" if C == 1 else (", %d successive frames per launch" % K if K > 1 else " each, one batched launch per step")),
                           "ring_frame_sets": args.ring, "channels": world * (1 if K > 1 else C), "frames_per_launch": K, "realtime_target_fps": 50,
                           "content": args.content},
                "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": recorded_traffic(),
                             "traffic_source": "recorded: profiles/pmc_traffic.json, rocprofv3 FETCH_SIZE / WRITE_SIZE passes "
                                               "of this command (tools/profile_round.sh), not measured in this run",
                             "kernel": kernel_name, "algorithmic_bytes_per_launch": algo_bytes,
                             "avg_launch_ms": round(kernel_ms, 5)},
            }
            # the same fraction on the OTHER clock (SURVEY 8d's formula): algorithmic bytes x the frames/s of `value` per GPU.
            # `frac` is the kernel's own time (HIP events on its stream); `value` is wall clock over the K steps including the
            # closing ctx.wait() + device sync, which a short run does not amortise - the difference is stated, not hidden
            by_value = algo_bytes / C * (fps / world) / 1e9
            line["roofline"]["frac_by_value"] = round(by_value / HBM_PEAK_GBS, 4)
            line["roofline"]["clocks"] = ("frac: HIP events around the timed launches on the kernel's stream (avg_launch_ms); "
                                          "frac_by_value: algorithmic bytes x value / n_gpus (host wall clock, barrier + sync on both sides)")
            line["roofline"]["shader_clock"] = shader_clock
            line["roofline"]["sync_overhead_us_per_step"] = round(1e3 * (1e3 * elapsed / max(args.steps, 1) - kernel_ms), 3)
            if args.steps < 200:
                line["roofline"]["sync_overhead_note"] = ("%d timed steps: the closing sync (a few hundred microseconds of host time, once) is "
                                                          "spread over few steps; with the default 2000 steps both fractions agree" % args.steps)
            if per_rank is not None:
                rates = [f / e for f, e, _ in per_rank]
                slow = min(range(len(rates)), key=lambda i: rates[i])
                line["per_rank"] = {"frames_per_sec": [round(x, 2) for x in rates], "min": round(min(rates), 2), "max": round(max(rates), 2),
                                    "elapsed_s": [round(e, 6) for _, e, _ in per_rank],
                                    "avg_launch_ms": [round(k, 5) for _, _, k in per_rank],
                                    "slowest_rank": slow, "slowest_rank_avg_launch_ms": round(per_rank[slow][2], 5),
                                    "roofline_frac": [round(algo_bytes / (k * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k > 0 else None for _, _, k in per_rank]}
            if route_rec is not None:
                line["route"] = route_rec
            headline = C == 1 and (w, h, n) == (WIDTH, HEIGHT, LAYERS)

            def put_valu(insts, source):
                rate = insts / (kernel_ms * 1e-3)
                line["roofline"]["valu"] = {"achieved": round(rate / 1e12, 4), "peak": round(VALU_PEAK_WAVE_INSTR_PER_S / 1e12, 4),
                                            "unit": "T wave64-instr/s", "frac": round(rate / VALU_PEAK_WAVE_INSTR_PER_S, 4),
                                            "instructions_per_launch": insts, "source": source}
                put_ceiling()

            def put_ceiling():
                # north_star asks for 0.80 of the HBM roofline; the reference's arithmetic (15 gamma lookups and ~240 VALU lane-instructions
                # per output pixel) puts the kernel on VALU issue, not on HBM.  What that allows, as HBM fractions of this very kernel:
                # its instruction count at the issue peak, and at the rate the chip sustains under this instruction mix (tools/opbench3:
                # 2.2 - 2.4 cycles per wave instruction at 1.9 - 2.1 GHz under full VALU load -> ~1.02 - 1.26 ns)
                insts_l = line["roofline"]["valu"]["instructions_per_launch"]
                at_peak_ms = insts_l / VALU_PEAK_WAVE_INSTR_PER_S * 1e3
                sustained_ms = [insts_l / 1024.0 * ns * 1e-9 * 1e3 for ns in (1.26, 1.02)]
                frac_at = lambda ms_: round(algo_bytes / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)
                line["roofline"]["binding_resource"] = "valu issue (with the LDS gather pipe beside it)"
                line["roofline"]["ceiling"] = {
                    "hbm_frac_at_valu_issue_peak": frac_at(at_peak_ms),
                    "hbm_frac_at_sustained_issue_rate": [frac_at(sustained_ms[0]), frac_at(sustained_ms[1])],
                    "note": "the HBM fraction this kernel's %.1f M wave instructions per launch allow: at one instruction per SIMD every 2 cycles at "
                            "2.4 GHz (%.1f us), and at the 1.26 .. 1.02 ns per instruction the chip sustains under a full VALU load (%.1f .. %.1f us); "
                            "north_star's 0.80 would need the reference's per-pixel arithmetic to be 3x cheaper than one instruction per operation "
                            "(DESIGN.md 4)" % (insts_l / 1e6, at_peak_ms * 1e3, sustained_ms[0] * 1e3, sustained_ms[1] * 1e3)}

            insts, src = recorded_valu_instructions()
            if insts and headline:
                put_valu(insts, "recorded: profiles/" + src)
            if not minimal and world == 1 and headline and not args.no_traffic and os.environ.get("PH_BENCH_TRAFFIC", "1") != "0":
                got, detail = measured_traffic()
                if got is not None:
                    line["roofline"]["traffic"] = got
                    line["roofline"]["traffic_source"] = ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE "
                                                          "(separate child passes of this script), gfx950 corrections applied")
                    line["roofline"]["traffic_detail"] = detail
                    line["roofline"]["traffic_over_algorithmic"] = round(got / algo_bytes, 4)
                    if detail.get("SQ_INSTS_VALU_mean"):  # the instruction count of THIS binary in THIS run replaces the recorded one
                        put_valu(detail["SQ_INSTS_VALU_mean"], "measured in this run: rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU (child pass of this script)")
                else:
                    line["roofline"]["traffic_not_measured"] = detail
            if not minimal and world == 1 and headline and args.content == "noise" and not args.no_picture:
                # `value` is measured on independent uniform code values per sample - the worst case for tables in LDS (the 64 lanes of a wave hit
                # random banks: two thirds of the LDS cycles are bank conflicts).  The same kernel on picture-like frames (gradients + a few codes
                # of noise: neighbouring pixels look up neighbouring entries), measured here after the timed region, beside it:
                try:
                    pring = []
                    for r in range(2):
                        pins = [synth_v210(torch, w, h, 0x5EED7000 + 16 * r + l, device, "picture") for l in range(n)]
                        pring.append(ctx.fused_v210_combine(pins, torch.empty(frame_words, dtype=torch.int32, device=device), w, h, *rd, *wr, prepare_only=True))
                    torch.cuda.synchronize()
                    for i in range(300):
                        pring[i & 1]()
                    sync()
                    p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    psteps = 1000
                    p0.record(stream)
                    for i in range(psteps):
                        pring[i & 1]()
                    p1.record(stream)
                    sync()
                    pms = p0.elapsed_time(p1) / psteps
                    line["content_picture"] = {"frames_per_sec_by_kernel_time": round(1e3 / pms, 1), "avg_launch_ms": round(pms, 5),
                                               "roofline_frac": round(algo_bytes / (pms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "steps": psteps,
                                               "note": "the headline kernel on picture-like frames (bench.py --content picture), HIP events on its stream; "
                                                       "`value` and `roofline.frac` stay the noise figures (worst case for the LDS tables)"}
                    del pring
                except Exception as e:
                    line["content_picture"] = {"error": "%s: %s" % (type(e).__name__, e)}
            if not minimal and world == 1 and not args.no_secondary and C == 1 and (w, h, n) == (WIDTH, HEIGHT, LAYERS):
                # BASELINE configs 2 and 3 (the compositing configs: real alpha, transforms, de-interlace), fastest route of each,
                # measured after the timed region; tools/config_bench.py prints every route
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import config_bench
                del ring[:]
                if jobs:
                    del jobs[:]
                torch.cuda.empty_cache()
                try:
                    line["secondary"] = config_bench.measure(ctx, torch, np, capi, "best", reps=150)
                    if not args.no_traffic and os.environ.get("PH_BENCH_TRAFFIC", "1") != "0":
                        counters, why = secondary_counters()
                        if counters:
                            name_binding_resources(line["secondary"], counters)
                        else:
                            line["secondary_counters_not_measured"] = why
                except Exception as e:  # the headline figure must not depend on the secondary workloads
                    line["secondary"] = [{"error": "%s: %s" % (type(e).__name__, e)}]
            if not minimal and world == 1 and args.cpu_seconds > 0:
                try:
                    line["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
                except Exception as e:  # report rather than lose the line
                    line["cpu_baseline"] = {"value": None, "unit": "frames/sec", "cores": 0, "kind": "port", "sample": "failed: %s: %s" % (type(e).__name__, e)}
            else:
                line["cpu_baseline"] = None
            print(json.dumps(line), flush=True)

    # N > 1 (or the forced distributed path of the tests): BASELINE config 5 in the same command - 2 channels per rank,
    # every channel's fourth layer routed from channel k + N (ph_route_*: RCCL on its own stream), fingerprint-checked
    route_rec = None
    if dist is not None and not args.no_route and C == 1 and (w, h, n) == (WIDTH, HEIGHT, LAYERS):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import route_bench
        del ring[:]
        if jobs:
            del jobs[:]
        torch.cuda.empty_cache()
        rh = int(os.environ.get("PH_BENCH_ROUTE_HEIGHT", HEIGHT))  # tests shrink the frame
        r_args = route_bench.parse(["--check", "--steps", "40", "--warmup", "5", "--width", str(WIDTH), "--height", str(rh),
                                    "--backend", "gloo" if share_gpu else "nccl"] + (["--loopback"] if world == 1 else []) +
                                   (["--same-gpu"] if share_gpu else []))
        # The hand-off between processes on different GPUs cannot be rehearsed on the one-GPU boxes this is developed on: a
        # watchdog keeps the headline figure if it stalls - rank 0 prints the line with the failure recorded, every rank leaves
        import threading
        limit = float(os.environ.get("PH_BENCH_ROUTE_TIMEOUT", "240"))

        def gave_up():
            report({"error": "config 5 did not finish within %.0f s; abandoned by the watchdog" % limit}, minimal=True)
            sys.stdout.flush()
            os._exit(WATCHDOG_EXIT)  # the line is out; the exit code still tells the launcher that a rank hung
        watchdog = threading.Timer(limit, gave_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            route_rec = route_bench.measure(r_args, ctx, dist, rank, world, device, log=lambda m: print(m, file=sys.stderr, flush=True))
        except Exception as e:  # the headline figure must not depend on it; every rank takes the same path or the job hangs
            route_rec = {"error": "%s: %s" % (type(e).__name__, e)}
        finally:
            watchdog.cancel()

    report(route_rec)
    ctx.close()
    if dist:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
