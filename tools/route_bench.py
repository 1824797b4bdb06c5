#!/usr/bin/env python3
"""BASELINE config 5: C channels per GPU, every channel's fourth layer is the ROUTE of another
channel's combiner output (f32 RGBA, 132.7 MB at 2160p) - channel k shows channel (k + total/2) mod
total, so with more than one rank every route crosses GPUs (routeProducer.ts:63-126; RCCL send/recv
over xGMI, phaneron_amd/multigpu.py).  Per channel and frame: v210 read x3 -> combine_4 with the routed
frame of the PREVIOUS step (a route is one frame late in the reference too) -> v210 write.

  python tools/route_bench.py                                       # one GPU: both routes are local aliases
  python tools/route_bench.py --loopback --check                    # one GPU, routes through RCCL to the own rank
  torchrun --nproc-per-node 8 tools/route_bench.py                  # 16 channels on 8 GPUs (RCCL over xGMI)
  torchrun --nproc-per-node 2 tools/route_bench.py --backend gloo --same-gpu --check   # functional test on one GPU

With RCCL (backend nccl, or --loopback) the hand-off runs on the library's path (ph_route_*): a communication stream
of its own, event-ordered against the process queue, overlapping the three v210 reads of the receiving channels.
The gloo variant stages through host memory with torch.distributed (functional test only).

Prints one JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels-per-gpu", type=int, default=2)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--same-gpu", action="store_true", help="every rank uses cuda:0 (functional test)")
    ap.add_argument("--check", action="store_true", help="verify (bit for bit) that a routed layer is the source channel's output")
    ap.add_argument("--print-fingerprints", action="store_true", help="print a fingerprint of every channel's final v210 output")
    ap.add_argument("--loopback", action="store_true", help="send same-rank routes through RCCL too (peer = own rank)")
    args = ap.parse_args(argv)
    if args.loopback:  # the one-device rehearsal of the hand-off always verifies what arrived (profiles/r05_route_loopback.jsonl said "not run")
        args.check = True
    return args


def main():
    args = parse()
    import torch
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if args.same_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.backend, **({"device_id": device} if args.backend == "nccl" else {}))
    from phaneron_amd import capi
    ctx = capi.Context(local)
    rec = measure(args, ctx, dist, rank, world, device)
    if rank == 0:
        print(json.dumps(rec), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


def measure(args, ctx, dist, rank, world, device, log=lambda m: print(m, flush=True)):
    """Config 5 on an existing context / process group (bench.py --gpus N calls this for its `route` entry).
    Returns the record (meaningful on rank 0); `log` receives the check / fingerprint lines."""
    import numpy as np
    import torch
    from phaneron_amd import capi, multigpu
    w, h, C = args.width, args.height, args.channels_per_gpu
    total = world * C
    mine = multigpu.channels_of_rank(rank, total, world, C)
    routes = [multigpu.Route(src=(k + total // 2) % total, dst=k) for k in range(total)]
    words, npx = capi.v210_pitch_bytes(w) * h // 4, w * h

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(device)

    rd = [dev(capi.ycbcr2rgb_matrix("709")), dev(capi.gamma2linear_lut("709")),
          dev(np.concatenate([capi.rgb2rgb_matrix("709", "2020"), np.zeros(3, np.float32)]))]
    wr = [dev(capi.rgb2ycbcr_matrix("2020")), dev(capi.linear2gamma_lut("2020"))]
    torch.cuda.synchronize()
    ctx.register_lut(rd[1], capi.gamma2linear_lut("709"))
    ctx.register_lut(wr[1], capi.linear2gamma_lut("2020"))
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    chan = {}
    for ch in mine:
        chan[ch] = dict(
            layers=[torch.randint(0, 2 ** 30, (words,), dtype=torch.int32, device=device, generator=g) for _ in range(3)],
            rgba=[torch.empty(npx * 4, dtype=torch.float32, device=device) for _ in range(3)],
            out=[torch.zeros(npx * 4, dtype=torch.float32, device=device) for _ in range(2)],  # [previous, current]
            v210=torch.empty(words, dtype=torch.int32, device=device))
        chan[ch]["out"][0][3::4] = 1.0  # frame -1 of every channel: opaque black
    on_device = args.backend == "nccl" and (world > 1 or args.loopback)
    if on_device:
        ex = multigpu.DeviceRouteExchange(ctx, routes, rank, world, npx * 4, torch.float32, device, C,
                                          unique_id=multigpu.share_route_id(dist, rank), loopback=args.loopback)
    else:
        ex = multigpu.RouteExchange(routes, rank, world, npx * 4, torch.float32, device, C, via_host=(args.backend == "gloo"))
    torch.cuda.synchronize()
    exch_s = [0.0]

    def step(i):
        if on_device:
            # the routed frames of the previous step travel on the communication stream WHILE this step's reads run;
            # nothing waits on the host
            ex.start({ch: chan[ch]["out"][0] for ch in mine})
            for ch in mine:
                for l in range(3):
                    ctx.v210_read(chan[ch]["layers"][l], chan[ch]["rgba"][l], w, h, *rd)
            routed = ex.finish()                     # process queue waits for the comm stream (on the device)
        else:
            ctx.wait()                               # previous outputs are complete before they travel
            t0 = time.perf_counter()
            routed = ex.exchange({ch: chan[ch]["out"][0] for ch in mine})
            torch.cuda.synchronize()                 # torch.distributed runs on torch's stream, the kernels on the library's
            exch_s[0] += time.perf_counter() - t0
            for ch in mine:
                for l in range(3):
                    ctx.v210_read(chan[ch]["layers"][l], chan[ch]["rgba"][l], w, h, *rd)
        for ch in mine:
            c = chan[ch]
            ctx.combine(c["rgba"] + [routed[ch]], c["out"][1], w, h)
            ctx.v210_write(c["out"][1], c["v210"], w, h, 0, *wr)
        for ch in mine:
            chan[ch]["out"].reverse()

    def sync():
        ctx.wait()
        torch.cuda.synchronize()

    checked = False
    if args.check:  # by hand: what arrives as channel k's routed layer is, bit for bit, channel src(k)'s output
        step(0)
        sync()
        prints = torch.zeros(total, dtype=torch.int64)
        for ch in mine:
            prints[ch] = multigpu.frame_fingerprint(chan[ch]["out"][0])
        if dist is not None:
            if args.backend == "nccl":
                p = prints.to(device)
                dist.all_reduce(p)
                prints = p.cpu()
            else:
                dist.all_reduce(prints)
        if on_device:
            ex.start({ch: chan[ch]["out"][0] for ch in mine})
            routed = ex.finish()
        else:
            routed = ex.exchange({ch: chan[ch]["out"][0] for ch in mine})
        sync()
        for ch in mine:
            src = (ch + total // 2) % total
            got = multigpu.frame_fingerprint(routed[ch])
            assert got == int(prints[src]), (ch, src, got, int(prints[src]))
            assert on_device or world > 1 or routed[ch] is chan[src]["out"][0]
        checked = True
        if rank == 0:
            log("route check ok: %d channels on %d rank(s), %s" % (total, world, "ph_route (RCCL)" if on_device else
                "torch.distributed" if world > 1 else "local alias"))
    # one hand-off alone (nothing to overlap with): what a hop costs when the sink has to wait for it
    hop_ms = None
    if on_device and (ex.plan.sends or ex.plan.recvs):
        sync()
        for rep in range(4):
            if rep == 1:
                sync()
                t0 = time.perf_counter()
            ex.start({ch: chan[ch]["out"][0] for ch in mine})
            ex.finish()
        sync()
        hop_ms = 1e3 * (time.perf_counter() - t0) / 3
    exch_s[0] = 0.0
    elapsed = multigpu.timed_steps(step, args.steps, args.warmup, sync, dist, device if args.backend == "nccl" else None)
    fps = total * args.steps / elapsed
    per_step_exch = exch_s[0] / (args.steps + args.warmup)
    rec = {
        "workload": "config 5: %d channels of %dx%d on %d GPU(s), 3 v210 layers + 1 routed RGBA layer each" % (total, w, h, world),
        "frames_per_sec": round(fps, 1), "steps": args.steps, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "channels": total, "channels_per_gpu": C, "ranks": world,
        "routes_crossing_ranks_per_rank": len(ex.plan.sends), "route_bytes_per_rank_per_step": ex.traffic_bytes(),
        "bytes_per_hop": npx * 16,
        "hop_ms_unoverlapped": round(hop_ms, 3) if hop_ms is not None else None,
        "hop_GBps_unoverlapped": round(ex.traffic_bytes() / (hop_ms * 1e-3) / 1e9, 1) if hop_ms else None,
        "rccl_ranks_in_communicator": ex.route.comm_count() if on_device else None,
        "fingerprint_check": "ok" if checked else "not run",
        "exchange_ms_per_step_rank0": round(1e3 * per_step_exch, 3),
        "exchange_GBps_rank0": round(ex.traffic_bytes() / per_step_exch / 1e9, 1) if per_step_exch > 0 and ex.traffic_bytes() else None,
        "path": "ph_route: RCCL on its own stream, event-ordered, overlapped with the v210 reads" if on_device else
                ("torch.distributed %s, host-synchronised" % args.backend) if world > 1 else "single rank: routes alias local buffers"}
    if args.print_fingerprints:
        sync()
        log("fingerprints rank %d: %s" % (rank, " ".join("%d:%x" % (ch, multigpu.frame_fingerprint(chan[ch]["v210"]) & (2 ** 64 - 1)) for ch in mine)))
    if on_device:
        ex.close()
    return rec


if __name__ == "__main__":
    main()
