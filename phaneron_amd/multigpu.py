"""Channel partitioning across GPUs and the ROUTE frame hand-off.

The reference runs every channel in one process on one OpenCL device; channels share nothing
but the context (src/index.ts:156-160), so they shard one-per-GPU with no collective on the
data path.  The only cross-channel data movement is the ROUTE producer
(src/producer/routeProducer.ts:63-126), which taps another channel's combiner output and, in
the reference, just adds a reference to the same OpenCL buffer.  When source and sink channels
live on different GPUs that reference becomes one point-to-point message per frame per route:
`torch.distributed` send/recv, which is RCCL over xGMI with backend "nccl" and plain sockets
with "gloo" (CPU tests).  No all-reduce / all-gather is ever issued for pixels.

One process per GPU: rank r owns the channels `channels_of_rank(r)`.
"""
import time
from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple


def channel_rank(channel: int, world: int, channels_per_rank: int = 0) -> int:
    """Rank that owns `channel`.  channels_per_rank == 0: round-robin (configs 1-4);
    otherwise blocks of that many consecutive channels per GPU (config 5: 2 per GPU, so that
    channel k and channel k + 8 of 16 sit on different GPUs)."""
    if channels_per_rank > 0:
        return (channel // channels_per_rank) % world
    return channel % world


def channels_of_rank(rank: int, num_channels: int, world: int, channels_per_rank: int = 0) -> List[int]:
    return [c for c in range(num_channels) if channel_rank(c, world, channels_per_rank) == rank]


@dataclass(frozen=True)
class Route:
    """Channel `dst` shows the combiner output of channel `src` as one of its layers
    (AMCP `PLAY 1-10 route://2`, routeProducer.ts:51-61)."""
    src: int
    dst: int


@dataclass
class RoutePlan:
    local: List[Route]                     # both ends on this rank: alias, no copy (as the reference)
    sends: List[Tuple[Route, int]]         # (route, destination rank)
    recvs: List[Tuple[Route, int]]         # (route, source rank)


def plan_routes(routes: Sequence[Route], rank: int, world: int, channels_per_rank: int = 0) -> RoutePlan:
    """Deterministic per-rank plan.  Both ends of a rank pair walk the routes in the same
    (src, dst) order, so untagged RCCL P2P operations match up."""
    plan = RoutePlan([], [], [])
    for r in sorted(routes, key=lambda r: (r.src, r.dst)):
        s, d = channel_rank(r.src, world, channels_per_rank), channel_rank(r.dst, world, channels_per_rank)
        if s == rank and d == rank:
            plan.local.append(r)
        elif s == rank:
            plan.sends.append((r, d))
        elif d == rank:
            plan.recvs.append((r, s))
    return plan


class RouteExchange:
    """Per-frame hand-off of routed frames.  `exchange(frames)` takes {src channel: tensor} for the
    channels this rank owns and returns {dst channel: tensor} for the routes that end here.
    Receive buffers are allocated once (a frame is 132 710 400 B at 2160p f32 RGBA)."""

    def __init__(self, routes: Sequence[Route], rank: int, world: int, frame_numel: int, dtype, device,
                 channels_per_rank: int = 0, via_host: bool = False):
        """via_host: stage every message through host memory (for process groups that cannot move
        device tensors, i.e. gloo with frames on a GPU: functional tests on one GPU)."""
        import torch
        self.plan = plan_routes(routes, rank, world, channels_per_rank)
        self.rank, self.world, self.via_host = rank, world, via_host
        self.recv_bufs = {rt: torch.empty(frame_numel, dtype=dtype, device=device) for rt, _ in self.plan.recvs}
        self.host_bufs = {rt: torch.empty(frame_numel, dtype=dtype) for rt, _ in self.plan.recvs} if via_host else {}
        self.bytes_per_frame = frame_numel * torch.empty((), dtype=dtype).element_size()

    def exchange(self, frames: Dict[int, "object"]) -> Dict[int, "object"]:
        import torch.distributed as dist
        out = {}
        for rt in self.plan.local:
            out[rt.dst] = frames[rt.src]  # same device: share the buffer (reference: addRef per fork)
        ops = []
        for rt, peer in self.plan.sends:
            ops.append(dist.P2POp(dist.isend, frames[rt.src].cpu() if self.via_host else frames[rt.src], peer))
        for rt, peer in self.plan.recvs:
            ops.append(dist.P2POp(dist.irecv, self.host_bufs[rt] if self.via_host else self.recv_bufs[rt], peer))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for rt, _ in self.plan.recvs:
            if self.via_host:
                self.recv_bufs[rt].copy_(self.host_bufs[rt])
            out[rt.dst] = self.recv_bufs[rt]
        return out

    def traffic_bytes(self) -> int:
        """bytes this rank sends + receives per frame"""
        return self.bytes_per_frame * (len(self.plan.sends) + len(self.plan.recvs))


class DeviceRouteExchange:
    """The same hand-off on the library's own path (ph_route_*, include/phaneron_hip.h): RCCL send / recv on a
    communication stream of its own, ordered against the context's process queue ON THE DEVICE - no host wait, and
    the transfer overlaps whatever the process queue runs between `start` and `finish` (the sink's v210 reads).

        ex.start(frames)      # comm stream waits for the process queue's work so far, then posts this period's
                              # sends and receives as one RCCL group
        ... enqueue work that does not need the routed frames ...
        routed = ex.finish()  # the process queue waits for the comm stream; returns {dst channel: tensor}

    loopback=True sends same-rank routes through RCCL too (peer == own rank) instead of aliasing the buffer: the
    way to exercise this path on a single GPU."""

    def __init__(self, ctx, routes: Sequence[Route], rank: int, world: int, frame_numel: int, dtype, device,
                 channels_per_rank: int = 0, unique_id: bytes = None, loopback: bool = False):
        import torch
        from . import capi
        self.plan = plan_routes(routes, rank, world, channels_per_rank)
        if loopback:
            self.plan = RoutePlan([], [(rt, rank) for rt in self.plan.local] + self.plan.sends,
                                  [(rt, rank) for rt in self.plan.local] + self.plan.recvs)
        self.route = capi.Route(ctx, unique_id, rank, world)
        self.recv_bufs = {rt: torch.empty(frame_numel, dtype=dtype, device=device) for rt, _ in self.plan.recvs}
        self.bytes_per_frame = frame_numel * torch.empty((), dtype=dtype).element_size()
        self._out = {}

    def start(self, frames: Dict[int, "object"]) -> None:
        self._out = {rt.dst: frames[rt.src] for rt in self.plan.local}  # same device: share the buffer
        if not (self.plan.sends or self.plan.recvs):
            return
        self.route.after_queue()
        with self.route.group():
            for rt, peer in self.plan.sends:
                self.route.send(frames[rt.src], peer, self.bytes_per_frame)
            for rt, peer in self.plan.recvs:
                self.route.recv(self.recv_bufs[rt], peer, self.bytes_per_frame)

    def finish(self) -> Dict[int, "object"]:
        if self.plan.sends or self.plan.recvs:
            self.route.queue_after()
        for rt, _ in self.plan.recvs:
            self._out[rt.dst] = self.recv_bufs[rt]
        return self._out

    def traffic_bytes(self) -> int:
        return self.bytes_per_frame * (len(self.plan.sends) + len(self.plan.recvs))

    def close(self):
        self.route.destroy()


def share_route_id(dist, rank: int):
    """One ph_route_unique_id for the job: made on rank 0, handed round over the existing process group."""
    from . import capi
    box = [capi.route_unique_id() if rank == 0 else None]
    if dist is not None:
        dist.broadcast_object_list(box, src=0)
    return box[0]


def frame_fingerprint(t):
    """Position-sensitive 64-bit fingerprint of a frame, computed on its device (a permuted or partly stale frame
    changes it; a plain sum would not notice)."""
    import torch
    v = t.view(torch.int32).to(torch.int64)
    k = (torch.arange(v.numel(), device=v.device, dtype=torch.int64) * 2654435761 + 40503) | 1
    return int(((v * k).sum()).item())


def timed_steps(step, steps: int, warmup: int, sync, dist=None, device=None, local: dict = None) -> float:
    """The bench timing contract: `warmup` untimed steps, then exactly `steps` steps bracketed by
    barrier + device sync on both sides; returns the MAX elapsed seconds over ranks.  `local`, if given,
    receives this rank's own elapsed seconds under "elapsed" (per-rank diagnostics)."""
    for i in range(warmup):
        step(i)
    sync()
    if dist is not None:
        dist.barrier()
        sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    sync()
    elapsed = time.perf_counter() - t0
    if local is not None:
        local["elapsed"] = elapsed
    if dist is not None:
        import torch
        dist.barrier()
        sync()
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed
