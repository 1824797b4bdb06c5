"""Producer / consumer staging for one channel: the queue.load / queue.unload roles of
the reference's io.ts:79-98,166-174 with a ring of frames in flight.

The reference stages one frame at a time and orders its three queues through host-side
`waitFinish` (ffmpegProducer / macadamConsumer).  Here a ring of `depth` slots keeps the PCIe
link and the kernels busy together, ordered on the device:

    host fills slot k's pinned source mirrors  (producer thread / decoder output)
    LOAD    : n async H2D copies from the mirrors
    PROCESS : waits for LOAD's copies (event), runs the frame's kernels
    UNLOAD  : waits for PROCESS (event), async D2H into the output mirror, records the slot's event
    host waits for that event only when slot k comes round again (depth frames later), then hands
    the output mirror to the consumer.

Only the boundary's host-buffer hand-over lives here; the kernels are whatever `process` enqueues.
"""
from . import capi


class Slot:
    def __init__(self, ctx, source_bytes, output_bytes, tag):
        self.sources = [ctx.create_buffer(b, "readonly", "coarse", owner="%s src%d" % (tag, i))
                        for i, b in enumerate(source_bytes)]
        self.output = ctx.create_buffer(output_bytes, "writeonly", "coarse", owner="%s out" % tag)
        self.done = None      # event behind the slot's download
        self.frame = None     # frame number in flight in this slot

    def release(self):
        if self.done:
            self.done.destroy()
        for b in self.sources + [self.output]:
            b.release()


class StagedChannel:
    """`process(ctx, sources, output)` enqueues the frame's kernels on the PROCESS queue from the
    slot's source Buffers into its output Buffer.  `fill(frame_no, mirrors)` writes the frame's
    source bytes into the pinned mirrors (numpy uint8 views); `consume(frame_no, mirror)` is given
    the finished output bytes."""

    def __init__(self, ctx, source_bytes, output_bytes, process, depth=3, tag="chan"):
        self.ctx, self.process, self.depth = ctx, process, depth
        self.slots = [Slot(ctx, source_bytes, output_bytes, "%s slot%d" % (tag, k)) for k in range(depth)]
        self.submitted = 0

    def _retire(self, slot, consume):
        if slot.done is not None:
            slot.done.wait()
            slot.done.destroy()
            slot.done = None
            if consume is not None:
                consume(slot.frame, slot.output.host())

    def submit(self, fill, consume=None):
        """Stage one frame; returns its frame number.  May call consume() for the frame that used
        the slot `depth` frames ago."""
        ctx = self.ctx
        slot = self.slots[self.submitted % self.depth]
        self._retire(slot, consume)
        for b in slot.sources:
            b.host_access("writeonly", capi.QUEUE_LOAD)       # expose the mirror
        fill(self.submitted, [b.host() for b in slot.sources])
        for b in slot.sources:
            b.host_access("none", capi.QUEUE_LOAD)            # async H2D on LOAD
        ctx.queue_wait_queue(capi.QUEUE_PROCESS, capi.QUEUE_LOAD)
        self.process(ctx, slot.sources, slot.output)
        ctx.queue_wait_queue(capi.QUEUE_UNLOAD, capi.QUEUE_PROCESS)
        slot.output.download_async(capi.QUEUE_UNLOAD)
        slot.done = ctx.record_event(capi.QUEUE_UNLOAD)
        # the next upload into this slot must not overtake this frame's kernels; LOAD is ordered
        # behind PROCESS when the slot is reused because _retire() waited for the download
        slot.frame = self.submitted
        self.submitted += 1
        return slot.frame

    def drain(self, consume=None):
        """Retire every frame still in flight, oldest first."""
        first = max(0, self.submitted - self.depth)
        for f in range(first, self.submitted):
            self._retire(self.slots[f % self.depth], consume)

    def close(self):
        self.drain()
        for s in self.slots:
            s.release()
        self.slots = []
