"""Multi-GPU frame assembly over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).

The reference is single-GPU.  Pixels are independent, the scene is replicated, and the Philox streams are keyed on
(pixel, sample, bounce), so a frame can be cut two ways without changing any sample:
  tile mode   : 8x8 tiles dealt round-robin to ranks (tile k -> rank k % world: cat pixels cost ~10x background pixels,
                so fine interleaving is what balances load); each rank renders a compact tile list; ONE all-gather of
                W*H*12/world bytes per rank; crt_untile de-interleaves.  Bitwise identical to a 1-GPU frame.
  sample mode : rank r renders samples [r*spp/world, (r+1)*spp/world) of every pixel into raw float sums; ONE
                reduce(sum) of W*H*3 floats to rank 0; crt_finalize divides and applies gamma.  Equal to the 1-GPU frame
                up to float summation order.
torch is plumbing here (buffers, streams, the collective); rendering and the epilogues are the HIP library's.
"""
import contextlib
import os

import torch
import torch.distributed as dist

from . import capi


def _device_untile(compact_all, width, height, world, frame):
    capi.untile(compact_all.data_ptr(), width, height, world, frame.data_ptr(), compact_all.device.index,
                torch.cuda.current_stream(compact_all.device).cuda_stream)


def _device_finalize(accum, width, height, total_spp, frame):
    capi.finalize(accum.data_ptr(), width, height, total_spp, frame.data_ptr(), accum.device.index,
                  torch.cuda.current_stream(accum.device).cuda_stream)


def _gloo_on_gpu_guard(slots):
    """gloo on device tensors (only used to rehearse the N-rank path with all ranks on ONE GPU, where RCCL refuses to run:
    bench.py, CRT_BENCH_BACKEND=gloo): ProcessGroupGloo stages device tensors through streams of torch's HIGH-PRIORITY pool, and
    a high-priority stream that waits for an event behind the persistent render kernel is the configuration of the hang
    DESIGN.md section 5 describes (reproduced here in 2 ranks x 3 frames, every time, with and without the two-launch frame,
    with 8 or 24 hardware queues; it does not happen when the render has finished before the collective is issued).  So the
    rehearsal waits for the render first: it exercises the dealing, the collectives, the epilogues and the bookkeeping, not
    the overlap.  RCCL (the real path) runs at normal priority and is not touched by this."""
    if slots.gpu and dist.is_initialized() and dist.get_backend() == "gloo" and not os.environ.get("CRT_DIST_NO_GUARD"):  # (the switch: to study the hang)
        torch.cuda.synchronize(slots.device)


class _Slots:
    """Frames in flight.  Not pipelined: one buffer set, everything on the caller's current stream.  Pipelined: consecutive
    frames alternate between two render streams, and everything that follows a frame's render -- the wait for its collective
    and its epilogue -- runs on a third stream, over four buffer sets, so that no render ever waits for a collective.

    Why: the render kernel is persistent and fills every wave slot of the GPU, so the younger frame's workgroups only get in
    as the older frame's waves drain: the ramp-down of one frame is covered by the ramp-up of the next.  The collective's
    own kernels need wave slots too and only find them when a render drains; if the next render on that stream had to wait
    for the collective (as it would with two buffer sets), that overlap would be lost on every rank.  (crt_render keeps one
    scratch set per stream, which is what allows two renders to overlap.)"""

    def __init__(self, device, pipelined):
        # pipelined: False / True (two frames in flight) / an integer number of frames in flight (render streams).  A rank's
        # share of a frame shrinks with the number of ranks while the kernel's ramp-up and drain do not: measured on one GPU
        # (tools/rank_share.py), three render streams instead of two are 2 % slower on a whole frame and 1.0 / 1.5 / 2.6 %
        # faster on 1/2, 1/4, 1/8 of it.
        self.device = torch.device(device)
        self.gpu = self.device.type == "cuda"
        self.pipelined = bool(pipelined)
        self.in_flight = (2 if pipelined is True else int(pipelined)) if pipelined else 1   # render streams
        self.n = self.in_flight + 2 if pipelined else 1                                    # buffer sets
        on_gpu = self.gpu and self.pipelined
        self.render_streams = [torch.cuda.Stream(self.device) for _ in range(self.in_flight)] if on_gpu else [None]
        self.epilogue = torch.cuda.Stream(self.device) if on_gpu else None
        self.done = [None] * self.n  # per buffer set: event after the last thing that used it
        self.k = 0                   # frames started

    @property
    def slot(self):
        return self.k % self.n

    def render_stream(self):
        return self.render_streams[self.k % len(self.render_streams)]

    @staticmethod
    def on(stream):
        """Context: `stream` is torch's current stream (None: leave it alone)."""
        return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()

    def begin(self):
        """Start frame k: returns (buffer set, render stream).  The render stream first waits for what the caller enqueued so
        far (uploads, ...) and for the last user of the buffer set (the epilogue of frame k - number of buffer sets)."""
        slot, st = self.slot, self.render_stream()
        if st is not None:
            st.wait_stream(torch.cuda.current_stream(self.device))
            if self.done[slot] is not None:
                st.wait_event(self.done[slot])
        return slot, st

    def mark_done(self, slot, stream):
        if stream is not None:
            ev = torch.cuda.Event()
            ev.record(stream)
            self.done[slot] = ev

    def drain(self):
        """The caller's current stream waits for everything in flight."""
        if self.gpu and self.pipelined:
            cur = torch.cuda.current_stream(self.device)
            for st in self.render_streams + [self.epilogue]:
                cur.wait_stream(st)

    def handle(self, stream):
        if not self.gpu:
            return None
        return (stream if stream is not None else torch.cuda.current_stream(self.device)).cuda_stream


class TileFrame:
    """Tile-partitioned frame.  Synchronous use: render_local(); gather() -> self.frame (on every rank).

    Pipelined use (pipelined=True): render_local(); gather_async(); ... ; drain().  gather_async() enqueues the frame's
    all-gather behind its render and its de-interleave behind that, on the epilogue stream (see _Slots); nothing waits on
    the host (nccl).  Every frame is assembled on every rank exactly as gather() would; self.frame is the frame submitted
    last, complete once drain() has returned (for the caller's current stream) or on `self.slots.epilogue`."""

    def __init__(self, width, height, world, rank, device, untile_fn=None, sample_chunks=0, pipelined=False, force_collective=False):
        self.width, self.height, self.world, self.rank = width, height, world, rank
        self.sample_chunks = sample_chunks
        self.n = capi.compact_floats(width, height, world)
        # one rank renders straight into the frame; force_collective sends it through the compact list, the all-gather
        # and the de-interleave all the same (a 1-rank group: how the RCCL path is exercised on a single-GPU box)
        self.collective = world > 1 or force_collective
        self.slots = _Slots(device, pipelined)
        self._dev_flag = capi.OUT_DEVICE if self.slots.gpu else 0  # device buffers: the library need not query the pointers
        nbuf = self.slots.n
        self._frame = [torch.zeros(height * width * 3, dtype=torch.float32, device=device) for _ in range(nbuf)]
        self._local = [torch.zeros(self.n, dtype=torch.float32, device=device) for _ in range(nbuf)] if self.collective else None
        self._all = [torch.zeros(self.n * world, dtype=torch.float32, device=device) for _ in range(nbuf)] if self.collective else None
        self._newest = 0      # buffer set of the frame submitted last
        self._current = None  # (buffer set, render stream) of the frame between render_local() and gather_async()
        self.untile_fn = untile_fn or _device_untile

    @property
    def frame(self):
        return self._frame[self._newest]

    @property
    def local(self):
        """This rank's compact tile list (the buffer the next render_local() fills)."""
        return self._local[self.slots.slot] if self._local else None

    def render_local(self, scene, cam, spp, max_depth, seed, want_stats=False, flags=0):
        """Launch this rank's tiles (asynchronous unless want_stats)."""
        slot, st = self.slots.begin()
        self._current = (slot, st)
        with self.slots.on(st):
            if not self.collective:
                return scene.render(cam, self.width, self.height, spp, max_depth=max_depth, seed=seed, out_rgb=self._frame[slot].data_ptr(),
                                    want_stats=want_stats, stream=self.slots.handle(st), flags=flags | self._dev_flag, sample_chunks=self.sample_chunks)[2]
            return scene.render(cam, self.width, self.height, spp, max_depth=max_depth, seed=seed, tile_rank=self.rank, tile_ranks=self.world,
                                flags=flags | capi.OUT_COMPACT | self._dev_flag, out_rgb=self._local[slot].data_ptr(), want_stats=want_stats,
                                stream=self.slots.handle(st), sample_chunks=self.sample_chunks)[2]

    def gather_async(self):
        """Enqueue the all-gather of the frame just rendered and, behind it, the de-interleave into its frame buffer."""
        slot, st = self._current if self._current is not None else (self.slots.slot, self.slots.render_stream())
        self._current = None
        if self.collective:
            _gloo_on_gpu_guard(self.slots)
            with self.slots.on(st):  # the collective's stream waits for the render stream up to here
                work = dist.all_gather_into_tensor(self._all[slot], self._local[slot], async_op=True)
            ep = self.slots.epilogue
            with self.slots.on(ep):
                work.wait()          # the epilogue stream waits for the collective; the host does not (nccl)
                self.untile_fn(self._all[slot], self.width, self.height, self.world, self._frame[slot])
            self.slots.mark_done(slot, ep)
        else:
            self.slots.mark_done(slot, st)
        self._newest = slot
        self.slots.k += 1

    def gather(self):
        self.gather_async()
        return self.drain()

    def drain(self):
        """Everything in flight completes before what the caller enqueues next; returns the frame submitted last."""
        self.slots.drain()
        return self.frame


class SampleFrame:
    """Sample-partitioned frame: render_local(); reduce() -> self.frame (valid on rank 0).
    Pipelined (see TileFrame): render_local(); reduce_async(); ...; drain()."""

    def __init__(self, width, height, total_spp, world, rank, device, finalize_fn=None, sample_chunks=0, pipelined=False):
        if total_spp % world:
            raise ValueError("total_spp must be a multiple of the world size")
        self.width, self.height, self.world, self.rank, self.total_spp = width, height, world, rank, total_spp
        self.spp_local = total_spp // world
        # a rank's share of the samples must divide into the chunks; one sample per unit (chunks = total_spp) stays one sample per unit
        self.sample_chunks = self.spp_local if sample_chunks >= total_spp else min(sample_chunks, self.spp_local)
        while self.sample_chunks > 1 and self.spp_local % self.sample_chunks:
            self.sample_chunks -= 1
        self.slots = _Slots(device, pipelined)
        self._dev_flag = capi.OUT_DEVICE if self.slots.gpu else 0
        nbuf = self.slots.n
        self._accum = [torch.zeros(height * width * 3, dtype=torch.float32, device=device) for _ in range(nbuf)]
        self._frame = [torch.zeros(height * width * 3, dtype=torch.float32, device=device) for _ in range(nbuf if rank == 0 else 1)]
        self._newest = 0
        self._current = None
        self.finalize_fn = finalize_fn or _device_finalize

    @property
    def frame(self):
        return self._frame[self._newest if self.rank == 0 else 0]

    @property
    def accum(self):
        """Raw float sums of this rank's samples (the buffer the next render_local() fills)."""
        return self._accum[self.slots.slot]

    def render_local(self, scene, cam, max_depth, seed, want_stats=False, flags=0):
        slot, st = self.slots.begin()
        self._current = (slot, st)
        with self.slots.on(st):
            return scene.render(cam, self.width, self.height, self.spp_local, max_depth=max_depth, seed=seed, sample_offset=self.rank * self.spp_local,
                                flags=flags | capi.OUT_ACCUM | self._dev_flag, out_rgb=self._accum[slot].data_ptr(), want_stats=want_stats,
                                stream=self.slots.handle(st), sample_chunks=self.sample_chunks)[2]

    def reduce_async(self):
        slot, st = self._current if self._current is not None else (self.slots.slot, self.slots.render_stream())
        self._current = None
        ep = self.slots.epilogue if self.slots.epilogue is not None else st
        work = None
        if self.world > 1:
            _gloo_on_gpu_guard(self.slots)
            with self.slots.on(st):
                work = dist.reduce(self._accum[slot], dst=0, op=dist.ReduceOp.SUM, async_op=True)
        with self.slots.on(ep):
            if self.slots.epilogue is not None and st is not None:
                ep.wait_stream(st)  # (with a collective its wait below implies this)
            if work is not None:
                work.wait()
            if self.rank == 0:
                self.finalize_fn(self._accum[slot], self.width, self.height, self.total_spp, self._frame[slot])
        self.slots.mark_done(slot, ep)
        self._newest = slot
        self.slots.k += 1

    def reduce(self):
        self.reduce_async()
        return self.drain()

    def drain(self):
        self.slots.drain()
        return self.frame
