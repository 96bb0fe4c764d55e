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

import torch
import torch.distributed as dist

from . import capi


def _device_untile(compact_all, width, height, world, frame):
    capi.untile(compact_all.data_ptr(), width, height, world, frame.data_ptr(), compact_all.device.index,
                torch.cuda.current_stream(compact_all.device).cuda_stream)


def _device_finalize(accum, width, height, total_spp, frame):
    capi.finalize(accum.data_ptr(), width, height, total_spp, frame.data_ptr(), accum.device.index,
                  torch.cuda.current_stream(accum.device).cuda_stream)


class _Slots:
    """Frames in flight.  Not pipelined: one slot on the caller's current stream.  Pipelined: two slots, each with its own
    HIP stream and buffers; frame k uses slot k % 2, so frame k + 1 is launched while frame k is still running.  The render
    kernel is persistent and fills every wave slot of the GPU, so the younger frame's workgroups only get in as the older
    frame's waves drain: the ramp-down of one frame is covered by the ramp-up of the next, and so are its collective
    and its epilogue.  (crt_render keeps one scratch set per stream, which is what allows the overlap.)"""

    def __init__(self, device, pipelined):
        self.device = torch.device(device)
        self.gpu = self.device.type == "cuda"
        self.n = 2 if pipelined else 1
        self.streams = [torch.cuda.Stream(self.device) for _ in range(self.n)] if (self.gpu and pipelined) else [None] * self.n
        self.k = 0                      # frames started
        self.pending = [None] * self.n  # per slot: what finishes the frame that used it last

    @property
    def slot(self):
        return self.k % self.n

    def on(self, slot):
        """Context: the slot's stream is torch's current stream."""
        st = self.streams[slot]
        return torch.cuda.stream(st) if st is not None else contextlib.nullcontext()

    def begin(self, finish):
        """Start a frame on the current slot: finish the frame that used the slot before; returns the slot."""
        slot = self.slot
        st = self.streams[slot]
        if st is not None:
            st.wait_stream(torch.cuda.current_stream(self.device))  # what the caller enqueued so far (uploads, ...) comes first
        self.finish_slot(slot, finish)
        return slot

    def finish_slot(self, slot, finish):
        if self.pending[slot] is not None:
            pending, self.pending[slot] = self.pending[slot], None
            with self.on(slot):
                finish(pending, slot)

    def drain(self, finish):
        """Finish every frame in flight, oldest first; the caller's current stream then waits for the slots' streams."""
        for d in range(self.n):
            self.finish_slot((self.k + d) % self.n, finish)
        for st in self.streams:
            if st is not None:
                torch.cuda.current_stream(self.device).wait_stream(st)

    def stream_handle(self, slot):
        if not self.gpu:
            return None
        st = self.streams[slot]
        return (st if st is not None else torch.cuda.current_stream(self.device)).cuda_stream


class TileFrame:
    """Tile-partitioned frame.  Synchronous use: render_local(); gather() -> self.frame (on every rank).

    Pipelined use (pipelined=True): render_local(); gather_async(); ... ; drain().  Two frames are in flight on two HIP
    streams with their own buffers: gather_async() starts the frame's all-gather behind its render, the frame is
    finished (wait for the collective + de-interleave, on its stream) when its slot is used again or by drain().
    Every frame is assembled on every rank exactly as gather() would; self.frame is the youngest finished frame."""

    def __init__(self, width, height, world, rank, device, untile_fn=None, sample_chunks=0, pipelined=False, force_collective=False):
        self.width, self.height, self.world, self.rank = width, height, world, rank
        self.sample_chunks = sample_chunks
        self.n = capi.compact_floats(width, height, world)
        # one rank renders straight into the frame; force_collective sends it through the compact list, the all-gather
        # and the de-interleave all the same (a 1-rank group: how the RCCL path is exercised on a single-GPU box)
        self.collective = world > 1 or force_collective
        self.slots = _Slots(device, pipelined)
        nbuf = self.slots.n
        self._frame = [torch.zeros(height * width * 3, dtype=torch.float32, device=device) for _ in range(nbuf)]
        self._local = [torch.zeros(self.n, dtype=torch.float32, device=device) for _ in range(nbuf)] if self.collective else None
        self._all = [torch.zeros(self.n * world, dtype=torch.float32, device=device) for _ in range(nbuf)] if self.collective else None
        self._newest = 0  # slot of the youngest finished (or, without a collective, launched) frame
        self.untile_fn = untile_fn or _device_untile

    @property
    def frame(self):
        return self._frame[self._newest]

    @property
    def local(self):
        """This rank's compact tile list (the buffer the next render_local() fills)."""
        return self._local[self.slots.slot] if self._local else None

    def _finish(self, work, slot):
        if work is not True:
            work.wait()  # the slot's stream waits for the collective; the host does not (nccl)
            self.untile_fn(self._all[slot], self.width, self.height, self.world, self._frame[slot])
        self._newest = slot

    def render_local(self, scene, cam, spp, max_depth, seed, want_stats=False, flags=0, timing=None):
        """Launch this rank's tiles (asynchronous unless want_stats).  timing = (start, end) torch events recorded around
        the launch on the stream it runs on."""
        slot = self.slots.begin(self._finish)
        stream = self.slots.stream_handle(slot)
        with self.slots.on(slot):
            if timing:
                timing[0].record()
            if not self.collective:
                st = scene.render(cam, self.width, self.height, spp, max_depth=max_depth, seed=seed, out_rgb=self._frame[slot].data_ptr(),
                                  want_stats=want_stats, stream=stream, flags=flags, sample_chunks=self.sample_chunks)[2]
            else:
                st = scene.render(cam, self.width, self.height, spp, max_depth=max_depth, seed=seed, tile_rank=self.rank, tile_ranks=self.world,
                                  flags=flags | capi.OUT_COMPACT, out_rgb=self._local[slot].data_ptr(), want_stats=want_stats, stream=stream,
                                  sample_chunks=self.sample_chunks)[2]
            if timing:
                timing[1].record()
        return st

    def gather_async(self):
        """Start the all-gather of the frame just rendered (behind its render, on its slot's stream)."""
        slot = self.slots.slot
        if self.collective:
            with self.slots.on(slot):
                self.slots.pending[slot] = dist.all_gather_into_tensor(self._all[slot], self._local[slot], async_op=True)
        else:
            self.slots.pending[slot] = True
        self.slots.k += 1

    def gather(self):
        self.gather_async()
        return self.drain()

    def drain(self):
        """Finish the frames in flight; self.frame is then the last frame rendered, complete on the caller's stream."""
        self.slots.drain(self._finish)
        return self.frame


class SampleFrame:
    """Sample-partitioned frame: render_local(); reduce() -> self.frame (valid on rank 0).
    Pipelined (two frames in flight, see TileFrame): render_local(); reduce_async(); ...; drain()."""

    def __init__(self, width, height, total_spp, world, rank, device, finalize_fn=None, sample_chunks=0, pipelined=False):
        if total_spp % world:
            raise ValueError("total_spp must be a multiple of the world size")
        self.width, self.height, self.world, self.rank, self.total_spp = width, height, world, rank, total_spp
        self.spp_local = total_spp // world
        self.sample_chunks = sample_chunks
        while self.sample_chunks > 1 and self.spp_local % self.sample_chunks:
            self.sample_chunks //= 2
        self.slots = _Slots(device, pipelined)
        nbuf = self.slots.n
        self._accum = [torch.zeros(height * width * 3, dtype=torch.float32, device=device) for _ in range(nbuf)]
        self._frame = [torch.zeros(height * width * 3, dtype=torch.float32, device=device) for _ in range(nbuf if rank == 0 else 1)]
        self._newest = 0
        self.finalize_fn = finalize_fn or _device_finalize

    @property
    def frame(self):
        return self._frame[self._newest if self.rank == 0 else 0]

    @property
    def accum(self):
        """Raw float sums of this rank's samples (the buffer the next render_local() fills)."""
        return self._accum[self.slots.slot]

    def _finish(self, work, slot):
        if work is not True:
            work.wait()
        if self.rank == 0:
            self.finalize_fn(self._accum[slot], self.width, self.height, self.total_spp, self._frame[slot])
            self._newest = slot

    def render_local(self, scene, cam, max_depth, seed, want_stats=False, timing=None, flags=0):
        slot = self.slots.begin(self._finish)
        stream = self.slots.stream_handle(slot)
        with self.slots.on(slot):
            if timing:
                timing[0].record()
            st = scene.render(cam, self.width, self.height, self.spp_local, max_depth=max_depth, seed=seed, sample_offset=self.rank * self.spp_local,
                              flags=flags | capi.OUT_ACCUM, out_rgb=self._accum[slot].data_ptr(), want_stats=want_stats, stream=stream,
                              sample_chunks=self.sample_chunks)[2]
            if timing:
                timing[1].record()
        return st

    def reduce_async(self):
        slot = self.slots.slot
        if self.world > 1:
            with self.slots.on(slot):
                self.slots.pending[slot] = dist.reduce(self._accum[slot], dst=0, op=dist.ReduceOp.SUM, async_op=True)
        else:
            self.slots.pending[slot] = True
        self.slots.k += 1

    def reduce(self):
        self.reduce_async()
        return self.drain()

    def drain(self):
        self.slots.drain(self._finish)
        return self.frame
