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
import torch
import torch.distributed as dist

from . import capi


def _device_untile(compact_all, width, height, world, frame):
    capi.untile(compact_all.data_ptr(), width, height, world, frame.data_ptr(), compact_all.device.index,
                torch.cuda.current_stream(compact_all.device).cuda_stream)


def _device_finalize(accum, width, height, total_spp, frame):
    capi.finalize(accum.data_ptr(), width, height, total_spp, frame.data_ptr(), accum.device.index,
                  torch.cuda.current_stream(accum.device).cuda_stream)


class TileFrame:
    """Tile-partitioned frame: render_local() -> gather() -> self.frame (on every rank)."""

    def __init__(self, width, height, world, rank, device, untile_fn=None, sample_chunks=0):
        self.width, self.height, self.world, self.rank = width, height, world, rank
        self.sample_chunks = sample_chunks
        self.n = capi.compact_floats(width, height, world)
        self.frame = torch.zeros(height * width * 3, dtype=torch.float32, device=device)
        self.local = torch.zeros(self.n, dtype=torch.float32, device=device) if world > 1 else None
        self.all = torch.zeros(self.n * world, dtype=torch.float32, device=device) if world > 1 else None
        self.untile_fn = untile_fn or _device_untile

    def render_local(self, scene, cam, spp, max_depth, seed, want_stats=False, flags=0):
        """Launch this rank's tiles on torch's current stream (asynchronous unless want_stats)."""
        stream = torch.cuda.current_stream(self.frame.device).cuda_stream
        if self.world == 1:
            return scene.render(cam, self.width, self.height, spp, max_depth=max_depth, seed=seed, out_rgb=self.frame.data_ptr(),
                                want_stats=want_stats, stream=stream, flags=flags, sample_chunks=self.sample_chunks)[2]
        return scene.render(cam, self.width, self.height, spp, max_depth=max_depth, seed=seed, tile_rank=self.rank, tile_ranks=self.world,
                            flags=flags | capi.OUT_COMPACT, out_rgb=self.local.data_ptr(), want_stats=want_stats, stream=stream,
                            sample_chunks=self.sample_chunks)[2]

    def gather(self):
        if self.world == 1:
            return self.frame
        dist.all_gather_into_tensor(self.all, self.local)
        self.untile_fn(self.all, self.width, self.height, self.world, self.frame)
        return self.frame


class SampleFrame:
    """Sample-partitioned frame: render_local() -> reduce() -> self.frame (valid on rank 0)."""

    def __init__(self, width, height, total_spp, world, rank, device, finalize_fn=None, sample_chunks=0):
        if total_spp % world:
            raise ValueError("total_spp must be a multiple of the world size")
        self.width, self.height, self.world, self.rank, self.total_spp = width, height, world, rank, total_spp
        self.spp_local = total_spp // world
        self.sample_chunks = sample_chunks
        while self.sample_chunks > 1 and self.spp_local % self.sample_chunks:
            self.sample_chunks //= 2
        self.accum = torch.zeros(height * width * 3, dtype=torch.float32, device=device)
        self.frame = torch.zeros_like(self.accum)
        self.finalize_fn = finalize_fn or _device_finalize

    def render_local(self, scene, cam, max_depth, seed, want_stats=False):
        stream = torch.cuda.current_stream(self.accum.device).cuda_stream
        return scene.render(cam, self.width, self.height, self.spp_local, max_depth=max_depth, seed=seed, sample_offset=self.rank * self.spp_local,
                            flags=capi.OUT_ACCUM, out_rgb=self.accum.data_ptr(), want_stats=want_stats, stream=stream,
                            sample_chunks=self.sample_chunks)[2]

    def reduce(self):
        if self.world > 1:
            dist.reduce(self.accum, dst=0, op=dist.ReduceOp.SUM)
        if self.rank == 0:
            self.finalize_fn(self.accum, self.width, self.height, self.total_spp, self.frame)
        return self.frame
