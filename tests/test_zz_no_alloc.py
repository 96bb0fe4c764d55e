"""crt_render_noalloc (include/crt.h): the entry point for callers that cannot allocate.  Its output must be crt_render's,
bit for bit, on two calls in a row, the second with a smaller input (the static scratch keeps the first call's contents)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(capi, s, cam, W, H, spp, chunks, **kw):
    import torch
    want = torch.empty(H * W * 3, dtype=torch.float32, device="cuda")
    want_ids = torch.empty(H * W, dtype=torch.int32, device="cuda")
    st = s.render(cam, W, H, spp, out_rgb=want.data_ptr(), out_ids=want_ids.data_ptr(), sample_chunks=chunks, flags=capi.OUT_DEVICE | kw.get("flags", 0),
                  **{k: v for k, v in kw.items() if k != "flags"})[2]
    got = torch.full_like(want, -1.0)
    got_ids = torch.full_like(want_ids, -7)
    rays = torch.zeros(1, dtype=torch.int64, device="cuda")
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())  # the fills above ran on the current stream: the render must come behind them
    capi.render_noalloc(s, cam, W, H, spp, got.data_ptr(), got_ids.data_ptr(), rays.data_ptr(), sample_chunks=chunks, stream=stream.cuda_stream, **kw)
    stream.synchronize()
    assert torch.equal(got, want), "image differs from crt_render's: %d floats" % int((got != want).sum())
    assert torch.equal(got_ids, want_ids) and int(rays.item()) == st.rays
    return got


@pytest.mark.parametrize("node_cache", ["1", "4"])  # 4: every frame as two launches on two streams (the helper stream is made at commit)
def test_noalloc_matches_crt_render_twice_in_a_row(capi, scenes, monkeypatch, node_cache):
    import torch
    monkeypatch.setenv("CRT_NODE_CACHE", node_cache)
    s = scenes.build(scenes.cats(), 0)
    # first call: sample chunks through slabs in the static scratch, 480 x 270 x 4 chunks = 6.2 MB
    _, cam = scenes.make_camera(480)
    _both(capi, s, cam, 480, 270, 8, 4)
    # sample chunks whose slabs (1280 x 720 x 16 x 12 B = 177 MB) do not fit the 32 MiB: the waves' rings do, and an odd chunk count
    _, cam3 = scenes.make_camera(1280)
    _both(capi, s, cam3, 1280, 720, 16, 16)
    _both(capi, s, cam, 480, 270, 9, 3, seed=3)
    # second call, smaller input, other seed and sample window, raw sums: nothing of the first frame may show through
    _, cam2 = scenes.make_camera(203)
    _both(capi, s, cam2, 203, 117, 4, 2, seed=7, sample_offset=11, ids_sample=12, flags=capi.OUT_ACCUM)
    # whole pixels (no slab) at a size with a ragged tile edge, and a compact tile list of one rank of three
    _both(capi, s, cam2, 203, 117, 3, 0)
    n = capi.compact_floats(203, 117, 3)
    want = torch.empty(n, dtype=torch.float32, device="cuda")
    got = torch.full_like(want, -1.0)
    s.render(cam2, 203, 117, 2, out_rgb=want.data_ptr(), tile_rank=1, tile_ranks=3, flags=capi.OUT_COMPACT | capi.OUT_DEVICE, sample_chunks=2)
    capi.render_noalloc(s, cam2, 203, 117, 2, got.data_ptr(), tile_rank=1, tile_ranks=3, flags=capi.OUT_COMPACT, sample_chunks=2)
    torch.cuda.synchronize()
    assert torch.equal(got, want)


@pytest.mark.parametrize("node_cache, size", [("1", (480, 270, 4, 4)), ("4", (1536, 864, 8, 2))])
def test_noalloc_under_stream_capture(capi, scenes, monkeypatch, node_cache, size):
    """What the entry point is for: captured into a HIP graph (torch.cuda.graph) -- one launch, or the two-launch form with its
    helper stream forked and joined by events -- the capture itself renders nothing, every replay renders crt_render's frame."""
    import torch
    monkeypatch.setenv("CRT_NODE_CACHE", node_cache)
    W, H, spp, chunks = size
    s = scenes.build(scenes.cats(), 0)
    _, cam = scenes.make_camera(W)
    want = torch.empty(H * W * 3, dtype=torch.float32, device="cuda")
    s.render(cam, W, H, spp, out_rgb=want.data_ptr(), sample_chunks=chunks, flags=capi.OUT_DEVICE)
    got = torch.zeros_like(want)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=torch.cuda.Stream()):
        capi.render_noalloc(s, cam, W, H, spp, got.data_ptr(), sample_chunks=chunks, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert bool((got == 0).all().item())
    for _ in range(2):
        got.fill_(-1.0)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(got, want)


def test_noalloc_rejects_what_does_not_fit(capi, scenes):
    import torch
    s = scenes.build(scenes.cat(), 0)
    _, cam = scenes.make_camera(1920)
    out = torch.empty(1080 * 1920 * 3, dtype=torch.float32, device="cuda")
    # the benchmarked configuration at its full size: 1.6 GB as slabs, 31.6 MB as rings of four 128-sample blocks per wave -- accepted,
    # and crt_render's frame (which takes the slabs) bit for bit
    want = torch.empty_like(out)
    s.render(cam, 1920, 1080, 64, out_rgb=want.data_ptr(), sample_chunks=64, flags=capi.OUT_DEVICE)
    capi.render_noalloc(s, cam, 1920, 1080, 64, out.data_ptr(), sample_chunks=64)
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    # 2048 samples per pixel: a ring block is at least one pixel's samples (24 KB), two of them for every persistent wave are 250 MB
    with pytest.raises(capi.CrtError) as e:
        capi.render_noalloc(s, cam, 1920, 1080, 2048, out.data_ptr(), sample_chunks=2048)
    assert e.value.code == capi.CRT_ERR_LIMIT and "32 MiB" in str(e.value)
    # the same frame with whole pixels needs no slab: accepted whatever the size
    capi.render_noalloc(s, cam, 1920, 1080, 1, out.data_ptr())
    torch.cuda.synchronize()
    want = s.render(cam, 1920, 1080, 1)[0]
    assert np.array_equal(out.cpu().numpy().reshape(want.shape), want)
