"""Output path (SURVEY section 8 "next" row N2; reference export_framebuffer_to_png, utility.h:122-145).

The oracle here is the reference's host loop restated in numpy (utility.h:133-137):
    (unsigned char) min(max(x * 255.0f, 0.0f), 255.0f)      float multiply, clamp, truncation
and, for the container, an independent PNG decoder written below from the PNG / zlib specifications (python's zlib
inflates the IDAT stream).  The reference's stb encoder is third-party code that is not in /root/reference; decoded
pixels, not file bytes, are what has to agree.
"""
import struct
import zlib

import numpy as np
import pytest


def quantize_oracle(rgb):
    v = rgb.astype(np.float32) * np.float32(255.0)
    v = np.where(np.isnan(v), np.float32(0.0), v)  # std::max(NaN, 0.0f) returns 0.0f (first argument kept only if not less)
    v = np.minimum(np.maximum(v, np.float32(0.0)), np.float32(255.0))
    return v.astype(np.uint8)  # truncation


def decode_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(typ + body) & 0xFFFFFFFF == crc, "bad CRC in %r" % typ
        chunks.append((typ, body))
        pos += 12 + n
    assert chunks[0][0] == b"IHDR" and chunks[-1][0] == b"IEND"
    w, h, depth, ctype, comp, filt, inter = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, ctype, comp, filt, inter) == (8, 2, 0, 0, 0)
    raw = zlib.decompress(b"".join(b for t, b in chunks if t == b"IDAT"))
    rows = np.frombuffer(raw, dtype=np.uint8).reshape(h, 1 + 3 * w)
    assert (rows[:, 0] == 0).all()  # filter type 0 on every scanline
    return rows[:, 1:].reshape(h, w, 3)


@pytest.mark.parametrize("w,h", [(1, 1), (7, 5), (300, 200), (1920, 1080)])
def test_png_round_trip(capi, tmp_path, w, h):
    rng = np.random.default_rng(w * 1000 + h)
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    path = str(tmp_path / "t.png")
    capi.write_png(path, img)
    assert np.array_equal(decode_png(path), img)


def test_png_rejects_bad_arguments(capi, tmp_path):
    with pytest.raises(capi.CrtError):
        capi.write_png(str(tmp_path / "no_such_dir" / "x.png"), np.zeros((2, 2, 3), np.uint8))


@pytest.mark.gpu
def test_device_quantisation_matches_reference_loop(capi):
    rng = np.random.default_rng(5)
    w, h = 333, 77
    rgb = rng.uniform(-0.2, 1.2, size=(h, w, 3)).astype(np.float32)
    special = np.array([0.0, -0.0, 1.0, 0.999999, 254.5 / 255.0, 255.0 / 255.0, 1e-9, np.inf, -np.inf, np.nan, 0.5, 2.0 / 255.0], np.float32)
    rgb.reshape(-1)[:special.size] = special
    assert np.array_equal(capi.quantize_rgb8(rgb, w, h), quantize_oracle(rgb))


@pytest.mark.gpu
def test_rendered_frame_to_png(capi, scenes, tmp_path):
    s = scenes.build(scenes.cat(), 0)
    _, cam = scenes.make_camera(160)
    rgb, _, _ = s.render(cam, 160, 90, 4)
    img = capi.quantize_rgb8(rgb, 160, 90)
    assert np.array_equal(img, quantize_oracle(rgb.reshape(90, 160, 3)))
    path = str(tmp_path / "frame.png")
    capi.write_png(path, img)
    assert np.array_equal(decode_png(path), img)
