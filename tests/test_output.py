"""Output stage (what the reference leaves to `save(path, clamp01.(img))`, README.md:50-67): clamp, N0f8 quantisation,
PNG / PPM writers.  Host side here; the device kernel is checked in test_gpu_parity.py."""
import struct
import zlib

import numpy as np

import raytracing_amd as rt
from raytracing_amd import _ffi as ffi

F = np.float32


def decode_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(kind + body)
        chunks.append((kind, body))
        pos += 12 + n
    assert [k for k, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    w, h, depth, colour, comp, filt, lace = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, colour, comp, filt, lace) == (8, 2, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(h, 1 + 3 * w)
    assert np.all(raw[:, 0] == 0)
    return raw[:, 1:].reshape(h, w, 3)


def test_quantisation_matches_n0f8():
    # clamp01nan + round-half-even of x * 255 in Float32
    vals = np.array([-1.0, 0.0, 0.5 / 255, 1.5 / 255, 2.5 / 255, 0.25, 0.5, 0.999, 1.0, 7.0, np.nan, np.inf, -np.inf], F)
    img = np.zeros((3, len(vals), 3), F)
    img[0, :, 0] = vals
    img[1, :, 1] = vals[::-1]
    got = rt.to_rgb8(img)
    want = np.rint(rt.clamp01(img) * F(255)).astype(np.uint8)
    np.testing.assert_array_equal(got, want)
    assert list(got[0, :5, 0]) == [0, 0, 0, 2, 2]           # ties go to the even neighbour
    assert got[0, 9, 0] == 255 and got[0, 10, 0] == 0 and got[0, 11, 0] == 255


def test_quantisation_of_a_float64_image_is_done_in_double():
    # N0f8(x::Float64): x * 255 and the rounding in double; the Float32 product of the same values rounds some the other way
    rng = np.random.default_rng(11)
    k = rng.integers(0, 255, 4000)
    vals = np.concatenate([(k + 0.5) / 255.0, np.nextafter((k + 0.5) / 255.0, 0), np.nextafter((k + 0.5) / 255.0, 1),
                           rng.random(4000) * 1.2 - 0.1, [np.nan, np.inf, -np.inf, -0.0, 1.0]])
    img = np.zeros((2, len(vals), 3), np.float64)
    img[0, :, 0] = vals
    img[1, :, 2] = vals[::-1]
    got = rt.to_rgb8(img)
    want = np.rint(rt.clamp01(img) * 255.0).astype(np.uint8)
    assert rt.clamp01(img).dtype == np.float64
    np.testing.assert_array_equal(got, want)
    assert np.any(got != rt.to_rgb8(img.astype(F)))
    L = ffi.lib()
    rgb = np.zeros(12, np.uint8)
    assert L.rt_image_to_rgb8_f64(None, 2, 2, ffi.bptr(rgb)) == -1
    assert L.rt_image_to_rgb8_f64(ffi.dptr(np.zeros(12)), 0, 2, ffi.bptr(rgb)) == -1


def test_png_and_ppm_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    for h, w in ((1, 1), (5, 7), (64, 33)):
        img = (rng.random((h, w, 3)) * 1.4 - 0.2).astype(F)
        rgb = rt.to_rgb8(img)
        p = rt.save(tmp_path / f"a_{h}x{w}.png", img)
        np.testing.assert_array_equal(decode_png(p), rgb)
        q = rt.save(tmp_path / f"a_{h}x{w}.ppm", img)
        data = open(q, "rb").read()
        head = b"P6\n%d %d\n255\n" % (w, h)
        assert data.startswith(head)
        np.testing.assert_array_equal(np.frombuffer(data[len(head):], np.uint8).reshape(h, w, 3), rgb)
    try:
        from PIL import Image
    except ImportError:
        return
    np.testing.assert_array_equal(np.asarray(Image.open(p).convert("RGB")), rgb)


def test_writer_errors(tmp_path):
    from raytracing_amd import _ffi as ffi
    L = ffi.lib()
    rgb = np.zeros(12, np.uint8)
    assert L.rt_write_png(str(tmp_path / "no_such_dir" / "x.png").encode(), ffi.bptr(rgb), 2, 2) == -1
    assert b"cannot open" in L.rt_last_error()
    assert L.rt_write_png(str(tmp_path / "x.png").encode(), ffi.bptr(rgb), 0, 2) == -1
    assert L.rt_image_to_rgb8(None, 2, 2, ffi.bptr(rgb)) == -1
