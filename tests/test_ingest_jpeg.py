"""The ingest library's JPEG decoder (csrc/jpeg.inc -> cimbar_jpeg_decode): cv::imread hands JPEG files to libjpeg(-turbo) with its default
settings, so "the" pixels of a JPEG are libjpeg's -- the decoder restates that pipeline (Huffman sequential DCT, ISLOW inverse DCT, fancy chroma
upsampling, the 16-bit YCbCr tables) and is held bit-equal to Pillow's libjpeg-turbo here, over qualities, sampling modes, restart intervals,
odd sizes and gray images. Also the PNG forms the device kernels leave to the host (Adam7, 16-bit, sub-byte), and -- on the GPU -- files of
all kinds through both ingest modes."""
import io

import numpy as np
import pytest
from PIL import Image, features

from libcimbar_amd import decoder, ingest
from tests import frames as F

pytestmark = pytest.mark.skipif(not features.check_feature("libjpeg_turbo"), reason="the reference pixels are libjpeg-turbo's: Pillow here is built on another libjpeg")


def jpeg_bytes(arr, **kw):
    buf = io.BytesIO()
    Image.fromarray(arr).save(buf, format="JPEG", **kw)
    return buf.getvalue()


def pillow_rgb(data):
    return np.array(Image.open(io.BytesIO(data)).convert("RGB"))


@pytest.mark.parametrize("subsampling", [0, 1, 2], ids=["4:4:4", "4:2:2", "4:2:0"])
@pytest.mark.parametrize("quality", [30, 75, 90, 95, 100])
def test_jpeg_frames_decode_to_libjpegs_pixels(synth, subsampling, quality):
    _, fr = F.clean_frames(synth, 1, seed=1000 + quality)
    data = jpeg_bytes(fr[0], quality=quality, subsampling=subsampling)
    assert (ingest.jpeg_decode(data) == pillow_rgb(data)).all()


def test_jpeg_odd_sizes_gray_restarts_and_optimised_tables(synth):
    _, fr = F.clean_frames(synth, 1, seed=77)
    g = np.random.default_rng(12)
    noise = g.integers(0, 256, (67, 41, 3), dtype=np.uint8)
    cases = [(fr[0][:333, :517], dict(quality=85, subsampling=2)), (fr[0][:9, :7], dict(quality=85, subsampling=2)), (fr[0][:1, :1], dict(quality=85, subsampling=1)),
             (fr[0][:17, :33], dict(quality=60, subsampling=1)), (noise, dict(quality=75, subsampling=2)), (noise, dict(quality=100, subsampling=0)),
             (noise, dict(quality=5, subsampling=2)), (g.integers(0, 256, (64, 48), dtype=np.uint8), dict(quality=80)),
             (fr[0][:200, :200], dict(quality=85, subsampling=2, restart_marker_blocks=3)), (fr[0][:200, :200], dict(quality=85, subsampling=0, restart_marker_rows=1)),
             (fr[0][:200, :200], dict(quality=85, subsampling=0, optimize=True)), (np.full((40, 40, 3), 255, np.uint8), dict(quality=90)),
             (np.zeros((40, 24, 3), np.uint8), dict(quality=90, subsampling=2))]
    for arr, kw in cases:
        data = jpeg_bytes(arr, **kw)
        got = ingest.jpeg_decode(data)
        want = pillow_rgb(data)
        assert got.shape == want.shape and (got == want).all(), (arr.shape, kw, int((got != want).sum()))
        assert (ingest.image_decode(data) == want).all()


def test_jpeg_decoder_refuses_what_it_does_not_handle(synth):
    _, fr = F.clean_frames(synth, 1, seed=3)
    prog = jpeg_bytes(fr[0][:64, :64], quality=80, progressive=True)
    with pytest.raises(decoder.CimbarHipError):
        ingest.jpeg_decode(prog)                       # SOF2: not this decoder's (the file is skipped like an unreadable one)
    base = jpeg_bytes(fr[0][:64, :64], quality=80)
    for cut in (2, 20, len(base) // 2):
        try:
            ingest.jpeg_decode(base[:cut])            # truncated: an error or libjpeg-like garbage in the missing part, never a crash
        except decoder.CimbarHipError:
            pass
    with pytest.raises(decoder.CimbarHipError):
        ingest.jpeg_decode(b"\xff\xd8" + b"\x00" * 64)


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "P", "LA", "I;16", "1"])
def test_interlaced_pngs_are_put_back_together(mode):
    """Adam7 (ISO/IEC 15948 8.2): seven reduced images, each with its own scanline filters"""
    g = np.random.default_rng(6)
    for (w, h) in ((67, 41), (1, 1), (8, 8), (5, 3), (9, 17)):
        if mode == "I;16":
            arr = g.integers(0, 65536, (h, w), dtype=np.uint16)
            im = Image.fromarray(arr)
            want = np.repeat((arr >> 8).astype(np.uint8)[:, :, None], 3, 2)
        elif mode == "1":
            arr = g.integers(0, 2, (h, w), dtype=np.uint8) * 255
            im = Image.fromarray(arr).convert("1")
            want = np.repeat(arr[:, :, None], 3, 2)
        else:
            rgb = g.integers(0, 256, (h, w, 4), dtype=np.uint8)
            im = Image.fromarray(rgb, "RGBA").convert(mode) if mode != "P" else Image.fromarray(rgb[:, :, :3], "RGB").quantize(64)
            want = np.array(im.convert("RGB"))
        plain = io.BytesIO()
        im.save(plain, format="PNG")
        laced = interlace_png(plain.getvalue())
        assert np.array(Image.open(io.BytesIO(laced)).convert("RGB") if mode not in ("I;16",) else want).shape == want.shape
        got = ingest.png_decode(laced)
        assert got.shape == want.shape and (got == want).all(), (mode, w, h)


def interlace_png(png):
    """re-encode a non-interlaced PNG as Adam7 (Pillow cannot write interlaced files): same IHDR but interlace = 1, the seven passes' scanlines
    (filter 0, and Sub on every other row so that un-filtering inside a pass is exercised) deflated into one IDAT"""
    import struct
    import zlib
    assert png[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(png):
        n = struct.unpack(">I", png[pos:pos + 4])[0]
        chunks.append((png[pos + 4:pos + 8], png[pos + 8:pos + 8 + n]))
        pos += 12 + n
    ihdr = dict(chunks)[b"IHDR"]
    w, h, depth, ctype = struct.unpack(">IIBB", ihdr[:10])
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bits = channels * depth
    row_bytes = (w * bits + 7) // 8
    raw = zlib.decompress(b"".join(d for t, d in chunks if t == b"IDAT"))
    rows = []
    prev = bytearray(row_bytes)
    bpp = max(1, bits // 8)
    for y in range(h):                                    # un-filter the source
        ft = raw[y * (row_bytes + 1)]
        cur = bytearray(raw[y * (row_bytes + 1) + 1:(y + 1) * (row_bytes + 1)])
        for i in range(row_bytes):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 1:
                cur[i] = (cur[i] + a) & 255
            elif ft == 2:
                cur[i] = (cur[i] + b) & 255
            elif ft == 3:
                cur[i] = (cur[i] + ((a + b) >> 1)) & 255
            elif ft == 4:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                cur[i] = (cur[i] + (a if pa <= pb and pa <= pc else (b if pb <= pc else c))) & 255
        rows.append(bytes(cur))
        prev = cur

    def pixel(y, x):                                      # the `bits` bits of pixel (x, y) as an int
        if bits >= 8:
            return rows[y][x * bits // 8:(x + 1) * bits // 8]
        per = 8 // bits
        return (rows[y][x // per] >> ((per - 1 - x % per) * bits)) & ((1 << bits) - 1)

    out = bytearray()
    for (x0, y0, dx, dy) in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        xs, ys = range(x0, w, dx), range(y0, h, dy)
        if not len(xs) or not len(ys):
            continue
        for k, y in enumerate(ys):
            if bits >= 8:
                line = bytearray(b"".join(pixel(y, x) for x in xs))
            else:
                per = 8 // bits
                line = bytearray((len(xs) * bits + 7) // 8)
                for i, x in enumerate(xs):
                    line[i // per] |= pixel(y, x) << ((per - 1 - i % per) * bits)
            if k % 2:                                     # Sub filter on odd rows of the pass
                f = bytearray(line)
                for i in range(len(line) - 1, bpp - 1, -1):
                    f[i] = (line[i] - line[i - bpp]) & 255
                out += b"\x01" + bytes(f)
            else:
                out += b"\x00" + bytes(line)

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    res = png[:8] + chunk(b"IHDR", ihdr[:12] + b"\x01")
    for t, d in chunks:
        if t in (b"PLTE", b"tRNS"):
            res += chunk(t, d)
    return res + chunk(b"IDAT", zlib.compress(bytes(out))) + chunk(b"IEND", b"")


@pytest.mark.gpu
@pytest.mark.parametrize("device_png", [False, True], ids=["host-pool", "device-png"])
def test_files_of_every_kind_through_the_ingest_pipeline(tmp_path, synth, hip_decoder, device_png):
    """what cv::imread takes: PNGs (plain, interlaced, 16-bit) and JPEGs in one list -- in device mode the kernels take the plain PNGs and
    the host threads the rest, which join the batch on the device; every frame's chunks equal the batch decode of the same pixels"""
    payload, frames = F.clean_frames(synth, 12, seed=88)
    paths, pixels = [], []
    for k in range(12):
        kind = k % 4
        p = tmp_path / (f"f{k:02d}.jpg" if kind == 1 else f"f{k:02d}.png")
        if kind == 0:
            Image.fromarray(frames[k]).save(p, compress_level=1)
            px = frames[k]
        elif kind == 1:
            Image.fromarray(frames[k]).save(p, format="JPEG", quality=95, subsampling=0)
            px = np.array(Image.open(p).convert("RGB"))
        elif kind == 2:
            buf = io.BytesIO()
            Image.fromarray(frames[k]).save(buf, format="PNG", compress_level=1)
            p.write_bytes(interlace_png(buf.getvalue()))
            px = frames[k]
        else:
            gray16 = (frames[k][:, :, 1].astype(np.uint16) << 8) | 0x55
            Image.fromarray(gray16).save(p)
            px = np.repeat(frames[k][:, :, 1:2], 3, 2)
        paths.append(str(p))
        pixels.append(px)
    pixels = np.ascontiguousarray(np.stack(pixels))
    hip_decoder.reset_ccm()
    want_total, want_chunks, want_masks = hip_decoder.decode_batch(pixels)
    assert (want_masks[[0, 4, 8]] == 0xFFF).all() and (want_chunks[[0, 4, 8]].reshape(3, -1) == payload[[0, 4, 8]]).all()
    assert (want_masks[[1, 5, 9]] == 0xFFF).all(), "a quality-95 4:4:4 JPEG of a clean frame still decodes completely"
    ing = ingest.Ingest(hip_decoder, threads=4, batch_frames=5, ring=3, png_device=device_png)
    hip_decoder.reset_ccm()
    total, chunks, masks = ing.run_files(paths)
    assert total == want_total and (masks == want_masks).all() and (chunks == want_chunks.reshape(12, -1)).all()
    if device_png:
        assert ing.host_decoded() == 9 and ing.png_stats()["refused_by_host_walk"] == 0 and ing.png_stats()["refused_by_device"] == 0
        assert ing.fallback_overflow() == 0
    ing.close()


@pytest.mark.gpu
def test_host_decoded_files_beyond_a_batchs_fallback_frames_are_counted_as_such(tmp_path, synth, hip_decoder):
    """device PNG mode keeps 32 pinned frames per batch for files the kernels do not take; the 33rd such file of a batch is dropped -- and
    reported by cimbar_ingest_fallback_overflow, not only as one more "refused" file"""
    payload, frames = F.clean_frames(synth, 1, seed=89)
    p = tmp_path / "f.jpg"
    Image.fromarray(frames[0]).save(p, format="JPEG", quality=95, subsampling=0)
    ing = ingest.Ingest(hip_decoder, threads=4, batch_frames=64, ring=2, png_device=True)
    hip_decoder.reset_ccm()
    total, chunks, masks = ing.run_files([str(p)] * 40)
    assert ing.host_decoded() == 32 and ing.fallback_overflow() == 8 and ing.png_stats()["refused_by_host_walk"] == 8
    assert int((masks == 0xFFF).sum()) == 32 and int((masks == 0).sum()) == 8
    ing.close()
