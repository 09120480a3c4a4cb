"""PNG decode on the device (cimbar_hip_png_decode_batch: k_png_inflate + k_png_unfilter) against known pixels and against the host decoder,
one batch holding every case of tests/png_cases.py (all filter types, stored / fixed / dynamic blocks, zlib strategies, small windows, split
IDATs, odd sizes, gray / RGB / RGBA / palette, Pillow's writer, full frames), damaged streams, and the ingest library's device mode end to
end against its host mode."""
import io

import numpy as np
import pytest

from tests import png_cases
from tests.frames import clean_frames

pytestmark = pytest.mark.gpu

# the inflate kernel's variants: one stream per wavefront with the whole deflate window in LDS | with an 8 KiB ring + far matches read back from
# global memory; four streams per wavefront (sixteen lanes each) with a 2 KiB | 4 KiB ring
# "par...": the one-stream kernel with the all-offsets turn (every lane decodes the token that would start at its bit of the window)
VARIANTS = ["32768", "8192", "simt2048", "simt4096", "par8192", "par32768"]


def select(monkeypatch, variant):
    monkeypatch.setenv("CIMBAR_HIP_PNG_PAR", "1" if variant.startswith("par") else "0")
    if variant.startswith("simt"):
        monkeypatch.setenv("CIMBAR_HIP_PNG_SIMT", variant[4:])
    else:
        monkeypatch.setenv("CIMBAR_HIP_PNG_SIMT", "0")
        monkeypatch.setenv("CIMBAR_HIP_PNG_RING", variant[3:] if variant.startswith("par") else variant)


@pytest.mark.parametrize("ring", VARIANTS)
def test_device_png_equals_known_pixels_and_host_decoder(synth, hip_decoder, monkeypatch, ring):
    from libcimbar_amd import decoder, ingest
    select(monkeypatch, ring)
    _p, frames = clean_frames(synth, 1, seed=5151)
    cs = png_cases.cases(frames[0])
    got, status = decoder.png_decode_batch_device([png for _n, png, _w in cs])
    bad = [(name, int(st)) for (name, _png, _w), st in zip(cs, status) if st != 0]
    assert not bad, bad
    for (name, png, want), img in zip(cs, got):
        assert img.shape == want.shape, name
        assert (img == want).all(), (name, np.argwhere(img != want)[:4].tolist())
        assert (ingest.png_decode(png) == img).all(), name


@pytest.mark.parametrize("ring", VARIANTS)
def test_device_png_fuzz(hip_decoder, monkeypatch, ring):
    """600 random small PNGs (size, content, colour type, per-row filters, deflate level / strategy / window / memLevel, IDAT split) in one batch"""
    from libcimbar_amd import decoder
    select(monkeypatch, ring)
    cs = png_cases.fuzz_cases(600, seed=sum(map(ord, ring)))
    got, status = decoder.png_decode_batch_device([png for _n, png, _w in cs])
    bad = [(name, int(st)) for (name, _p, _w), st in zip(cs, status) if st != 0]
    assert not bad, bad[:10]
    wrong = [name for (name, _p, want), img in zip(cs, got) if img.shape != want.shape or not (img == want).all()]
    assert not wrong, wrong[:10]


@pytest.mark.parametrize("ring", VARIANTS)
def test_device_png_refuses_damaged_streams(hip_decoder, monkeypatch, ring):
    from libcimbar_amd import decoder
    select(monkeypatch, ring)
    g = np.random.default_rng(3)
    im = g.integers(0, 256, (40, 30, 3), dtype=np.uint8)
    good = png_cases.make_png(im, [0] * 40)
    bad = [png for _n, png in png_cases.corrupt_cases() if _n != "truncated"]
    pngs = [good] + bad
    got, status = decoder.png_decode_batch_device(pngs)
    assert status[0] == 0 and (got[0] == im).all()
    assert (status[1:] != 0).all(), status.tolist()
    assert set(status[1:].tolist()) <= {decoder.PNG_ESTREAM, decoder.PNG_ECODES, decoder.PNG_ESIZE, decoder.PNG_ECHECK, decoder.PNG_EHEADER}


@pytest.mark.parametrize("ring", VARIANTS)
def test_device_png_truncated_and_garbage_streams_terminate(hip_decoder, monkeypatch, ring):
    """the kernel must come back with an error on streams that end early or are noise (never hang, never write outside its slot)"""
    import ctypes
    select(monkeypatch, ring)
    import torch
    from libcimbar_amd import decoder
    g = np.random.default_rng(5)
    im = g.integers(0, 256, (64, 48, 3), dtype=np.uint8)
    w, h, ct, _d, _i, z, _pal = decoder.png_split(png_cases.make_png(im, g.integers(0, 5, 64), level=6))
    streams = [z, z[:len(z) // 2], z[:7], bytes(g.integers(0, 256, 500, dtype=np.uint8)), b"\x78\x9c" + bytes(300), z[:-4] + b"\0\0\0\0"]
    lib = decoder.load_library()
    n = len(streams)
    desc = (decoder.PngDesc * n)()
    blob = bytearray()
    for i, s in enumerate(streams):
        while len(blob) % 16:
            blob.append(0)
        desc[i].zoff, desc[i].zlen, desc[i].width, desc[i].height, desc[i].color_type = len(blob), len(s), w, h, ct
        blob += s
    while len(blob) % 16:
        blob.append(0)
    dev = torch.device("cuda", 0)
    d_z = torch.from_numpy(np.frombuffer(bytes(blob), np.uint8).copy()).to(dev)
    d_desc = torch.from_numpy(np.frombuffer(bytes(desc), np.uint8).copy()).to(dev)
    ss = int(lib.cimbar_hip_png_scratch_bytes(w, h, ct))
    rs = (w * h * 3 + 15) & ~15
    guard = 4096
    d_scratch = torch.full((n * ss + guard,), 0xA5, dtype=torch.uint8, device=dev)
    d_rgb = torch.full((n * rs + guard,), 0x5A, dtype=torch.uint8, device=dev)
    d_status = torch.full((n,), 777, dtype=torch.int32, device=dev)
    rc = lib.cimbar_hip_png_decode_batch(0, d_z.data_ptr(), d_z.numel(), d_desc.data_ptr(), n, d_scratch.data_ptr(), ss, d_rgb.data_ptr(), rs,
                                         d_status.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    assert rc == 0
    torch.cuda.synchronize(dev)
    st = d_status.cpu().numpy()
    assert st[0] == 0 and (st[1:] != 0).all(), st.tolist()
    assert (d_rgb[:rs][:w * h * 3].cpu().numpy().reshape(h, w, 3) == im).all()
    assert (d_scratch[n * ss:] == 0xA5).all() and (d_rgb[n * rs:] == 0x5A).all()


@pytest.mark.parametrize("simt", ["0", "2048"])       # the inflate kernel the ingest's device mode ends up with: one | four streams per wavefront
def test_ingest_device_png_mode_equals_host_mode(tmp_path, synth, hip_decoder, monkeypatch, simt):
    """files -> libcimbar_ingest.so in device PNG mode (compressed bytes over PCIe, inflate + un-filter on the GPU) == its host mode == the
    payload; unreadable / damaged / wrong-size / 16-bit files deliver nothing in both"""
    from PIL import Image
    from libcimbar_amd import ingest
    monkeypatch.setenv("CIMBAR_HIP_PNG_SIMT", simt)
    payload, frames = clean_frames(synth, 12, seed=909)
    paths = []
    for k in range(12):
        p = tmp_path / f"f{k:03d}.png"
        if k % 3 == 0:
            Image.fromarray(frames[k]).save(p, compress_level=1)
        elif k % 3 == 1:
            Image.fromarray(frames[k]).save(p, compress_level=6)
        else:
            p.write_bytes(png_cases.make_png(frames[k], np.random.default_rng(k).integers(0, 5, frames[k].shape[0]), level=2))
        paths.append(str(p))
    (tmp_path / "broken.png").write_bytes(b"\x89PNG\r\n\x1a\nnope")
    Image.fromarray(frames[0][:512]).save(tmp_path / "small.png")
    dmg = bytearray(open(paths[1], "rb").read())
    dmg[dmg.index(b"IDAT") + 1000] ^= 0x10           # (inside the first IDAT chunk's data: the chunk walk still succeeds)
    (tmp_path / "damaged.png").write_bytes(bytes(dmg))
    mixed = paths[:5] + [str(tmp_path / "broken.png"), str(tmp_path / "missing.png"), str(tmp_path / "small.png"), str(tmp_path / "damaged.png")] + paths[5:]
    host = ingest.Ingest(hip_decoder, threads=4, batch_frames=8, ring=3)
    hip_decoder.reset_ccm()
    th, ch, mh = host.run_files(mixed)
    host.close()
    dev = ingest.Ingest(hip_decoder, threads=4, batch_frames=8, ring=3, png_device=True)
    hip_decoder.reset_ccm()
    td, cd, md = dev.run_files(mixed)
    stats = dev.png_stats()
    dev.close()
    assert td == th == 12 * 7500
    assert (md == mh).all() and (cd == ch).all()
    good = [i for i, p in enumerate(mixed) if p in paths]
    assert (cd[good] == payload.reshape(12, -1)).all()
    assert stats["files"] == len(mixed) and stats["refused_by_host_walk"] == 3 and stats["refused_by_device"] == 1
    assert 0 < stats["bytes_to_device"] < 12 * 1024 * 1024 * 3 // 2


@pytest.mark.parametrize("first", ["pillow_l1", "crop_cvdefault", "black_f0"])
def test_device_chooses_the_inflate_kernel_for_large_launches(synth, hip_decoder, monkeypatch, first):
    """variant 4 ("many images in flight") without overrides: a one-wavefront launch classifies the first image's first block, both inflate kernels
    are enqueued and the one not named returns at once. Whatever it picks -- Pillow's long matches first: four streams per wavefront; a
    cv::imwrite-style stream first: the all-offsets turn; a stream that begins with something else -- every image decodes."""
    from libcimbar_amd import decoder
    for k in ("CIMBAR_HIP_PNG_SIMT", "CIMBAR_HIP_PNG_RING", "CIMBAR_HIP_PNG_PAR"):
        monkeypatch.delenv(k, raising=False)
    _p, frames = clean_frames(synth, 1, seed=5151)
    cs = png_cases.cases(frames[0], big=False)
    lead = [c for c in cs if c[0] == first]
    assert lead, first
    cs = lead + [c for c in cs if c[0] != first]
    got, status = decoder.png_decode_batch_device([png for _n, png, _w in cs], variant=4)
    bad = [(name, int(st)) for (name, _png, _w), st in zip(cs, status) if st != 0]
    assert not bad, bad
    for (name, png, want), img in zip(cs, got):
        assert img.shape == want.shape and (img == want).all(), name


@pytest.mark.parametrize("ring", VARIANTS)
def test_device_png_matches_around_the_ring_boundaries(hip_decoder, monkeypatch, ring):
    from libcimbar_amd import decoder
    select(monkeypatch, ring)
    cs = png_cases.ring_stress_cases()
    got, status = decoder.png_decode_batch_device([png for _n, png, _w in cs])
    bad = [(name, int(st)) for (name, _p, _w), st in zip(cs, status) if st != 0]
    assert not bad, bad[:10]
    wrong = [name for (name, _p, want), img in zip(cs, got) if img.shape != want.shape or not (img == want).all()]
    assert not wrong, wrong[:10]
