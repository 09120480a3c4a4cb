"""The host-side ingest library (include/cimbar_ingest.h): its PNG decoder against Pillow on every colour type it claims (CPU), and the
whole pool -> pinned ring -> overlapped H2D -> decode pipeline against the batch entry point (GPU)."""
import base64
import hashlib
import io
import json
import os
import subprocess

import numpy as np
import pytest
import torch
from PIL import Image

from libcimbar_amd import ingest
from tests import frames as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import re
    text = open(os.path.join(ROOT, "include", "cimbar_ingest.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(cimbar_(?:ingest|png|jpeg|image)_[a-z_]+)\s*\(", text)))
    assert declared == sorted(ingest.EXPORTS)
    L = ingest.load_library()
    for name in declared:
        assert hasattr(L, name), name


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "P", "LA", "I;16", "1"])
def test_png_decoder_matches_pillow(mode):
    g = np.random.default_rng(5)
    w, h = 67, 41
    if mode == "I;16":
        arr = g.integers(0, 65536, (h, w), dtype=np.uint16)
        im = Image.fromarray(arr, mode="I;16")
        want = np.repeat((arr >> 8).astype(np.uint8)[:, :, None], 3, 2)
    elif mode == "1":
        arr = g.integers(0, 2, (h, w), dtype=np.uint8) * 255
        im = Image.fromarray(arr).convert("1")
        want = np.repeat(arr[:, :, None], 3, 2)
    else:
        rgb = g.integers(0, 256, (h, w, 4), dtype=np.uint8)
        im = Image.fromarray(rgb, "RGBA").convert(mode) if mode != "P" else Image.fromarray(rgb[:, :, :3], "RGB").quantize(64)
        want = np.array(im.convert("RGB"))      # alpha dropped (not composited), gray replicated, palette looked up
    buf = io.BytesIO()
    im.save(buf, format="PNG")
    got = ingest.png_decode(buf.getvalue())
    assert got.shape == want.shape and (got == want).all()


def test_png_decoder_rejects_what_it_does_not_handle():
    from libcimbar_amd import decoder
    buf = io.BytesIO()
    Image.fromarray(np.zeros((16, 16, 3), np.uint8)).save(buf, format="PNG", interlace=True) if False else None
    with pytest.raises(decoder.CimbarHipError):
        ingest.png_decode(b"not a png at all, just bytes................................")


def test_png_decoder_on_a_full_frame(synth):
    _, frames = F.clean_frames(synth, 1, seed=8)
    buf = io.BytesIO()
    Image.fromarray(frames[0]).save(buf, format="PNG", compress_level=6)
    assert (ingest.png_decode(buf.getvalue()) == frames[0]).all()


@pytest.mark.gpu
def test_ingest_pipeline_equals_batch_decode(tmp_path, synth, hip_decoder):
    payload, frames = F.clean_frames(synth, 21, seed=31)
    frames = frames.copy()
    frames[5] = F.shift(frames[5], 2, 1)                  # one frame through the flood path
    paths = []
    for k in range(21):
        p = tmp_path / f"f{k:03d}.png"
        Image.fromarray(frames[k]).save(p, compress_level=1)
        paths.append(str(p))
    (tmp_path / "broken.png").write_bytes(b"\x89PNG\r\n\x1a\nnope")
    Image.fromarray(frames[0][:512]).save(tmp_path / "small.png")
    paths_bad = paths[:7] + [str(tmp_path / "broken.png"), str(tmp_path / "missing.png"), str(tmp_path / "small.png")] + paths[7:]
    hip_decoder.reset_ccm()
    want_total, want_chunks, want_masks = hip_decoder.decode_batch(frames)
    ing = ingest.Ingest(hip_decoder, threads=4, batch_frames=4, ring=3)       # small batches: the ring wraps several times
    hip_decoder.reset_ccm()
    total, chunks, masks = ing.run_files(paths)
    assert total == want_total and (masks == want_masks).all() and (chunks == want_chunks.reshape(21, -1)).all()
    hip_decoder.reset_ccm()
    total, chunks, masks = ing.run_files(paths_bad)
    good = [k for k in range(len(paths_bad)) if k not in (7, 8, 9)]
    assert (masks[[7, 8, 9]] == 0).all() and not chunks[[7, 8, 9]].any()
    assert (masks[good] == want_masks).all() and (chunks[good] == want_chunks.reshape(21, -1)).all()
    hip_decoder.reset_ccm()
    total, chunks, masks = ing.run_raw(frames)
    assert total == want_total and (masks == want_masks).all() and (chunks == want_chunks.reshape(21, -1)).all()
    t = ing.timings()
    assert t["wall_s"] > 0
    # page-locked source: copied to the device where it lies (no staging threads)
    import torch
    pinned = torch.from_numpy(frames).pin_memory()
    hip_decoder.reset_ccm()
    total, chunks, masks = ing.run_raw(pinned.numpy())
    assert total == want_total and (masks == want_masks).all() and (chunks == want_chunks.reshape(21, -1)).all()
    ing.close()


@pytest.mark.gpu
@pytest.mark.parametrize("device_png", [False, True])       # PNG inflate + un-filter on the host pool | on the GPU
def test_cimbar_cli_shaped_program_decodes_config1_to_the_file(tmp_path, hip_decoder, device_png):
    """BASELINE configs[0] end to end in C++: PNG file -> cimbar_amd_cli (ingest pool, device decode, the reference's sink + zstd writer,
    built against the reference headers) -> the file ./cimbar would have written"""
    exe = os.path.join(ROOT, "oracle", "_ref", "cimbar_amd_cli")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/cimbar_amd_cli not built (needs /root/reference at build time: make -C oracle dropin)")
    fix = json.load(open(os.path.join(ROOT, "tests", "golden", "config1.json")))
    payload = np.frombuffer(base64.b64decode(fix["payload_b64"]), np.uint8).reshape(1, 7500)
    frame = hip_decoder.encode_batch(payload)[0]
    assert hashlib.sha256(frame.tobytes()).hexdigest() == fix["frame_sha256"]
    Image.fromarray(frame).save(tmp_path / "frame_0.png")
    out_dir = tmp_path / "out"
    out_dir.mkdir()
    res = subprocess.run([exe] + (["--device-png"] if device_png else []) + [str(out_dir), str(tmp_path / "frame_0.png")], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    written = out_dir / fix["file"]
    assert written.exists(), (res.stdout, os.listdir(out_dir))
    data = written.read_bytes()
    assert len(data) == fix["file_size"] and hashlib.sha256(data).hexdigest() == fix["file_sha256"]
