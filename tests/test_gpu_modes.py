"""GPU: modes 67 ("Bm", Conf8x8_mini: 1024x720 frames, 112x78 cells, RS(179,143), 12 chunks of 429 bytes; GridConf.h:168-189) and 66 ("Bu",
Conf8x8_micro: 736x637, 80x69 cells, RS(168,135), 6 chunks of 540 bytes; GridConf.h:144-166) through the same C ABI as mode B --
cimbar_hip_create(device, mode) -- against the oracle built for that mode (oracle/libcimbar_oracle_m67.so / _m66.so, pinned to the reference
build in tests/test_modes.py). Bit-exact everywhere."""
import ctypes

import numpy as np
import pytest
import torch

from libcimbar_amd import decoder as D
from libcimbar_amd import framegen, geometry
from oracle import pyref
from oracle.pyref import P
from tests import frames as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[67, 66, 4, 8])
def MODE(request):
    return request.param


@pytest.fixture(scope="module")
def GEO(MODE):
    return geometry.for_mode(MODE)


@pytest.fixture(scope="module")
def dec67(MODE):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = D.HipDecoder(0, MODE)
    yield d
    d.close()


@pytest.fixture(scope="module")
def synth67(MODE):
    return framegen.FrameSynth("cpu", MODE)


def oracle_batch(frames, MODE, pre=0, cc=2):
    ccm = pyref.CoCcm()
    outs = []
    for fr in frames:
        r, chunks, mask, ccm = pyref.oracle_decode(fr, pre, cc, ccm, mode=MODE)
        sym, col, pos = pyref.oracle_stage(mode=MODE)
        outs.append(dict(r=r, chunks=chunks.copy(), mask=mask, sym=sym, col=col, pos=pos, ccm=np.array(list(ccm.m), np.float32), active=ccm.active))
    return outs


def check(dec, frames, pre=0, cc=2, names=None):
    frames = np.ascontiguousarray(np.stack(frames))
    n = len(frames)
    dec.reset_ccm()
    total, chunks, masks = dec.decode_batch(frames, should_preprocess=pre, color_correction=cc)
    MODE = dec.geo.MODE
    want = oracle_batch(frames, MODE, pre, cc)
    sym, col, drift, ccm = dec.tap(D.TAP_SYMBOLS, n), dec.tap(D.TAP_COLORS, n), dec.tap(D.TAP_DRIFT, n), dec.tap(D.TAP_CCM, n)
    xy = dec.geo.cell_positions()
    for k in range(n):
        tag = names[k] if names else str(k)
        w = want[k]
        assert (sym[k] == w["sym"]).all(), f"{tag}: symbols differ in {(sym[k] != w['sym']).sum()} cells"
        assert (xy + drift[k].astype(np.int32) == w["pos"]).all(), f"{tag}: colour positions differ"
        assert ccm[k, :9].tobytes() == w["ccm"].tobytes() or not w["active"], f"{tag}: CCM differs"
        assert bool(ccm[k, 9]) == bool(w["active"]), f"{tag}: CCM active flag"
        assert (col[k] == w["col"]).all(), f"{tag}: colours differ in {(col[k] != w['col']).sum()} cells"
        assert masks[k] == w["mask"], f"{tag}: mask {masks[k]:#x} vs {w['mask']:#x}"
        assert (chunks[k] == w["chunks"]).all(), f"{tag}: chunk bytes differ"
    assert total == sum(w["r"] for w in want)
    return chunks, masks, want


def test_geometry_reported_by_the_library(dec67, MODE):
    g = dec67.geo
    assert (g.MODE, g.IMG_W, g.IMG_H, g.NCELLS, g.CHUNK, g.BLOCKS, g.RS_BLOCK, g.RS_PARITY) == \
        {67: (67, 1024, 720, 8592, 429, 36, 179, 36), 66: (66, 736, 637, 5376, 540, 24, 168, 33), 4: (4, 1024, 1024, 12400, 750, 60, 155, 30),
         8: (8, 1024, 1024, 12400, 875, 70, 155, 30)}[MODE]
    o = (ctypes.c_int32 * 10)()
    pyref.oracle_lib(MODE).co_geometry(o)
    assert list(o) == [MODE, g.IMG_W, g.IMG_H, g.NCELLS, g.CHUNK, g.RS_BLOCK, g.RS_PARITY, g.DIM_X, g.DIM_Y, g.OFFSET]


def test_clean_batch_bit_exact(dec67, synth67, GEO):
    payload, frames = F.clean_frames(synth67, 6, seed=1234)
    chunks, masks, _ = check(dec67, list(frames))
    assert (masks == GEO.FULL_MASK).all()
    assert (chunks.reshape(6, -1) == payload).all()
    assert not dec67.tap(D.TAP_FLOOD, 6).any(), "clean frames must take the parallel path"


def test_bitplane_matches_oracle(dec67, synth67, MODE, GEO):
    _, frames = F.clean_frames(synth67, 2, seed=9)
    frames = [frames[0], F.add_noise(frames[1], 60, 5)]
    O = pyref.oracle_lib(MODE)
    for pre in (0, 1):
        dec67.decode_batch(np.stack(frames), should_preprocess=pre)
        got = dec67.tap(D.TAP_BITPLANE, 2)
        for k in range(2):
            want = np.zeros(GEO.IMG_W * GEO.IMG_H // 8, np.uint8)
            O.co_threshold_bitplane(P(np.ascontiguousarray(frames[k])), GEO.IMG_W, GEO.IMG_H, pre, P(want))
            assert (got[k] == want).all(), f"pre={pre} frame {k}: {(got[k] != want).sum()} bitplane bytes differ"


@pytest.mark.parametrize("pre,cc", [(0, 2), (1, 2), (0, 1), (1, 0)])
def test_distorted_frames_match_oracle(dec67, synth67, pre, cc):
    items = F.distorted_set(synth67)
    check(dec67, [fr for _, fr in items], pre=pre, cc=cc, names=[nm for nm, _ in items])


def test_tile_substitution_errors_are_corrected(dec67, synth67, GEO):
    payload, frames = F.tile_error_frames(synth67, 4, seed=4321, n_errors=GEO.NCELLS // 125)          # 0.8 % of the cells, BASELINE configs[2]
    chunks, masks, _ = check(dec67, list(frames))
    assert (masks == GEO.FULL_MASK).all() and (chunks.reshape(4, -1) == payload).all()
    assert dec67.tap(D.TAP_RS_OK, 4).all()


def test_flood_paths(dec67, synth67):
    """shifted frames certify on the batch-parallel flood, noisy shifted frames fall back to the exact replay: both equal the oracle"""
    _, frames = F.clean_frames(synth67, 4, seed=31)
    fr = [F.shift(frames[0], 2, 1), F.shift(frames[1], -3, 2), F.add_noise(F.shift(frames[2], 1, -2), 50, 7), F.rescale(frames[3], 6)]
    check(dec67, fr)
    assert dec67.tap(D.TAP_FLOOD, 4).all()
    path = dec67.tap(D.TAP_FLOOD_PATH, 4)
    assert set(path.tolist()) <= {1, 2} and path[0] == 2, path          # (which frames certify depends on the grid; the (2,1) shift does in both)


def test_exact_replay_instances(synth67, MODE):
    """every flagged frame through the exact replay, once with the instance batches of this size take and once with the dense one (eight frames
    per CU: visited bits in LDS, priorities in global memory, heap levels 0-11 in LDS) -- the kernel is built per mode, the grid sizes differ"""
    import os
    _, frames = F.clean_frames(synth67, 4, seed=33)
    fr = [F.shift(frames[0], 2, 1), F.shift(frames[1], -3, 2), F.add_noise(F.shift(frames[2], 1, -2), 50, 7), F.rescale(frames[3], 6)]
    for dense in ("0", "1"):
        old = {k: os.environ.get(k) for k in ("CIMBAR_HIP_FLOOD_WAVE", "CIMBAR_HIP_FLOOD_DENSE")}
        os.environ.update(CIMBAR_HIP_FLOOD_WAVE="0", CIMBAR_HIP_FLOOD_DENSE=dense)
        try:
            dec = D.HipDecoder(0, MODE)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        try:
            check(dec, fr)
            assert (dec.tap(D.TAP_FLOOD_PATH, 4) == 1).all()
        finally:
            dec.close()


def test_decode_plain_matches_oracle(dec67, synth67, MODE):
    payload, frames = F.clean_frames(synth67, 3, seed=77)
    frames = [frames[0], F.add_noise(frames[1], 120, 2), F.blank_region(frames[2], 200, 330, 0, 1024)]
    dec67.reset_ccm()
    r, data, ok = dec67.decode_plain_batch(np.stack(frames))
    ccm = pyref.CoCcm()
    tot = 0
    for k, fr in enumerate(frames):
        wr, wdata, wok, ccm = pyref.oracle_decode_plain(fr, 0, 2, ccm, mode=MODE)
        tot += wr
        assert (ok[k] == wok).all() and (data[k] == wdata).all(), k
    assert r == tot and (data[0] == payload[0]).all()


def test_encode_matches_reference_encoder(dec67, synth67, MODE, GEO):
    """E1 + E2 in this mode against FrameSynth(mode), which tests/test_modes.py pins byte-for-byte to the reference's Encoder::encode_next"""
    payload = framegen.synth_payload(5, seed=8, mode=MODE)
    want = synth67.frames_from_payload(payload).numpy()
    got = dec67.encode_batch(payload.numpy())
    assert got.shape == (5, *GEO.FRAME_SHAPE) and (got == want).all()


def test_pipelined_batches(dec67, synth67, GEO):
    payload, frames = F.clean_frames(synth67, 8, seed=5)
    dev = torch.device("cuda:0")
    d_in = [torch.from_numpy(frames[4 * k:4 * k + 4].copy()).to(dev) for k in range(2)]
    d_ch = [torch.zeros((4, GEO.FRAME_BYTES), dtype=torch.uint8, device=dev) for _ in range(2)]
    d_mk = [torch.zeros(4, dtype=torch.int32, device=dev) for _ in range(2)]
    st = torch.cuda.current_stream().cuda_stream
    dec67.reset_ccm()
    for k in range(2):
        dec67.decode_batch_pipelined(d_in[k].data_ptr(), 4, d_ch[k].data_ptr(), d_mk[k].data_ptr(), stream=st)
    dec67.pipeline_wait(stream=st)
    torch.cuda.synchronize()
    got = torch.cat(d_ch).cpu().numpy()
    assert (got == payload).all() and all((m.cpu().numpy() == GEO.FULL_MASK).all() for m in d_mk)


def test_other_sizes_and_both_modes_coexist(dec67, synth67, hip_decoder, synth, GEO):
    # this mode's frame is "too small" for the mode-B decoder: the reference's answer to that is zero chunks with a full mask (CimbReader.cpp:119)
    if GEO.FRAME_SHAPE != (1024, 1024, 3):
        r, ch, m = hip_decoder.decode_frame(np.zeros(GEO.FRAME_SHAPE, np.uint8))
        assert (r, m) == (7500, 0xFFF) and not ch.any()
    pb, fb = F.clean_frames(synth, 2, seed=3)
    pm, fm = F.clean_frames(synth67, 2, seed=3)
    for k in range(2):          # interleaved calls on the two contexts: each keeps its own tables, constants and CCM
        _, cb, mb = hip_decoder.decode_frame(fb[k])
        _, cm, mm = dec67.decode_frame(fm[k])
        assert mb == 0xFFF and mm == GEO.FULL_MASK and (cb.reshape(-1) == pb[k]).all() and (cm.reshape(-1) == pm[k]).all()
    other = D.HipDecoder(0, 5)       # not a mode of Config::temp_conf: its `default:` branch, i.e. mode B (Config.h:41-43)
    assert other.geo.MODE == 68 and other.bufsize() == 7500 and dec67.bufsize() == GEO.FRAME_BYTES
    other.close()


def test_camera_captures_scan_extract_decode(dec67, synth67, MODE, GEO):
    """cimbard_scan_extract_decode in this mode: 1080p captures of the mode's frames -> anchors -> deskew to the mode's size -> decode, against the
    oracle's co_extract + co_decode_fountain built for the same mode"""
    quads = {67: [((300, 150), (1600, 170), (290, 930), (1620, 915)), ((250, 100), (1700, 100), (250, 1000), (1700, 1000)), ((420, 200), (1500, 230), (400, 900), (1480, 880))],
             66: [((400, 60), (1500, 75), (395, 1010), (1510, 1000)), ((380, 40), (1540, 40), (380, 1044), (1540, 1044)), ((500, 120), (1420, 140), (480, 930), (1400, 905))],
             4: [((500, 40), (1480, 70), (470, 1030), (1500, 1000)), ((448, 28), (1472, 28), (448, 1052), (1472, 1052)), ((520, 60), (1450, 40), (540, 1010), (1430, 1040))]}
    quads = quads[4 if MODE == 8 else MODE]
    payload, frames = F.clean_frames(synth67, len(quads), seed=44)
    cams = np.ascontiguousarray(np.stack([F.camera_frame(frames[k], quad=q, background=bg, blur=bl) for k, (q, bg, bl) in enumerate(zip(quads, (0, 40, 255), (0.0, 0.6, 0.0)))]))
    O = pyref.oracle_lib(MODE)
    n, h, w = cams.shape[:3]
    status, corners, out = dec67.extract_batch(cams)
    ccm = pyref.CoCcm()
    want_chunks, want_masks, want_status = [], [], []
    for k in range(n):
        fr = np.zeros(GEO.FRAME_SHAPE, np.uint8)
        c8 = (ctypes.c_float * 8)()
        st = O.co_extract(P(cams[k]), w, h, P(fr), c8)
        want_status.append(st)
        assert status[k] == st and st != 0, (k, status[k], st)
        assert list(corners[k]) == list(c8)
        assert (out[k] == fr).all(), f"capture {k}: {(out[k] != fr).sum()} deskewed bytes differ"
        r, ch, m, ccm = pyref.oracle_decode(fr, 1 if st == 2 else 0, 2, ccm, mode=MODE)
        want_chunks.append(ch.copy()); want_masks.append(m)
    dec67.reset_ccm()
    total, chunks, masks, st2 = dec67.scan_extract_decode_batch(cams)
    assert list(st2) == want_status and list(masks) == want_masks
    for k in range(n):
        assert (chunks[k] == want_chunks[k]).all(), k
    assert (masks == GEO.FULL_MASK).sum() >= 2          # the captures do decode


def test_full_batch_round_trip(dec67, MODE, GEO):
    """1024 frames rendered and decoded in HBM: every payload byte back, every chunk delivered (size-independent property at bench size)"""
    dev = torch.device("cuda:0")
    n = 1024
    payload = framegen.synth_payload(n, seed=99, mode=MODE).to(dev)
    frames = torch.empty((n, *GEO.FRAME_SHAPE), dtype=torch.uint8, device=dev)
    dec67.encode_batch_device(payload.data_ptr(), n, frames.data_ptr())
    chunks = torch.zeros((n, GEO.FRAME_BYTES), dtype=torch.uint8, device=dev)
    masks = torch.zeros(n, dtype=torch.int32, device=dev)
    dec67.reset_ccm()
    dec67.decode_batch_device(frames.data_ptr(), n, chunks.data_ptr(), masks.data_ptr())
    torch.cuda.synchronize()
    assert bool((masks == GEO.FULL_MASK).all()) and bool((chunks == payload).all())


def test_golden_vectors_of_the_reference_build(dec67, synth67, MODE):
    """tests/golden/mode67.json / mode66.json: what the reference build returned for these frames in this mode, decoded as ONE batch on the GPU (the CCM
    carried frame to frame inside the batch like the reference's decode thread carries it)"""
    import hashlib
    import json
    import os
    from oracle.make_golden_modes import cases
    fix = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mode%d.json" % MODE)))
    items = cases(synth67)
    for pre in (0, 1):          # a batch has one should_preprocess flag: the two halves of the list are two batches, the second continuing the first's CCM
        rows = [(it, row) for it, row in zip(items, fix["frames"]) if row["preprocess"] == pre]
        frames = np.ascontiguousarray(np.stack([it[2] for it, _ in rows]))
        if pre == 0:
            dec67.reset_ccm()
        total, chunks, masks = dec67.decode_batch(frames, should_preprocess=pre, color_correction=2)
        for k, (it, row) in enumerate(rows):
            assert hashlib.sha256(frames[k].tobytes()).hexdigest() == row["frame_sha256"]
            assert int(masks[k]) == row["mask"], row["name"]
            assert hashlib.sha256(chunks[k].tobytes()).hexdigest() == row["chunks_sha256"], row["name"]
        assert total == sum(row["good_bytes"] for _, row in rows)


def test_ingest_pipeline_in_other_modes(tmp_path, dec67, synth67, GEO):
    """PNG files of this mode's frames through the host ingest pool (libcimbar_ingest.so sizes its ring from cimbar_hip_geometry); a mode-B sized
    file in the list is skipped like an unreadable one"""
    from PIL import Image
    from libcimbar_amd import ingest
    payload, frames = F.clean_frames(synth67, 9, seed=12)
    paths = []
    for k in range(9):
        p = tmp_path / f"m{k}.png"
        Image.fromarray(frames[k]).save(p, compress_level=1)
        paths.append(str(p))
    Image.fromarray(np.zeros((1024, 1024, 3), np.uint8)).save(tmp_path / "square.png")
    paths.insert(4, str(tmp_path / "square.png"))
    ing = ingest.Ingest(dec67, threads=3, batch_frames=4, ring=2)
    dec67.reset_ccm()
    total, chunks, masks = ing.run_files(paths)
    good = [k for k in range(10) if k != 4]
    assert masks[4] == 0 and not chunks[4].any()
    assert (masks[good] == GEO.FULL_MASK).all() and (chunks[good] == payload).all() and total == 9 * GEO.FRAME_BYTES
    dec67.reset_ccm()
    total, chunks, masks = ing.run_raw(frames)
    assert total == 9 * GEO.FRAME_BYTES and (chunks == payload).all()
    ing.close()
