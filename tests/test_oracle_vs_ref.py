"""Differential tests: oracle/cimbar_oracle.c against the reference's own code (oracle/_ref). Skipped where _ref is absent."""
import ctypes

import numpy as np
import pytest

from libcimbar_amd import modeb
from oracle import pyref
from oracle.pyref import P
from tests import frames as F


def test_tile_hashes_and_tiles(ref, synth):
    h = (ctypes.c_uint64 * 16)()
    assert ref.ref_tile_hashes(h) == 16
    assert list(h) == [int(x) for x in modeb.TILE_HASHES]
    for c in range(4):
        for s in range(16):
            t = np.zeros(192, np.uint8)
            assert ref.ref_tile_rgb(s, c, 1, P(t)) == 0
            assert (t.reshape(8, 8, 3) == synth.tiles[c * 16 + s].numpy()).all()


def test_interleave_and_rs_encode(ref, oracle):
    a = np.zeros(12400, np.uint32)
    b = np.zeros(12400, np.uint32)
    ref.ref_interleave_reverse(12400, 155, 2, P(a))
    oracle.co_interleave_reverse(P(b))
    assert (a == b).all()
    g = np.random.default_rng(5)
    for _ in range(50):
        msg = g.integers(0, 256, 125, dtype=np.uint8)
        e1, e2 = np.zeros(155, np.uint8), np.zeros(155, np.uint8)
        ref.ref_rs_encode(P(msg), 125, 30, P(e1))
        oracle.co_rs_encode(P(msg), 125, 30, P(e2))
        assert (e1 == e2).all()


def test_rs_decode_fuzz_including_failure_regime(ref, oracle):
    g = np.random.default_rng(6)
    succ = 0
    for t in range(1500):
        msg = g.integers(0, 256, 125, dtype=np.uint8)
        enc = np.zeros(155, np.uint8)
        ref.ref_rs_encode(P(msg), 125, 30, P(enc))
        ne = int(g.integers(0, 30))
        pos = g.choice(155, ne, replace=False)
        bad = enc.copy()
        bad[pos] ^= g.integers(1, 256, ne, dtype=np.uint8)
        o1, o2 = np.zeros(125, np.uint8), np.zeros(125, np.uint8)
        r1 = ref.ref_rs_decode(P(bad), 155, 30, P(o1))
        r2 = oracle.co_rs_decode(P(bad), 155, 30, P(o2))
        assert r1 == r2, (t, ne)
        if r1 > 0:
            assert (o1 == o2).all(), (t, ne)
            succ += 1
    assert 700 < succ < 1000


def test_color_classifier_fuzz(ref, oracle):
    ref.ref_reset_ccm()
    g = np.random.default_rng(8)
    for _ in range(3000):
        r, gg, b = (float(x) for x in g.integers(0, 256, 3))
        assert ref.ref_best_color(r, gg, b) == oracle.co_best_color(r, gg, b, None)


@pytest.mark.parametrize("pre", [0, 1])
def test_stages_on_distorted_frames(ref, oracle, synth, pre):
    for name, frame in F.distorted_set(synth, seed=123):
        frame = np.ascontiguousarray(frame)
        b1, v1 = np.zeros(131072, np.uint8), np.zeros(4 * 12400, np.int32)
        assert ref.ref_symbol_pass(P(frame), 1024, 1024, pre, P(b1), P(v1)) == 12400
        b2, v2 = np.zeros(131072, np.uint8), np.zeros(4 * 12400, np.int32)
        oracle.co_threshold_bitplane(P(frame), 1024, 1024, pre, P(b2))
        assert (b1 == b2).all(), name
        assert oracle.co_symbol_pass(P(b2), P(v2), None) == 12400
        assert (v1 == v2).all(), name


@pytest.mark.parametrize("pre", [0, 1])
def test_threshold_at_busy_borders(ref, oracle, pre):
    """the same on images whose borders are not flat (tests/frames.py border_images): where the box mean's BORDER_REPLICATE and the sharpen filter's
    BORDER_REFLECT_101 decide bits. (The GPU kernels are held to the oracle on these images in tests/test_gpu_k1_strips.py.)"""
    for k, img in enumerate(F.border_images(1024, 1024, 66 + 68)):
        img = np.ascontiguousarray(img)
        b1, v1 = np.zeros(131072, np.uint8), np.zeros(4 * 12400, np.int32)
        assert ref.ref_symbol_pass(P(img), 1024, 1024, pre, P(b1), P(v1)) == 12400
        b2 = np.zeros(131072, np.uint8)
        oracle.co_threshold_bitplane(P(img), 1024, 1024, pre, P(b2))
        assert (b1 == b2).all(), f"image {k}: {(b1 != b2).sum()} bitplane bytes differ"


@pytest.mark.parametrize("cc", [0, 1, 2])
def test_whole_decode_with_ccm_carry(ref, synth, cc):
    # one reference thread decoding a sequence == the oracle carrying its co_ccm through the same sequence
    items = F.distorted_set(synth, seed=321)
    ref.ref_reset_ccm()
    ccm = pyref.CoCcm()
    for name, frame in items:
        r1, c1, m1 = pyref.ref_decode(frame, 0, cc, reset_ccm=0)
        r2, c2, m2, ccm = pyref.oracle_decode(frame, 0, cc, ccm)
        assert (r1, m1) == (r2, m2), name
        assert (c1 == c2).all(), name
        rc = (ctypes.c_float * 9)()
        active = ref.ref_get_ccm(rc)
        assert active == ccm.active, name
        if active:
            assert np.array(list(rc), np.float32).tobytes() == np.array(list(ccm.m), np.float32).tobytes(), name


def test_escrow_writer_entry_point(ref, synth):
    # the literal public API (Decoder::decode_fountain into an escrow_buffer_writer, cimbar_recv_js.cpp:160-188)
    payload, frames = F.clean_frames(synth, 1, seed=2)
    buf = np.zeros(7500, np.uint8)
    assert ref.ref_decode_fountain_escrow(P(np.ascontiguousarray(frames[0])), 1024, 1024, 0, 2, 1, P(buf)) == 7500
    assert (buf == payload[0]).all()


@pytest.mark.parametrize("pre", [0, 1])
def test_plain_decode_no_fountain_path(ref, oracle, synth, pre):
    """Decoder::decode into a plain stream (cimbar.cpp:270-272): failed RS blocks become 125 zero bytes, tellp() is the return value"""
    for name, fr in F.distorted_set(synth):
        r, want = pyref.ref_decode_plain(fr, pre, 2, 1)
        r2, got, ok, _ = pyref.oracle_decode_plain(fr, pre, 2)
        assert r == r2 == 7500, name
        assert (got == want).all(), name
        blocks = got.reshape(60, 125)
        assert all(ok[b] or not blocks[b].any() for b in range(60)), name


CAMERA_CASES = [
    # background, quad (tl, tr, bl, br) in a 1920x1080 capture, blur
    (0, ((500, 40), (1480, 70), (470, 1030), (1500, 1000)), 0.0),
    (255, ((430, 10), (1500, 30), (410, 1060), (1490, 1075)), 0.8),
    (30, ((448, 28), (1472, 28), (448, 1052), (1472, 1052)), 0.0),
    (12, ((520, 60), (1450, 40), (540, 1010), (1430, 1040)), 0.6),
]


@pytest.mark.parametrize("case", range(len(CAMERA_CASES)))
def test_extractor_stage_scan_preprocess_and_deskew(ref, oracle, synth, case):
    """the oracle's restatement of Scanner::preprocess_image and Deskewer::deskew against the reference's own
    Scanner / Deskewer / Extractor code (compiled against the cv-shim), and the whole reference chain capture -> extract -> decode"""
    bg, quad, blur = CAMERA_CASES[case]
    payload, frames = F.clean_frames(synth, 1, seed=50 + case)
    cam = np.ascontiguousarray(F.camera_frame(frames[0], quad=quad, background=bg, blur=blur))
    h, w = cam.shape[:2]
    a, b = np.zeros((h, w), np.uint8), np.zeros((h, w), np.uint8)
    ref.ref_scan_preprocess(P(cam), w, h, P(a))
    assert 0 <= oracle.co_scan_preprocess(P(cam), w, h, P(b)) < 256
    assert (a == b).all()
    corners = (ctypes.c_float * 8)()
    assert ref.ref_scan_corners(P(cam), w, h, corners) == 4
    for (qx, qy), k in zip(quad, range(4)):      # anchor centres sit ~30/1024 of the way in from the quad's corners
        assert abs(corners[2 * k] - qx) < 60 and abs(corners[2 * k + 1] - qy) < 60
    o1, o2, o3 = (np.zeros((1024, 1024, 3), np.uint8) for _ in range(3))
    assert ref.ref_deskew(P(cam), w, h, corners, P(o1)) == 1024
    oracle.co_deskew(P(cam), w, h, corners, P(o2))
    assert (o1 == o2).all()
    rc = ref.ref_extract(P(cam), w, h, P(o3))
    assert rc in (1, 2) and (o3 == o1).all()
    r, chunks, mask = pyref.ref_decode(o3, rc == 2, 2, 1)           # cimbar.cpp:147-171: NEEDS_SHARPEN -> should_preprocess
    r2, chunks2, mask2, _ = pyref.oracle_decode(o2, rc == 2, 2)
    assert (r, mask) == (r2, mask2) and (chunks == chunks2).all()
    if blur < 1.0:
        assert r == 7500 and (chunks.reshape(-1) == payload[0]).all()


def test_cvshim_reproduces_the_reference_prethreshold_decode_test(ref):
    """cimb_translator/test/CimbDecoderTest.cpp:49-75 (testPrethresholdDecode) passes in the reference's CI with a real OpenCV: every tile
    through cvtColor(RGB2GRAY) + adaptiveThreshold(MEAN_C, 9, 0) + mat_to_bitbuffer decodes to itself, centre window, distance 0 -- all 64
    bits of all 16 tiles. The cv-shim's restatement of those two calls gives the same outcome (a behavioural pin, not a bit-level one)."""
    out = (ctypes.c_uint * 48)()
    assert ref.ref_prethreshold_decode_test(out) == 16, list(out)
    assert [out[3 * i] for i in range(16)] == list(range(16))


@pytest.mark.parametrize("seed", range(8))
def test_random_distortions_oracle_equals_reference(ref, synth, seed):
    """random combinations of shift, rescale, noise and wipe-outs: every stage of the oracle against the reference build"""
    rng = np.random.default_rng(1000 + seed)
    payload, frames = F.clean_frames(synth, 1, seed=2000 + seed)
    fr = frames[0]
    if rng.random() < 0.6:
        fr = F.shift(fr, int(rng.integers(-3, 4)), int(rng.integers(-3, 4)))
    if rng.random() < 0.4:
        fr = F.rescale(fr, int(rng.integers(2, 9)))
    if rng.random() < 0.7:
        fr = F.add_noise(fr, float(rng.integers(10, 140)), int(rng.integers(0, 1 << 30)))
    if rng.random() < 0.3:
        y0, x0 = int(rng.integers(0, 900)), int(rng.integers(0, 900))
        fr = F.blank_region(fr, y0, y0 + int(rng.integers(20, 120)), x0, x0 + int(rng.integers(20, 400)), value=int(rng.integers(0, 256)))
    fr = np.ascontiguousarray(fr)
    for pre in (0, 1):
        r, chunks, mask = pyref.ref_decode(fr, pre, 2, 1)
        r2, chunks2, mask2, _ = pyref.oracle_decode(fr, pre, 2)
        assert (r, mask) == (r2, mask2) and (chunks == chunks2).all(), (seed, pre)
        pr, plain = pyref.ref_decode_plain(fr, pre, 2, 1)
        pr2, plain2, _, _ = pyref.oracle_decode_plain(fr, pre, 2)
        assert pr == pr2 and (plain == plain2).all(), (seed, pre)


@pytest.mark.parametrize("size,quad", [((1280, 720), ((300, 10), (990, 20), (290, 700), (1000, 690))),
                                        ((3840, 2160), ((900, 60), (2950, 100), (880, 2090), (2980, 2050))),
                                        ((4000, 2600), ((800, 70), (3300, 120), (780, 2500), (3340, 2460))),      # short side >= 2500 px: 9x9 blur
                                        ((5600, 4600), ((600, 90), (4900, 160), (560, 4480), (4960, 4400)))])     # short side >= 4500 px: 17x17 blur
def test_extractor_stage_other_capture_sizes(ref, oracle, synth, size, quad):
    """720p (3x3 blur, heavy downscale) and 2160p (5x5 blur): preprocessing and warp of the oracle against the reference build"""
    w, h = size
    payload, frames = F.clean_frames(synth, 1, seed=77)
    cam = np.ascontiguousarray(F.camera_frame(frames[0], width=w, height=h, quad=quad, background=0))
    a, b = np.zeros((h, w), np.uint8), np.zeros((h, w), np.uint8)
    ref.ref_scan_preprocess(P(cam), w, h, P(a))
    assert oracle.co_scan_preprocess(P(cam), w, h, P(b)) >= 0
    assert (a == b).all()
    corners = (ctypes.c_float * 8)()
    if ref.ref_scan_corners(P(cam), w, h, corners) != 4:      # (a capture the scanner cannot lock onto still has a well-defined warp)
        corners = (ctypes.c_float * 8)(*[float(v) for p in quad for v in p])
    o1, o2 = np.zeros((1024, 1024, 3), np.uint8), np.zeros((1024, 1024, 3), np.uint8)
    ref.ref_deskew(P(cam), w, h, corners, P(o1))
    oracle.co_deskew(P(cam), w, h, corners, P(o2))
    assert (o1 == o2).all()


def scan_cases(synth, count=14, seed=5):
    """synthetic captures of several sizes / orientations, some too small or partly out of view (the search then fails -- identically)"""
    payload, clean = F.clean_frames(synth, 4, seed=9)
    g = np.random.default_rng(seed)
    out = []
    for it in range(count):
        W, H = [(1920, 1080), (1280, 720), (1080, 1920), (2560, 1440)][it % 4]
        s = min(W, H)
        cx, cy = W / 2 + g.integers(-W // 10, W // 10), H / 2 + g.integers(-H // 12, H // 12)
        half = s * (0.34 + 0.15 * g.random())

        def j():
            return int(g.integers(-s // 25, s // 25))
        quad = ((int(cx - half) + j(), int(cy - half) + j()), (int(cx + half) + j(), int(cy - half) + j()),
                (int(cx - half) + j(), int(cy + half) + j()), (int(cx + half) + j(), int(cy + half) + j()))
        if it % 7 == 3:
            quad = tuple((x + W // 3, y) for x, y in quad)
        cam = np.ascontiguousarray(F.camera_frame(clean[it % 4], width=W, height=H, quad=quad, background=int(g.integers(0, 256)),
                                                  blur=float(g.choice([0, 0, 0.6, 1.2]))))
        if it % 5 == 4:
            cam = F.add_noise(cam, 25, it)
        out.append(cam)
    return out


def test_anchor_scan_and_extract_restatement_match_the_reference(ref, oracle, synth):
    """SURVEY 8(f) rank 2, the anchor search: oracle co_scan_anchors / co_extract == the reference's Scanner::scan / Extractor::extract
    (anchor rectangles, how many were found, the return code, the deskewed frame) on captures that succeed AND on ones that fail"""
    found4 = 0
    for it, cam in enumerate(scan_cases(synth)):
        H, W = cam.shape[:2]
        a, b = np.zeros(16, np.int32), np.zeros(16, np.int32)
        na = ref.ref_scan_anchors(P(cam), W, H, P(a))
        binimg = np.zeros((H, W), np.uint8)
        oracle.co_scan_preprocess(P(cam), W, H, P(binimg))
        nb = oracle.co_scan_anchors(P(binimg), W, H, P(b))
        assert na == nb, it
        k = 4 * min(na, 4)
        assert (a[:k] == b[:k]).all(), it
        o1, o2 = np.zeros((1024, 1024, 3), np.uint8), np.zeros((1024, 1024, 3), np.uint8)
        c8 = (ctypes.c_float * 8)()
        r1 = ref.ref_extract(P(cam), W, H, P(o1))
        r2 = oracle.co_extract(P(cam), W, H, P(o2), c8)
        assert r1 == r2, it
        if r1:
            assert (o1 == o2).all(), it
            found4 += 1
    assert found4 >= 4
