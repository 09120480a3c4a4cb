"""The OpenCV pin's inputs and bookkeeping -- numpy only, so that tools/opencv_pin_vectors.py runs on any machine with `cv2` and nothing of this
repository built, and tests/test_opencv_pin_vectors.py regenerates the SAME inputs here for the oracle and the cv-shim.

Every cv:: call the reference makes on the decode path is a case (file:line under /root/reference/src):
    cvtColor RGB2GRAY + adaptiveThreshold(MEAN_C, BINARY, 5, 0)           lib/cimb_translator/CimbReader.cpp:35,41
    filter2D([0 -1 0; -1 4.5 -1; 0 -1 0]) + adaptiveThreshold(.., 7, 0)   CimbReader.cpp:17-27,41
    cvtColor RGB2GRAY + GaussianBlur(3 | 5 | 9 | 17, sigma 0)             lib/extractor/Scanner.h:151-160
    threshold(BINARY | OTSU)                                              Scanner.h:126-130
    getPerspectiveTransform + warpPerspective(INTER_LINEAR)               lib/extractor/Deskewer.h:26-40
    cvtColor YUV2RGB_NV12 / YUV420p2RGB / RGBA2RGB                        lib/cimbar_js/cimbar_recv_js.cpp:102,108,118
    transpose + invert(DECOMP_SVD) + matrix product                       lib/chromatic_adaptation/color_correction.h:33-37
Inputs come from a counter-based generator written out below (splitmix64 over uint64 arrays): no dependence on numpy's own generators."""
import hashlib

import numpy as np

FORMAT_VERSION = 1


def rand_u8(seed, n):
    """n bytes of splitmix64(seed, counter): the same on every numpy"""
    with np.errstate(over="ignore"):
        z = np.arange((n + 7) // 8, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64((seed * 0xD1B54A32D192ED03 + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF)
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return z.view(np.uint8)[:n].copy()


def img_noise(w, h, c, seed):
    return rand_u8(seed, w * h * c).reshape(h, w, c) if c > 1 else rand_u8(seed, w * h).reshape(h, w)


def img_tiles(w, h, seed):
    """8x8 cells, each a random two-level glyph in one of four bright colours on a dark ground, plus +-6 of noise: the look of a cimbar frame
    (edges at every scale the 5x5 / 7x7 means see, ties in the adaptive threshold, saturated sharpen results)"""
    cw, ch = (w + 7) // 8, (h + 7) // 8
    r = rand_u8(seed, cw * ch * 10)
    glyph_bits = np.unpackbits(r[: cw * ch * 8]).reshape(ch, cw, 8, 8)
    colour = r[cw * ch * 8: cw * ch * 9].reshape(ch, cw) & 3
    level = 160 + (r[cw * ch * 9: cw * ch * 10].reshape(ch, cw) % 96)
    pal = np.array([[0, 255, 0], [0, 255, 255], [255, 255, 0], [255, 0, 255]], np.uint16)
    fg = (pal[colour] * level[..., None] // 255).astype(np.uint8)                      # (ch, cw, 3)
    img = np.where(glyph_bits[..., None].astype(bool), fg[:, :, None, None, :], np.uint8(12))   # (ch, cw, 8, 8, 3)
    img = img.transpose(0, 2, 1, 3, 4).reshape(ch * 8, cw * 8, 3)[:h, :w]
    noise = (rand_u8(seed + 1, w * h * 3).reshape(h, w, 3) % 13).astype(np.int16) - 6
    return np.clip(img.astype(np.int16) + noise, 0, 255).astype(np.uint8)


def img_camera(w, h, seed):
    """a dark, slowly varying background with a bright textured quadrilateral in it (what Scanner and Deskewer look at). Integer arithmetic only:
    no libm call may decide a pixel."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.int64)
    tri = lambda v, period: np.abs((v % (2 * period)) - period)                  # a triangle wave, 0 .. period
    bg = (20 + tri(xx, 97) * 20 // 97 + tri(yy + xx // 3, 61) * 14 // 61).astype(np.int16)
    tiles = img_tiles(w, h, seed)
    s = min(w, h)
    cx, cy = w // 2, h // 2
    u = (xx - cx) * 100 + (yy - cy) * 6                                            # a skewed rectangle over the middle ~60 % of the short side
    v = (yy - cy) * 100 - (xx - cx) * 4
    inside = (np.abs(u) < 31 * s) & (np.abs(v) < 30 * s)
    img = np.where(inside[..., None], tiles, np.clip(bg[..., None] + np.array([0, 4, 9], np.int16), 0, 255).astype(np.uint8))
    noise = (rand_u8(seed + 2, w * h * 3).reshape(h, w, 3) % 7).astype(np.int16) - 3
    return np.clip(img.astype(np.int16) + noise, 0, 255).astype(np.uint8)


def capture_bytes(w, h, fmt):
    return w * h * 4 if fmt == 4 else w * h * 3 // 2


def lsm_inputs(seed):
    """(actual, desired) as init_ccm builds them: four observed colour means + white, against the palette + (255,255,255)"""
    pal = np.array([[0, 255, 0], [0, 255, 255], [255, 255, 0], [255, 0, 255], [255, 255, 255]], np.float32)
    order = np.argsort(rand_u8(seed, 4))                       # the unordered_map's row order varies with the frame
    desired = np.concatenate([pal[:4][order], pal[4:]]).astype(np.float32)
    jitter = rand_u8(seed + 7, 15).reshape(5, 3).astype(np.float32)
    tint = np.array([160, 205, 140], np.int64) + rand_u8(seed + 9, 3).astype(np.int64) // 8           # / 256: a camera's colour cast
    actual = (desired.astype(np.int64) * tint // 256 + jitter.astype(np.int64) // 6 + 9).clip(0, 255).astype(np.float32)   # whole numbers, like uint sums / counts
    return actual, desired


# (corners are whole numbers: the reference's Corners holds point<int>, Corners.h:12-20)
# name -> (op, parameters). Sizes are chosen so that Scanner's rule (unit = max(3, nextPow2(min(w, h) * 0.002) + 1)) picks the kernel named.
CASES = {
    "threshold5_tiles": ("threshold", dict(pre=0, img=("tiles", 1024, 1024, 11))),
    "threshold5_noise": ("threshold", dict(pre=0, img=("noise", 1024, 1024, 12))),
    "sharpen_threshold7_tiles": ("threshold", dict(pre=1, img=("tiles", 1024, 1024, 13))),
    "sharpen_threshold7_noise": ("threshold", dict(pre=1, img=("noise", 1024, 1024, 14))),
    "blur3_otsu_1280x720": ("scan", dict(unit=3, img=("camera", 1280, 720, 21))),
    "blur5_otsu_2048x1536": ("scan", dict(unit=5, img=("camera", 2048, 1536, 22))),
    "blur9_otsu_3200x2600": ("scan", dict(unit=9, img=("camera", 3200, 2600, 23))),
    "blur17_otsu_4700x4600": ("scan", dict(unit=17, img=("camera", 4700, 4600, 24))),
    "blur3_noise_1000x701": ("scan", dict(unit=3, img=("noise", 1000, 701, 25))),
    "deskew_1920x1080": ("deskew", dict(img=("camera", 1920, 1080, 31), corners=[530.0, 70.0, 1452.0, 98.0, 501.0, 1001.0, 1470.0, 972.0])),
    "deskew_1280x720_steep": ("deskew", dict(img=("camera", 1280, 720, 32), corners=[300.0, 80.0, 1010.0, 21.0, 260.0, 690.0, 1100.0, 600.0])),
    "nv12_1920x1080": ("cvtcolor", dict(fmt=12, w=1920, h=1080, seed=41)),
    "nv12_70x38": ("cvtcolor", dict(fmt=12, w=70, h=38, seed=42)),
    "yuv420p_1920x1080": ("cvtcolor", dict(fmt=420, w=1920, h=1080, seed=43)),
    "yuv420p_64x40": ("cvtcolor", dict(fmt=420, w=64, h=40, seed=44)),
    "yuv420p_70x38": ("cvtcolor", dict(fmt=420, w=70, h=38, seed=45)),
    "rgba_130x50": ("cvtcolor", dict(fmt=4, w=130, h=50, seed=46)),
    "lsm_a": ("lsm", dict(seed=51)), "lsm_b": ("lsm", dict(seed=52)), "lsm_c": ("lsm", dict(seed=53)), "lsm_d": ("lsm", dict(seed=54)),
}


def make_image(spec):
    kind, w, h, seed = spec
    if kind == "tiles":
        return np.ascontiguousarray(img_tiles(w, h, seed))
    if kind == "noise":
        return np.ascontiguousarray(img_noise(w, h, 3, seed))
    if kind == "camera":
        return np.ascontiguousarray(img_camera(w, h, seed))
    raise ValueError(kind)


def capture(fmt, w, h, seed):
    buf = rand_u8(seed, capture_bytes(w, h, fmt))
    if fmt != 4:
        buf[:w] = np.arange(w) % 256          # every luma value next to random chroma: both saturation ends of the conversion
    return buf


def digest(arr):
    """what the pin file keeps of one output: shape, dtype, SHA-256 of the raw bytes and a small raw crop (for a first look at a mismatch)"""
    arr = np.ascontiguousarray(arr)
    out = {"shape": list(arr.shape), "dtype": str(arr.dtype), "sha256": hashlib.sha256(arr.tobytes()).hexdigest()}
    if arr.ndim >= 2:
        y0, x0 = arr.shape[0] // 3, arr.shape[1] // 3
        crop = arr[y0:y0 + 12, x0:x0 + 16]
        out["crop"] = {"at": [y0, x0], "shape": list(crop.shape), "hex": np.ascontiguousarray(crop).tobytes().hex()}
    else:
        out["values_hex"] = arr.tobytes().hex()
    return out


def run_all(backend, names=None):
    """backend: an object with threshold(img, pre) -> plane bytes; gray_blur(img, unit) -> gray; otsu(img, unit) -> (t, binary);
    deskew(img, corners8) -> frame; cvtcolor(buf, w, h, fmt) -> rgb; lsm(actual, desired) -> 9 floats. Returns {case: {output: digest}}."""
    out = {}
    for name, (op, p) in CASES.items():
        if names is not None and name not in names:
            continue
        if op == "threshold":
            res = {"bitplane": backend.threshold(make_image(p["img"]), p["pre"])}
        elif op == "scan":
            img = make_image(p["img"])
            t, binary = backend.otsu(img, p["unit"])
            res = {"blurred_gray": backend.gray_blur(img, p["unit"]), "otsu_threshold": np.array([t], np.int32), "binary": binary}
        elif op == "deskew":
            res = {"frame": backend.deskew(make_image(p["img"]), np.array(p["corners"], np.float32))}
        elif op == "cvtcolor":
            res = {"rgb": backend.cvtcolor(capture(p["fmt"], p["w"], p["h"], p["seed"]), p["w"], p["h"], p["fmt"])}
        elif op == "lsm":
            a, d = lsm_inputs(p["seed"])
            res = {"ccm_f32": np.asarray(backend.lsm(a, d), np.float32).reshape(9)}
        else:
            raise ValueError(op)
        out[name] = {k: digest(v) for k, v in res.items()}
    return out


def compare(got, want):
    """[(case, output, what differs)] between two run_all results"""
    bad = []
    for name, outs in want.items():
        if name not in got:
            continue
        for key, w in outs.items():
            g = got[name].get(key)
            if g is None:
                bad.append((name, key, "missing"))
            elif g["sha256"] != w["sha256"]:
                where = "crop differs too" if g.get("crop", {}).get("hex") != w.get("crop", {}).get("hex") or g.get("values_hex") != w.get("values_hex") else "crop equal"
                bad.append((name, key, f"sha256 {g['sha256'][:12]} != {w['sha256'][:12]} ({where}; shapes {g['shape']} / {w['shape']})"))
    return bad
