"""PNG test inputs for the device PNG decoder (csrc/png.hip.inc) and its CPU model (tools/png_model.cpp): images written by Pillow at several
compression levels and colour types, and hand-assembled PNGs that force every filter type, every deflate block type (stored / fixed / dynamic,
zlib's strategies), small windows, long overlapped runs, odd sizes. Each case = (name, png bytes, expected (h, w, 3) RGB array)."""
import io
import struct
import zlib

import numpy as np


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a.astype(np.int32) + b - c
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    return np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))


def filter_rows(img, filters):
    """img (h, w, bpp) uint8, filters[h] in 0..4 -> the filtered scanlines with their type bytes"""
    h, w, bpp = img.shape
    flat = img.reshape(h, w * bpp).astype(np.int32)
    out = bytearray()
    zero = np.zeros(w * bpp, np.int32)
    for y in range(h):
        cur = flat[y]
        up = flat[y - 1] if y else zero
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if w * bpp > bpp else np.zeros_like(cur)
        if w * bpp <= bpp:
            left = np.zeros_like(cur)
        upleft = np.concatenate([np.zeros(bpp, np.int32), up[:-bpp]]) if w * bpp > bpp else np.zeros_like(cur)
        ft = int(filters[y])
        pred = [zero, left, up, (left + up) >> 1, _paeth(left, up, upleft)][ft]
        out.append(ft)
        out += bytes(((cur - pred) & 255).astype(np.uint8))
    return bytes(out)


def make_png(img, filters, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=15, mem=8, ctype=None, palette=None, idat_split=0):
    h, w, bpp = img.shape
    if ctype is None:
        ctype = {1: 0, 3: 2, 4: 6}[bpp]
    raw = filter_rows(img, filters)
    co = zlib.compressobj(level, zlib.DEFLATED, wbits, mem, strategy)
    z = co.compress(raw) + co.flush()
    png = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0))
    if palette is not None:
        png += _chunk(b"PLTE", bytes(np.asarray(palette, np.uint8).reshape(-1)))
    if idat_split:
        for i in range(0, len(z), idat_split):
            png += _chunk(b"IDAT", z[i:i + idat_split])
    else:
        png += _chunk(b"IDAT", z)
    return png + _chunk(b"IEND", b"")


def to_rgb(img, ctype, palette=None):
    if ctype == 2:
        return img
    if ctype == 6:
        return img[..., :3]
    if ctype == 0:
        return np.repeat(img, 3, axis=2)
    pal = np.asarray(palette, np.uint8).reshape(-1, 3)
    idx = img[..., 0]
    return pal[np.minimum(idx, len(pal) - 1) * (idx < len(pal))]


def cases(frame=None, seed=7, big=True):
    """frame: a (1024, 1024, 3) cimbar frame to cut realistic content from (or None: synthetic tile-like content)"""
    g = np.random.default_rng(seed)
    out = []
    if frame is None:
        tile = g.integers(0, 2, (8, 8, 1), dtype=np.uint8) * g.integers(0, 256, (1, 1, 3), dtype=np.uint8)
        frame = np.tile(np.pad(tile, ((0, 1), (0, 1), (0, 0))), (114, 114, 1))[:1024, :1024]
    noise = g.integers(0, 256, (90, 70, 3), dtype=np.uint8)
    crop = np.ascontiguousarray(frame[100:230, 40:300])
    black = np.zeros((70, 400, 3), np.uint8)
    grad = (np.add.outer(np.arange(140), np.arange(260)) % 256).astype(np.uint8)[..., None].repeat(3, 2)
    named = {"crop": crop, "noise": noise, "black": black, "grad": grad}
    # every filter type on every kind of content, default deflate
    for nm, im in named.items():
        h = im.shape[0]
        for ft in range(5):
            out.append((f"{nm}_f{ft}", make_png(im, [ft] * h), im))
        out.append((f"{nm}_fmix", make_png(im, g.integers(0, 5, h)), im))
    # deflate variety on the cimbar crop and the noise (mixed filters)
    mixc, mixn = g.integers(0, 5, crop.shape[0]), g.integers(0, 5, noise.shape[0])
    for lvl in (0, 1, 2, 4, 9):
        out.append((f"crop_l{lvl}", make_png(crop, mixc, level=lvl), crop))
        out.append((f"noise_l{lvl}", make_png(noise, mixn, level=lvl), noise))
    for nm, st in (("fixed", zlib.Z_FIXED), ("huff", zlib.Z_HUFFMAN_ONLY), ("rle", zlib.Z_RLE), ("filtered", zlib.Z_FILTERED)):
        out.append((f"crop_{nm}", make_png(crop, mixc, strategy=st), crop))
        out.append((f"black_{nm}", make_png(black, [0] * black.shape[0], strategy=st), black))
    for wb in (9, 10, 12):
        out.append((f"crop_w{wb}", make_png(crop, mixc, wbits=wb), crop))
    out.append(("crop_mem1", make_png(crop, mixc, mem=1, level=1), crop))          # many small blocks
    out.append(("crop_idat7", make_png(crop, mixc, idat_split=7), crop))           # IDAT in 7-byte pieces
    out.append(("crop_idat4k", make_png(crop, mixc, idat_split=4096), crop))
    # sizes around the 64-row bands and tiny widths
    for (h, w) in ((1, 1), (1, 5), (2, 2), (63, 3), (64, 64), (65, 33), (128, 7), (129, 1), (200, 2)):
        im = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
        out.append((f"rgb_{h}x{w}", make_png(im, g.integers(0, 5, h)), im))
    # other colour types
    rgba = g.integers(0, 256, (77, 50, 4), dtype=np.uint8)
    rgba[..., :3] = crop[:77, :50]
    out.append(("rgba", make_png(rgba, g.integers(0, 5, 77)), to_rgb(rgba, 6)))
    gray = np.ascontiguousarray(crop[:, :, :1])
    out.append(("gray", make_png(gray, g.integers(0, 5, gray.shape[0])), to_rgb(gray, 0)))
    pal = g.integers(0, 256, (16, 3), dtype=np.uint8)
    pimg = g.integers(0, 16, (66, 90, 1), dtype=np.uint8)
    out.append(("palette", make_png(pimg, g.integers(0, 5, 66), ctype=3, palette=pal), to_rgb(pimg, 3, pal)))
    # Pillow's own writer
    try:
        from PIL import Image
        for lvl in (1, 6):
            buf = io.BytesIO()
            Image.fromarray(crop).save(buf, format="PNG", compress_level=lvl)
            out.append((f"pillow_l{lvl}", buf.getvalue(), crop))
        if big:
            for lvl in (1, 6):
                buf = io.BytesIO()
                Image.fromarray(frame).save(buf, format="PNG", compress_level=lvl)
                out.append((f"pillow_frame_l{lvl}", buf.getvalue(), frame))
    except ImportError:
        pass
    # what cv::imwrite writes by default, i.e. the reference encoder's own files (imgcodecs grfmt_png.cpp: PNG_FILTER_SUB on every row,
    # Z_BEST_SPEED, strategy Z_RLE): streams of short literal codes -- the inflate kernels' literal-burst path
    for nm, im in (("crop", crop), ("noise", noise), ("black", black), ("grad", grad)):
        out.append((f"{nm}_cvdefault", make_png(im, [1] * im.shape[0], level=1, strategy=zlib.Z_RLE), im))
    # the Sub / None-only un-filter (rows are prefix sums): widths of 16 k pixels, one lane short of a wavefront (736 = mode Bu), two segments
    # (1040, 2048), None and Sub rows mixed, a single row
    for (hh, ww) in ((130, 256), (37, 736), (5, 1040), (3, 2048), (1, 16), (70, 1024)):
        im = np.ascontiguousarray(np.tile(crop, (1 + hh // crop.shape[0], 1 + ww // crop.shape[1], 1))[:hh, :ww])
        if ww == 736:
            im = g.integers(0, 256, (hh, ww, 3), dtype=np.uint8)
        out.append((f"sub_{hh}x{ww}", make_png(im, [1] * hh, level=1, strategy=zlib.Z_RLE), im))
        out.append((f"subnone_{hh}x{ww}", make_png(im, g.integers(0, 2, hh), level=1), im))
    out.append(("crop_sub_huff", make_png(crop, [1] * crop.shape[0], level=6, strategy=zlib.Z_HUFFMAN_ONLY), crop))   # literals only
    out.append(("crop_sub_fixed", make_png(crop, [1] * crop.shape[0], level=1, strategy=zlib.Z_FIXED), crop))         # 8- and 9-bit literals
    if big:
        out.append(("frame_fmix", make_png(frame, g.integers(0, 5, frame.shape[0]), level=1), frame))
        out.append(("frame_cvdefault", make_png(frame, [1] * frame.shape[0], level=1, strategy=zlib.Z_RLE), frame))
        fn = np.clip(frame.astype(np.int16) + g.normal(0, 30, frame.shape).astype(np.int16), 0, 255).astype(np.uint8)
        out.append(("frame_noisy", make_png(fn, g.integers(0, 5, fn.shape[0]), level=1), fn))
    return out


def corrupt_cases(seed=11):
    """streams a decoder must refuse (never crash on, never run away with): returns (name, png bytes)"""
    g = np.random.default_rng(seed)
    im = g.integers(0, 256, (40, 30, 3), dtype=np.uint8)
    good = make_png(im, [0] * 40)
    out = []
    i = good.index(b"IDAT") + 4
    n = struct.unpack(">I", good[i - 8:i - 4])[0]
    for k in range(12):
        b = bytearray(good)
        pos = i + 2 + int(g.integers(0, n - 6))
        b[pos] ^= 1 << int(g.integers(0, 8))
        out.append((f"bitflip{k}", bytes(b)))
    b = bytearray(good); b[i + 2] = (b[i + 2] & 0xF9) | 0x06
    out.append(("btype3", bytes(b)))
    out.append(("truncated", good[:i + n // 2] + good[i + n:]))
    return out


def fuzz_cases(n, seed):
    """n random small PNGs: random size, content (noise / runs / tiles / gradients / mixtures), colour type, per-row filters and deflate
    parameters (level, strategy, window, memLevel, IDAT split): (name, png, expected RGB)"""
    g = np.random.default_rng(seed)
    out = []
    strategies = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]
    for i in range(n):
        h, w = int(g.integers(1, 150)), int(g.integers(1, 260))
        bpp = int(g.choice([1, 3, 3, 3, 4]))
        kind = int(g.integers(0, 5))
        if kind == 0:
            im = g.integers(0, 256, (h, w, bpp), dtype=np.uint8)
        elif kind == 1:
            im = np.full((h, w, bpp), int(g.integers(0, 256)), np.uint8)
            im[g.integers(0, h, 5), :, :] = g.integers(0, 256, (5, 1, bpp), dtype=np.uint8)
        elif kind == 2:
            t = g.integers(0, 256, (int(g.integers(1, 12)), int(g.integers(1, 12)), bpp), dtype=np.uint8)
            im = np.tile(t, (h // t.shape[0] + 1, w // t.shape[1] + 1, 1))[:h, :w]
        elif kind == 3:
            im = ((np.add.outer(np.arange(h) * int(g.integers(1, 5)), np.arange(w) * int(g.integers(1, 5))) % 256).astype(np.uint8))[..., None].repeat(bpp, 2)
        else:
            im = np.where(g.random((h, w, 1)) < 0.1, g.integers(0, 256, (h, w, bpp), dtype=np.uint8), np.uint8(int(g.integers(0, 256)))).astype(np.uint8)
        im = np.ascontiguousarray(im)
        ctype, pal = {1: 0, 3: 2, 4: 6}[bpp], None
        if bpp == 1 and g.random() < 0.5:
            ctype, pal = 3, g.integers(0, 256, (256, 3), dtype=np.uint8)
        png = make_png(im, g.integers(0, 5, h), level=int(g.integers(0, 10)), strategy=strategies[int(g.integers(0, 5))], wbits=int(g.integers(9, 16)),
                       mem=int(g.integers(1, 10)), ctype=ctype, palette=pal, idat_split=int(g.choice([0, 0, 100, 8192])))
        out.append((f"fuzz{seed}_{i}_{h}x{w}x{bpp}_k{kind}" + ("_pal" if ctype == 3 else ""), png, to_rgb(im, ctype, pal)))
    return out


def ring_stress_cases(seed=11):
    """images whose deflate streams are full of matches from just inside to just outside the kernels' LDS rings (8 KiB ring: distances around
    8192 - 2048 and 8192 - 320; whole-window kernels: around 32768 - 2048 .. 32768), with a few literals between them: rows that repeat k rows
    up with sparse random changes, so that zlib finds its matches at k * (3 w + 1) bytes. What the all-offsets turn must get right: a turn's
    literals are stored before its matches are copied, so a match from far back must not find them where its source was."""
    g = np.random.default_rng(seed)
    out = []
    for (w, k, h) in ((683, 3, 40), (682, 3, 40), (700, 3, 36), (655, 4, 44), (683, 4, 44), (640, 4, 40), (2047, 1, 30), (1366, 2, 30), (1300, 2, 30),
                      (2040, 5, 60), (2047, 5, 60), (2048, 5, 60), (1800, 6, 60), (1820, 6, 60), (1707, 6, 60)):
        for changes in (0.002, 0.02):
            img = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
            for r in range(k, h):
                img[r] = img[r - k]
                m = g.random((w, 3)) < changes
                img[r][m] = g.integers(0, 256, int(m.sum()), dtype=np.uint8)
            for lvl in (1, 9):
                out.append((f"ring_w{w}_k{k}_c{changes}_l{lvl}", make_png(img, [0] * h, level=lvl), img))
    return out
