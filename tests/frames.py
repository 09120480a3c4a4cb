"""Shared input manufacture for the parity tests (seeded, deterministic)."""
import numpy as np
import torch

from libcimbar_amd import framegen


def clean_frames(synth, n, seed=1234, **kw):
    payload = framegen.synth_payload(n, seed=seed, mode=synth.geo.MODE, **kw)
    frames = synth.frames_from_payload(payload).numpy()
    return payload.numpy(), frames


def tile_error_frames(synth, n, seed=1234, n_errors=99):
    payload = framegen.synth_payload(n, seed=seed, mode=synth.geo.MODE)
    tiles = synth.cell_tiles(payload)
    bad = framegen.inject_cell_errors(tiles, n_errors=n_errors, seed=5678)
    return payload.numpy(), synth.render(bad).numpy()


def add_noise(frame, sigma, seed):
    g = np.random.default_rng(seed)
    return np.clip(frame.astype(np.int16) + g.normal(0, sigma, frame.shape).astype(np.int16), 0, 255).astype(np.uint8)


def shift(frame, dy, dx):
    return np.roll(frame, (dy, dx), (0, 1))


def rescale(frame, grow):
    """grow the image by `grow` pixels and crop the centre: cells drift progressively away from the grid."""
    from PIL import Image
    h, w = frame.shape[:2]
    im = Image.fromarray(frame).resize((w + grow, h + grow), Image.BILINEAR)
    o = grow // 2
    return np.array(im.crop((o, o, o + w, o + h)))


def blank_region(frame, y0, y1, x0, x1, value=0):
    out = frame.copy()
    out[y0:y1, x0:x1] = value
    return out


def distorted_set(synth, seed=77):
    """A labelled list of frames covering: clean, pixel noise, rigid shifts, progressive drift, partial wipe-outs."""
    payload, frames = clean_frames(synth, 4, seed=seed)
    f = frames
    out = [
        ("clean", f[0]),
        ("noise40", add_noise(f[1], 40, 1)),
        ("noise120", add_noise(f[2], 120, 2)),
        ("shift+2+1", shift(f[0], 2, 1)),
        ("shift-3+4", shift(f[1], -3, 4)),
        ("shift+1+0", shift(f[2], 1, 0)),
        ("rescale+6", rescale(f[3], 6)),
        ("rescale+10", rescale(f[0], 10)),
        ("wipe_band", blank_region(f[1], 300, 420, 0, 1024)),
        ("wipe_half", blank_region(f[2], 0, f[2].shape[0], 0, 560, value=255)),
        ("wipe_corner", blank_region(f[3], 60, 500, 60, 700)),
        ("noise200", add_noise(f[3], 200, 3)),
    ]
    return out


def camera_frame(frame, width=1920, height=1080, quad=((500, 40), (1480, 70), (470, 1030), (1500, 1000)), background=96, blur=0.0):
    """A synthetic camera capture: the 1024x1024 `frame` drawn into a width x height RGB image as the quadrilateral `quad`
    (top-left, top-right, bottom-left, bottom-right corners in capture pixels) over a flat background -- what the Scanner / Deskewer
    stage in front of the decoder has to undo."""
    from PIL import Image, ImageFilter
    fh, fw = frame.shape[:2]
    (x0, y0), (x1, y1), (x2, y2), (x3, y3) = quad
    # PIL wants the map output(x,y) -> input: solve the homography that sends the quad's corners to the frame's corners
    src = [(x0, y0), (x1, y1), (x2, y2), (x3, y3)]
    dst = [(0, 0), (fw, 0), (0, fh), (fw, fh)]
    A, b = [], []
    for (x, y), (u, v) in zip(src, dst):
        A.append([x, y, 1, 0, 0, 0, -u * x, -u * y]); b.append(u)
        A.append([0, 0, 0, x, y, 1, -v * x, -v * y]); b.append(v)
    coeffs = np.linalg.solve(np.array(A, dtype=np.float64), np.array(b, dtype=np.float64))
    im = Image.fromarray(frame).convert("RGBA")
    warped = im.transform((width, height), Image.PERSPECTIVE, tuple(coeffs), Image.BILINEAR)
    canvas = Image.new("RGBA", (width, height), (background, background, background, 255))
    canvas.alpha_composite(warped)
    out = canvas.convert("RGB")
    if blur > 0:
        out = out.filter(ImageFilter.GaussianBlur(blur))
    return np.array(out)


def border_images(h, w, seed):
    """images that are busy at their borders (a frame's margin is one flat colour, so frames cannot tell BORDER_REPLICATE from BORDER_REFLECT_101 from
    anything else there): noise, a diagonal ramp, noise with single bright / dark edge columns and rows, black-and-white noise"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    ramp = np.stack([(xx * 3 + yy) & 255, (xx + yy * 5) & 255, (xx * 7 - yy) & 255], -1).astype(np.uint8)
    cols = rng.integers(0, 256, (h, w, 3), dtype=np.uint8) // 4 + 96
    cols[:, -1] = 255; cols[:, -2] = 0; cols[:, -3] = 200; cols[:, 0] = 0; cols[:, 1] = 255; cols[0] = 255; cols[1] = 10; cols[-1] = 0; cols[-2] = 250
    return [rng.integers(0, 256, (h, w, 3), dtype=np.uint8), ramp, cols, rng.integers(0, 2, (h, w, 1), dtype=np.uint8).repeat(3, -1) * 255]
