"""Captures in the pixel formats the reference's C ABI takes (cimbar_recv_js.h:17 `format`; web/recv-worker.js:38-47 passes 12 for a
VideoFrame in NV12, 420 for I420, 4 for RGBA): test-side manufacture only. The forward conversion is an ordinary BT.601 limited-range one
(any encoder would do -- what is under test is the way BACK, which the reference delegates to cv::cvtColor)."""
import numpy as np

FORMATS = (3, 4, 12, 420)


def capture_bytes(w, h, fmt):
    return w * h * 3 // 2 if fmt in (12, 420) else w * h * (4 if fmt == 4 else 3)


def rgb_to_format(rgb, fmt, alpha=255):
    """(h, w, 3) uint8 -> flat uint8 array in `fmt`. 420 is laid out as the reference's cvtColor code reads it (COLOR_YUV420p2RGB =
    COLOR_YUV2RGB_YV12: the plane behind Y is V), so that colours survive the round trip."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w = rgb.shape[:2]
    if fmt == 3:
        return rgb.reshape(-1).copy()
    if fmt == 4:
        out = np.empty((h, w, 4), np.uint8)
        out[..., :3] = rgb
        out[..., 3] = alpha
        return out.reshape(-1)
    assert w % 2 == 0 and h % 2 == 0
    f = rgb.astype(np.float64)
    r, g, b = f[..., 0], f[..., 1], f[..., 2]
    y = 16 + (65.481 * r + 128.553 * g + 24.966 * b) / 255
    u = 128 + (-37.797 * r - 74.203 * g + 112.0 * b) / 255
    v = 128 + (112.0 * r - 93.786 * g - 18.214 * b) / 255
    sub = lambda c: c.reshape(h // 2, 2, w // 2, 2).mean(axis=(1, 3))
    q = lambda c: np.clip(np.rint(c), 0, 255).astype(np.uint8)
    yq, uq, vq = q(y), q(sub(u)), q(sub(v))
    if fmt == 12:
        uv = np.stack([uq, vq], axis=-1)
        return np.concatenate([yq.reshape(-1), uv.reshape(-1)])
    return np.concatenate([yq.reshape(-1), vq.reshape(-1), uq.reshape(-1)])
