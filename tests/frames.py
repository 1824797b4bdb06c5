"""Deterministic synthetic frames shared by the golden generator, the parity tests and bench.py.

PRNG = splitmix64 in counter mode (value i of stream `seed` = mix(seed + (i+1)*GAMMA)), so any
frame can be regenerated anywhere from (shape, seed) alone - fixtures store outputs only.
SURVEY.md 8(d): seeds are 0x5EED0000 + 16*channel + layer; legal-range v210 is Y in [64,940],
Cb/Cr in [64,960].
"""
import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, n):
    """n 64-bit values of stream `seed` (vectorised splitmix64)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (np.arange(1, n + 1, dtype=np.uint64) * _GAMMA)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def layer_seed(channel, layer):
    return 0x5EED0000 + 16 * channel + layer


def v210_pitch_pixels(width):
    return width + 47 - ((width - 1) % 48)


def v210_pitch_bytes(width):
    return v210_pitch_pixels(width) * 8 // 3


def v210_pack_codes(y, cb, cr, width, height):
    """Pack 10-bit code planes (y: HxW, cb/cr: Hx(W/2), any ints < 1024) into v210 words.
    Pixels beyond `width` (line padding) are zero, as in the reference's fillBuf (v210.ts:206-236)."""
    pitch_px = v210_pitch_pixels(width)
    Y = np.zeros((height, pitch_px), np.uint32)
    U = np.zeros((height, pitch_px // 2), np.uint32)
    V = np.zeros((height, pitch_px // 2), np.uint32)
    Y[:, :width] = y
    U[:, : (width + 1) // 2] = cb
    V[:, : (width + 1) // 2] = cr
    g = pitch_px // 6
    Y = Y.reshape(height, g, 6)
    U = U.reshape(height, g, 3)
    V = V.reshape(height, g, 3)
    w = np.empty((height, g, 4), np.uint32)
    w[..., 0] = (V[..., 0] << 20) | (Y[..., 0] << 10) | U[..., 0]
    w[..., 1] = (Y[..., 2] << 20) | (U[..., 1] << 10) | Y[..., 1]
    w[..., 2] = (U[..., 2] << 20) | (Y[..., 3] << 10) | V[..., 1]
    w[..., 3] = (Y[..., 5] << 20) | (V[..., 2] << 10) | Y[..., 4]
    return np.ascontiguousarray(w.reshape(-1))


def v210_unpack_codes(words, width, height):
    """Inverse of v210_pack_codes -> (y HxW, cb Hx(W/2), cr Hx(W/2))."""
    pitch_px = v210_pitch_pixels(width)
    g = pitch_px // 6
    w = np.asarray(words, np.uint32).reshape(height, g, 4)
    Y = np.empty((height, g, 6), np.uint32)
    U = np.empty((height, g, 3), np.uint32)
    V = np.empty((height, g, 3), np.uint32)
    m = np.uint32(0x3FF)
    U[..., 0], Y[..., 0], V[..., 0] = w[..., 0] & m, (w[..., 0] >> 10) & m, (w[..., 0] >> 20) & m
    Y[..., 1], U[..., 1], Y[..., 2] = w[..., 1] & m, (w[..., 1] >> 10) & m, (w[..., 1] >> 20) & m
    V[..., 1], Y[..., 3], U[..., 2] = w[..., 2] & m, (w[..., 2] >> 10) & m, (w[..., 2] >> 20) & m
    Y[..., 4], V[..., 2], Y[..., 5] = w[..., 3] & m, (w[..., 3] >> 10) & m, (w[..., 3] >> 20) & m
    hw = (width + 1) // 2
    return (Y.reshape(height, -1)[:, :width], U.reshape(height, -1)[:, :hw], V.reshape(height, -1)[:, :hw])


def v210_random(width, height, seed, legal=True):
    """Uniform random v210 frame.  legal: Y in [64,940], C in [64,960]; else all 10-bit codes
    (exercises the saturating conversions)."""
    n_y, n_c = width * height, ((width + 1) // 2) * height
    r = splitmix64(seed, n_y + 2 * n_c)
    if legal:
        y = 64 + (r[:n_y] % np.uint64(877))
        c = 64 + (r[n_y:] % np.uint64(897))
    else:
        y = r[:n_y] % np.uint64(1024)
        c = r[n_y:] % np.uint64(1024)
    y = y.astype(np.uint32).reshape(height, width)
    cb = c[:n_c].astype(np.uint32).reshape(height, -1)
    cr = c[n_c:].astype(np.uint32).reshape(height, -1)
    return v210_pack_codes(y, cb, cr, width, height)


def v210_ramp(width, height):
    """The reference's test-pattern generator semantics (v210.ts:206-236): Y steps once per
    6-pixel group 64..940 wrapping, Cb = Cr = 512.  (numpy restatement, widths % 6 == 0)"""
    assert width % 6 == 0
    groups = width // 6
    k = np.arange(groups * height, dtype=np.uint32).reshape(height, groups)
    yv = 64 + (k % 877)
    y = np.repeat(yv, 6, axis=1)
    c = np.full((height, width // 2), 512, np.uint32)
    return v210_pack_codes(y, c, c, width, height)


def uniform_f32(n, seed, lo=0.0, hi=1.0):
    """n float32 values uniform in [lo, hi): 24 random bits -> exact f32 in [0,1), then an f32 affine map."""
    r = (splitmix64(seed, n) >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
    return (np.float32(lo) + r * np.float32(hi - lo)).astype(np.float32)


def rgba_random(width, height, seed, lo=0.0, hi=1.0, alpha=None):
    """Random float RGBA image HxWx4.  alpha: None -> random in [0,1); float -> constant."""
    img = uniform_f32(width * height * 4, seed, lo, hi).reshape(height, width, 4)
    if alpha is None:
        img[..., 3] = uniform_f32(width * height, seed ^ 0xA1FA, 0.0, 1.0).reshape(height, width)
    else:
        img[..., 3] = np.float32(alpha)
    return np.ascontiguousarray(img)


def rgba_specials(width, height, seed):
    """rgba_random with NaN, +-Inf, -0, denormals, huge values and exact LUT-index ties sprinkled in
    (600 positions, deterministic): what convert_ushort_sat_rte and the table index do with them."""
    img = rgba_random(width, height, seed, -0.2, 1.2)
    flat = img.reshape(-1)
    specials = np.array([np.nan, np.inf, -np.inf, -0.0, 1e-42, -1e-42, 3.0e38, -3.0e38, 1.0, 0.0, 0.5 / 65535,
                         1.5 / 65535, 2.5 / 65535, 65534.5 / 65535], np.float32)
    n = min(600, flat.size // 2)
    idx = (splitmix64(seed ^ 0x5BEC, n) % np.uint64(flat.size)).astype(np.int64)
    flat[idx] = specials[np.arange(n) % specials.size]
    return img


def mask_ramp(width, height):
    """Horizontal ramp mask (r = x/(w-1)) used for transition_wipe."""
    m = np.zeros((height, width, 4), np.float32)
    m[..., 0] = (np.arange(width, dtype=np.float32) / np.float32(max(width - 1, 1)))[None, :]
    m[..., 3] = 1.0
    return m


# ---- the other pack formats (planes as raw uint8 byte arrays, like the node Buffers) -----------------
def pack_pitch(fmt, width):
    """luma samples per line: getPitch() of each reference format"""
    if fmt in ("rgba8", "bgra8"):
        return width
    if fmt == "v210":
        return v210_pitch_pixels(width)
    return width + 7 - ((width - 1) % 8)


def pack_plane_bytes(fmt, width, height):
    p = pack_pitch(fmt, width)
    if fmt == "v210":
        return [v210_pitch_bytes(width) * height]
    if fmt == "yuv422p10":
        return [p * 2 * height, p * height, p * height]
    if fmt == "yuv422p8":
        return [p * height, p * height // 2, p * height // 2]
    if fmt == "yuv420p":
        return [p * height, p * height // 4, p * height // 4]
    if fmt == "nv12":
        return [p * height, p * height // 2]
    return [p * 4 * height]


def pack_random(fmt, width, height, seed):
    """Random planes: every 8-bit code, or every 10-bit code for yuv422p10 (legal and illegal)."""
    sizes = pack_plane_bytes(fmt, width, height)
    out = []
    for i, n in enumerate(sizes):
        if fmt == "yuv422p10":
            v = (splitmix64(seed * 7 + i, n // 2) % np.uint64(1024)).astype(np.uint16)
            out.append(v.view(np.uint8).copy())
        else:
            out.append((splitmix64(seed * 7 + i, n) % np.uint64(256)).astype(np.uint8))
    return out


def pack_ramp(fmt, width, height):
    """numpy restatement of each format's reference test pattern (fillBuf): pinned by sha256 of
    the concatenated planes against the reference run under node (host_maths.json "ramp_fmt")."""
    p = pack_pitch(fmt, width)
    if fmt in ("rgba8", "bgra8"):
        px = np.array([16, 32, 64, 255] if fmt == "rgba8" else [16, 16, 16, 255], np.uint8)
        return [np.tile(px, width * height)]
    pairs = width // 2
    if fmt in ("yuv422p10", "yuv422p8"):
        wide = fmt == "yuv422p10"
        lo, n, dt = (64, 438, np.uint16) if wide else (16, 110, np.uint8)
        k = np.arange(pairs * height, dtype=np.int64).reshape(height, pairs)
        y0 = lo + 2 * (k % n)
        Y = np.full((height, p), 64 if wide else 16, dt)
        Y[:, 0:2 * pairs:2] = y0
        Y[:, 1:2 * pairs:2] = y0 + 1
        C = np.full((height, p // 2), 512 if wide else 128, dt)
        return [Y.reshape(-1).view(np.uint8).copy(), C.reshape(-1).view(np.uint8).copy(), C.reshape(-1).view(np.uint8).copy()]
    # 4:2:0: line pairs, Y0 ramps up on the first line, Y1 ramps down on the second
    hp = height // 2
    k = np.arange(pairs * hp, dtype=np.int64).reshape(hp, pairs)
    y0 = 16 + 2 * (k % 110)
    y1 = 234 - 2 * (k % 110)
    Y = np.full((height, p), 16, np.uint8)
    Y[0::2, 0:2 * pairs:2] = y0
    Y[0::2, 1:2 * pairs:2] = y0 + 1
    Y[1::2, 0:2 * pairs:2] = y1 + 1
    Y[1::2, 1:2 * pairs:2] = y1
    if fmt == "yuv420p":
        C = np.full(hp * (p // 2), 128, np.uint8)
        return [Y.reshape(-1), C, C.copy()]
    return [Y.reshape(-1), np.full(hp * p, 128, np.uint8)]
