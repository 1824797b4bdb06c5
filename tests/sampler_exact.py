"""OpenCL 1.2 section 8.2 (linear filter, normalised coordinates, CLK_ADDRESS_CLAMP) evaluated EXACTLY, as the yardstick
for the one piece of the path no reference run pins: read_imagef(LINEAR) in `transform` / `resize`
(src/process/transform.ts:25-28,54-57, resize.ts:25-28,50-56).

Two things are separated here:
  * the sampling COORDINATES (s, t) are part of the kernels' own f32 arithmetic (fma-dot, divide) and are pinned to
    AMD's device library like every other built-in; `transform_coords` / `resize_coords` reproduce them bit for bit in
    numpy (f32 fma emulated exactly through f64 with a tie fix-up) - `tests/test_sampler_bound.py` proves that by
    re-deriving the oracle's whole output from them;
  * the FILTER: given those f32 coordinates the spec's formula
        T = (1-a)(1-b) T(i0,j0) + a(1-b) T(i1,j0) + (1-a) b T(i0,j1) + a b T(i1,j1),
        u = s w, i0 = floor(u - 0.5), a = frac(u - 0.5)   (likewise v, j0, b), texels outside the image = 0,
    is evaluated in 80-bit long double (64-bit significand: u, a, 1-a exact; the weighted sum good to ~2^-60) and
    rounded ONCE to f32.  `ulp_distance` then measures how far an f32 implementation sits from that.
Test infrastructure only."""
import numpy as np

F32, F64, LD = np.float32, np.float64, np.longdouble


def fma32(a, b, c):
    """round_f32(a * b + c) for f32 arrays, exactly: the product of two f32 is exact in f64; the f64 sum may round, which
    can only change the final f32 rounding when it lands exactly on an f32 tie - TwoSum's error term says which side the
    true value lies on."""
    a, b, c = (np.asarray(x, F32).astype(F64) for x in (a, b, c))
    p = a * b
    s = p + c
    bb = s - p
    err = (p - (s - bb)) + (c - bb)
    bits = s.view(np.uint64) if s.ndim else np.array(s).view(np.uint64)
    tie = (bits & np.uint64(0x1FFFFFFF)) == np.uint64(0x10000000)
    fix = tie & (err != 0)
    if np.any(fix):
        s = np.where(fix, np.nextafter(s, np.where(err > 0, np.inf, -np.inf)), s)
    return s.astype(F32)


def dot3(m0, m1, m2, x, y, z):
    """AMD device-library dot(float3): fma(a2, b2, fma(a1, b1, a0 * b0)) (ph_device.h dot3)"""
    return fma32(m2, z, fma32(m1, y, (np.asarray(m0, F32) * np.asarray(x, F32)).astype(F32)))


def transform_coords(m9, ow, oh, xs, ys):
    """(s, t) of output pixels (xs, ys) as the `transform` kernel computes them (transform.ts:53-55)"""
    m = np.asarray(m9, F32)
    px = (np.asarray(xs, F32) / F32(ow) - F32(0.5)).astype(F32)
    py = (np.asarray(ys, F32) / F32(oh) - F32(0.5)).astype(F32)
    one = np.ones_like(px)
    s = (dot3(m[0], m[1], m[2], px, py, one) + F32(0.5)).astype(F32)
    t = (dot3(m[3], m[4], m[5], px, py, one) + F32(0.5)).astype(F32)
    return s, t


def resize_coords(scale, off_x, off_y, flip_h, flip_v, ow, oh, xs, ys):
    """(s, t) as the `resize` kernel computes them (resize.ts:49-56)"""
    flip = np.array([1.0 if flip_h else 0.0, -1.0 if flip_h else 1.0, 1.0 if flip_v else 0.0, -1.0 if flip_v else 1.0], F32)
    scale, off_x, off_y = F32(scale), F32(off_x), F32(off_y)
    cx = ((F32(-0.5) - off_x) / scale + F32(0.5)).astype(F32)
    cy = ((F32(-0.5) - off_y) / scale + F32(0.5)).astype(F32)
    ox, oy = fma32(cx, flip[1], flip[0]), fma32(cy, flip[3], flip[2])
    mx, my = (flip[1] / scale).astype(F32), (flip[3] / scale).astype(F32)
    s = fma32((np.asarray(xs, F32) / F32(ow)).astype(F32), mx, ox)
    t = fma32((np.asarray(ys, F32) / F32(oh)).astype(F32), my, oy)
    return s, t


def _taps(img, i, j):
    h, w, _ = img.shape
    inside = (i >= 0) & (i < w) & (j >= 0) & (j < h)
    v = img[np.clip(j, 0, h - 1), np.clip(i, 0, w - 1)]
    return np.where(inside[:, None], v, 0).astype(img.dtype)


def sample_pinned_f32(img, s, t):
    """the f32 evaluation this repository pins the filter to (ph_device.h sample_linear)"""
    h, w, _ = img.shape
    u, v = (s * F32(w)).astype(F32), (t * F32(h)).astype(F32)
    fu, fv = (u - F32(0.5)).astype(F32), (v - F32(0.5)).astype(F32)
    flu, flv = np.floor(fu), np.floor(fv)
    i0, j0 = flu.astype(np.int64), flv.astype(np.int64)
    a, b = (fu - flu).astype(F32), (fv - flv).astype(F32)
    oma, omb = (F32(1) - a).astype(F32), (F32(1) - b).astype(F32)
    w00, w10, w01, w11 = ((oma * omb).astype(F32)[:, None], (a * omb).astype(F32)[:, None],
                          (oma * b).astype(F32)[:, None], (a * b).astype(F32)[:, None])
    t00, t10, t01, t11 = _taps(img, i0, j0), _taps(img, i0 + 1, j0), _taps(img, i0, j0 + 1), _taps(img, i0 + 1, j0 + 1)
    acc = ((w00 * t00).astype(F32) + (w10 * t10).astype(F32)).astype(F32)
    acc = (acc + (w01 * t01).astype(F32)).astype(F32)
    return (acc + (w11 * t11).astype(F32)).astype(F32)


def sample_exact(img, s, t):
    """OpenCL 1.2 8.2 with exact weights, summed in long double, rounded once to f32"""
    assert np.finfo(LD).nmant >= 63, "needs x87 80-bit long double"
    h, w, _ = img.shape
    u, v = s.astype(LD) * LD(w), t.astype(LD) * LD(h)  # exact: 24-bit x 12-bit
    fu, fv = u - LD(0.5), v - LD(0.5)
    flu, flv = np.floor(fu), np.floor(fv)
    i0, j0 = flu.astype(np.int64), flv.astype(np.int64)
    a, b = (fu - flu)[:, None], (fv - flv)[:, None]
    im = img.astype(LD)
    t00, t10, t01, t11 = _taps(im, i0, j0), _taps(im, i0 + 1, j0), _taps(im, i0, j0 + 1), _taps(im, i0 + 1, j0 + 1)
    r = (1 - a) * (1 - b) * t00 + a * (1 - b) * t10 + (1 - a) * b * t01 + a * b * t11
    return r.astype(F32), r


def ulp_distance(x, y):
    """distance in units in the last place between f32 arrays (ordered-integer metric; +0 and -0 coincide)"""
    def key(z):
        k = np.ascontiguousarray(z, F32).view(np.int32).astype(np.int64)
        return np.where(k < 0, -(k & 0x7FFFFFFF), k)
    return np.abs(key(x) - key(y))


def bound_report(got_f32, exact_f32, exact_ld, contrast):
    """how far an implementation's samples sit from the exact formula.  `contrast` = per sample, the largest absolute
    difference between the four texels (the quantity every weight error multiplies)."""
    d = ulp_distance(got_f32, exact_f32)
    abs_err = np.abs(got_f32.astype(LD) - exact_ld).astype(F64)
    rel = abs_err.max(axis=1) / np.maximum(contrast, 1e-30)
    return {"samples": int(got_f32.shape[0]), "max_ulp": int(d.max()), "p999_ulp": float(np.quantile(d, 0.999)),
            "median_ulp": float(np.median(d)), "share_exact": float((d == 0).mean()), "share_within_1_ulp": float((d <= 1).mean()),
            "max_abs_error": float(abs_err.max()), "max_abs_error_over_texel_contrast": float(rel.max())}


def contrast_of(img, s, t):
    h, w, _ = img.shape
    fu, fv = s.astype(F64) * w - 0.5, t.astype(F64) * h - 0.5
    i0, j0 = np.floor(fu).astype(np.int64), np.floor(fv).astype(np.int64)
    im = img.astype(F64)
    taps = np.stack([_taps(im, i0, j0), _taps(im, i0 + 1, j0), _taps(im, i0, j0 + 1), _taps(im, i0 + 1, j0 + 1)])
    return (taps.max(axis=0) - taps.min(axis=0)).max(axis=1)
