"""ORACLE (test infrastructure only; the product never imports this).  numpy restatement of the image pre/post-processing of
examples/infer_image.rs:44-67 and :84-110.

The arithmetic lives in a third-party crate that is NOT under /root/reference: `image` 0.25.9 (Cargo.lock:1102-1104).  Its
resampler (src/imageops/sample.rs) is restated here from the published algorithm:
  resize(img, nw, nh, filter) = horizontal_sample(vertical_sample(img, nh, filter), nw, filter)
  *_sample: ratio = in/out; sratio = max(ratio, 1); src_support = support * sratio; for every output index o:
      inputx = (o + 0.5) * ratio; left = clamp(floor(inputx - src_support), 0, in-1); right = clamp(ceil(inputx + src_support), left+1, in)
      inputx -= 0.5; w_i = kernel((i - inputx) / sratio) for i in [left, right); w /= sum(w); t = sum_i w_i * p_i   (f32, in order)
  vertical_sample keeps f32; horizontal_sample clamps to [0, 255] and rounds to nearest (FloatNearest: half away from zero).
  triangle(x) = 1 - |x| for |x| < 1; lanczos3(x) = sinc(x) sinc(x/3) for |x| < 3, sinc(t) = sin(pi t) / (pi t), sinc(0) = 1.
PARITY UNPINNED: the reference ships no expected images, and the crate cannot be built here (no Rust toolchain).  f32 throughout,
same order of operations as the HIP kernels, so the two agree bit for bit; `sinf` comes from numpy's / glibc's libm on both sides.
"""
import numpy as np

F = np.float32


def _kernel(name, x):
    x = F(x)
    if name == "triangle":
        return F(1.0) - abs(x) if abs(x) < F(1.0) else F(0.0)
    if abs(x) >= F(3.0):
        return F(0.0)

    def sinc(t):
        t = F(t)
        a = F(t * F(np.pi))
        return F(1.0) if t == 0 else F(np.sin(a, dtype=np.float32) / a)
    return F(sinc(x) * sinc(F(x / F(3.0))))


def axis_table(in_n, out_n, name):
    support = F(1.0) if name == "triangle" else F(3.0)
    ratio = F(F(in_n) / F(out_n))
    sratio = F(1.0) if ratio < 1 else ratio
    src_support = F(support * sratio)
    tabs = []
    for o in range(out_n):
        inputx = F((F(o) + F(0.5)) * ratio)
        left = int(np.floor(F(inputx - src_support)))
        left = min(max(left, 0), in_n - 1)
        right = int(np.ceil(F(inputx + src_support)))
        right = min(max(right, left + 1), in_n)
        inputx = F(inputx - F(0.5))
        ws = [_kernel(name, F(F(F(i) - inputx) / sratio)) for i in range(left, right)]
        s = F(0.0)
        for v in ws:
            s = F(s + v)
        tabs.append((left, np.array([F(v / s) for v in ws], np.float32)))
    return tabs


def _sample(img_f32, tabs, axis):
    """weighted sums along `axis` in f32, taps accumulated in order, multiply then add (no fma)"""
    a = np.moveaxis(img_f32, axis, 0)
    out = np.zeros((len(tabs),) + a.shape[1:], np.float32)
    for o, (left, ws) in enumerate(tabs):
        t = np.zeros(a.shape[1:], np.float32)
        for j, wv in enumerate(ws):
            t = (t + (a[left + j] * wv).astype(np.float32)).astype(np.float32)
        out[o] = t
    return np.moveaxis(out, 0, axis)


def resize_u8(img_u8, nw, nh, name):
    """imageops::resize on a [h, w, C] u8 image -> [nh, nw, C] u8"""
    h, w, _ = img_u8.shape
    tmp = _sample(img_u8.astype(np.float32), axis_table(h, nh, name), 0)           # vertical_sample -> Rgba32FImage
    t = _sample(tmp, axis_table(w, nw, name), 1)                                   # horizontal_sample
    t = np.clip(t, F(0), F(255))
    r = np.trunc(t)                                                                # f32::round: half away from zero; t >= 0 here,
    r = r + ((t - r) >= F(0.5))                                                    # and t - trunc(t) is exact in f32 (no t + 0.5 rounding)
    return r.astype(np.uint8)


MEAN = np.array([0.485, 0.456, 0.406], np.float32)
STD = np.array([0.229, 0.224, 0.225], np.float32)


def preprocess(pixels_u8, S=1024):
    """infer_image.rs:44-67 -> [1, 3, S, S] f32"""
    r = resize_u8(pixels_u8, S, S, "triangle")[:, :, :3]                           # resize_exact(...).to_rgb8()
    x = ((r.astype(np.float32) / F(255.0)).astype(np.float32) - MEAN).astype(np.float32) / STD
    return np.ascontiguousarray(x.astype(np.float32).transpose(2, 0, 1))[None]


def postprocess(logits, out_h, out_w, apply_sigmoid=True):
    """infer_image.rs:84-110 -> [out_h, out_w] u8"""
    v = np.asarray(logits, np.float32).reshape(logits.shape[-2], logits.shape[-1])
    if apply_sigmoid:
        with np.errstate(over="ignore"):
            v = (F(1.0) / (F(1.0) + np.exp(-v, dtype=np.float32))).astype(np.float32)
    m = np.clip((v * F(255.0)).astype(np.float32), F(0), F(255)).astype(np.uint8)  # `as u8` truncates
    return resize_u8(m[:, :, None], out_w, out_h, "lanczos3")[:, :, 0]
