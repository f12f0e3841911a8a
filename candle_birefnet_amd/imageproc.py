"""Image pre/post-processing either side of forward_logits (examples/infer_image.rs:44-67, 84-110) on the GPU, plus a
dependency-free PNG reader / writer so that the example runs without an imaging package."""
import ctypes as C
import struct
import zlib

import numpy as np

from . import _ffi
from . import tensors as T


def preprocess_image(pixels, size=1024, device=0, to_device=True):
    """pixels: uint8 [h, w, 3|4] (RGB8 / RGBA8) -> x [1, 3, size, size] fp32, ImageNet-normalised (infer_image.rs:44-67)."""
    a = np.ascontiguousarray(pixels, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] not in (3, 4):
        raise ValueError(f"expected uint8 [h, w, 3|4], got {a.shape}")
    h, w, c = a.shape
    if to_device:
        import torch
        out = torch.empty((1, 3, size, size), dtype=torch.float32, device=f"cuda:{device}")
        _ffi.check(_ffi.lib.brn_preprocess_image(a.ctypes.data, h, w, c, size, out.data_ptr(), _ffi.BRN_MEM_DEVICE, device,
                                                 T.stream_of(out)))
        return out
    out = np.empty((1, 3, size, size), np.float32)
    _ffi.check(_ffi.lib.brn_preprocess_image(a.ctypes.data, h, w, c, size, out.ctypes.data, _ffi.BRN_MEM_HOST, device, None))
    return out


def postprocess_mask(logits, out_hw, apply_sigmoid=True, device=0):
    """logits [.., S, S] (one image; numpy or torch cuda) -> uint8 mask [out_h, out_w] (infer_image.rs:84-110)."""
    S = int(logits.shape[-1])
    if int(logits.shape[-2]) != S or int(np.prod(logits.shape)) != S * S:
        raise ValueError(f"expected one square map, got {tuple(logits.shape)}")
    px, loc, keep, _ = T.as_arg(logits)
    oh, ow = int(out_hw[0]), int(out_hw[1])
    out = np.empty((oh, ow), np.uint8)
    _ffi.check(_ffi.lib.brn_postprocess_mask(px, S, loc, int(bool(apply_sigmoid)), oh, ow, out.ctypes.data,
                                             T.device_of(keep, device), T.stream_of(keep)))
    return out


def infer_images(model, images, size=1024):
    """examples/infer_image.rs:44-110 for a batch, end to end on the device (brn_infer_images_u8): `images` = list of uint8 [h, w, 3|4]
    arrays (all RGB8 or all RGBA8, any sizes) -> list of uint8 masks [h, w].  One forward of the whole batch."""
    arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in images]
    if not arrs or any(a.ndim != 3 or a.shape[2] != arrs[0].shape[2] or a.shape[2] not in (3, 4) for a in arrs):
        raise ValueError("expected a non-empty list of uint8 [h, w, 3|4] images with the same channel count")
    n, ch = len(arrs), int(arrs[0].shape[2])
    outs = [np.empty(a.shape[:2], np.uint8) for a in arrs]
    pix = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
    msk = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    hs = (C.c_int * n)(*[int(a.shape[0]) for a in arrs])
    ws = (C.c_int * n)(*[int(a.shape[1]) for a in arrs])
    _ffi.check(_ffi.lib.brn_infer_images_u8(model._h, n, pix, hs, ws, ch, int(size), msk, None))
    return outs


# ---- minimal PNG codec (8-bit gray / RGB / RGBA, non-interlaced) ----------------------------------------------------
def read_png(path):
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    pos, idat, hdr, plte = 8, [], None, None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or interlace != 0 or ctype not in (0, 2, 3, 4, 6):
        raise ValueError(f"unsupported PNG (depth {depth}, colour type {ctype}, interlace {interlace})")
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, 1 + w * ch)
    out = np.zeros((h, w * ch), np.uint8)
    prev = np.zeros(w * ch, np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft in (1, 3, 4):
            cur = np.zeros(w * ch, np.int32)
            for x in range(w * ch):      # serial filters (sub / average / paeth)
                a = cur[x - ch] if x >= ch else 0
                b = prev[x]
                c = prev[x - ch] if x >= ch else 0
                if ft == 1:
                    p = a
                elif ft == 3:
                    p = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (line[x] + p) & 255
        else:
            raise ValueError("bad PNG filter")
        out[y] = cur
        prev = cur
    img = out.reshape(h, w, ch)
    if ctype == 3:
        img = plte[img[:, :, 0]]
    elif ctype == 0:
        img = np.repeat(img, 3, axis=2)                       # to_rgb8 of a Luma8 image
    elif ctype == 4:
        img = np.concatenate([np.repeat(img[:, :, :1], 3, axis=2), img[:, :, 1:]], axis=2)
    return np.ascontiguousarray(img)


def write_png_gray(path, mask):
    m = np.ascontiguousarray(mask, np.uint8)
    h, w = m.shape
    raw = b"".join(b"\x00" + m[y].tobytes() for y in range(h))

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xffffffff)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                           chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
