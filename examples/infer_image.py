#!/usr/bin/env python3
"""examples/infer_image.rs on the MI355X library: PNG in, u8 mask PNG out.

  python examples/infer_image.py photo.png [mask.png] --weights model.safetensors [--compute f32_split2]
  python examples/infer_image.py photo.png --synthetic        # random-init weights (no checkpoint at hand): exercises the path only
  python examples/infer_image.py a.png b.png c.png --batch-out masks/ --weights model.safetensors    # several images: ONE batched call

Every step runs in libbirefnet_hip.so: resize_exact(1024, 1024, Triangle) + ImageNet normalisation (infer_image.rs:44-67),
forward_logits, sigmoid -> u8 -> Lanczos3 resize back to the original size (:84-110).  Only PNG decoding / encoding is Python."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb  # noqa: E402
from candle_birefnet_amd.imageproc import infer_images, postprocess_mask, preprocess_image, read_png, write_png_gray  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("image")
    ap.add_argument("output", nargs="*", default=["output_mask.png"], help="mask PNG (one image), or further input images with --batch-out")
    ap.add_argument("--batch-out", help="directory for the masks of a batch: every positional argument is then an input image (brn_infer_images_u8)")
    ap.add_argument("--weights", help="ZhengPeng7/BiRefNet model.safetensors")
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--compute", default="f32_split2", choices=list(cb.BiRefNet.COMPUTE))
    ap.add_argument("--size", type=int, default=1024)
    a = ap.parse_args()
    if not a.weights and not a.synthetic:
        ap.error("give --weights model.safetensors (or --synthetic)")
    cfg = cb.BiRefNetConfig.swin_l()
    print("Loading model...")
    src = a.weights if a.weights else cb.VarBuilder.from_tensors(cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42))
    if a.batch_out:
        paths = [a.image] + [p for p in a.output if p != "output_mask.png"]
        model = cb.BiRefNet.new(cfg, src, max_batch=len(paths), max_size=(a.size, a.size), compute=a.compute)
        imgs = [read_png(p) for p in paths]
        ch = min(im.shape[2] for im in imgs)
        imgs = [im[:, :, :ch] if ch == 3 else im for im in imgs]           # one channel count per batch (RGB8 or RGBA8)
        os.makedirs(a.batch_out, exist_ok=True)
        t0 = time.perf_counter()
        masks = infer_images(model, imgs, a.size)                           # uploads, resize + normalise, one forward, resize back, downloads
        print(f"{len(paths)} images end to end (u8 in, u8 masks out): {(time.perf_counter() - t0) * 1e3:.2f} ms")
        for pth, mk in zip(paths, masks):
            out = os.path.join(a.batch_out, os.path.splitext(os.path.basename(pth))[0] + "_mask.png")
            write_png_gray(out, mk)
            print(f"Saved mask to: {out}")
        return
    a.output = a.output[0]
    model = cb.BiRefNet.new(cfg, src, max_batch=1, max_size=(a.size, a.size), compute=a.compute)
    print(f"Loading image: {a.image}")
    img = read_png(a.image)
    h, w = img.shape[:2]
    print(f"Original size: {w}x{h}")
    x = preprocess_image(img, a.size)
    print("Running inference...")
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    logits = model.forward_logits(x)
    torch.cuda.synchronize()
    print(f"Inference time: {(time.perf_counter() - t0) * 1e3:.2f} ms")
    lg = logits.float()
    print(f"Logits stats - min: {lg.min().item():.4f}, max: {lg.max().item():.4f}, mean: {lg.mean().item():.4f}")
    mask = postprocess_mask(logits, (h, w))
    write_png_gray(a.output, mask)
    print(f"Saved mask to: {a.output}")


if __name__ == "__main__":
    main()
