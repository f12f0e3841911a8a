#!/usr/bin/env python3
"""Where does the logit error of compute mode `bf16` come from?  

The model is cut at the two places the reference's pub fields cut it (bench_inference.rs: backbone / squeeze_module / decoder,
birefnet.rs:380-385) and every piece is run in mode `bf16` on the OUTPUT OF THE PARITY-GRADED MODE (`f32_split3`, 2e-6 from the fp64
golden) of the pieces before it, so that each row isolates one piece's own arithmetic:

  all            bf16 everywhere (what bench.py's c3 / c5 blocks report)
  backbone       bf16 backbone (both passes) + fp32 fusion / squeeze / decoder
  squeeze        fp32 backbone, bf16 squeeze_module, fp32 decoder
  decoder        fp32 backbone + squeeze, bf16 decoder
  sq+dec         fp32 backbone, bf16 squeeze + decoder

The fusion between the pieces (half-scale pass, bilinear up / down, concats: birefnet.rs:423-454) is done here with torch ops on
the GPU in fp32.  Errors are max-abs (and rms) over image 0 of the logits and of the mask (after the sigmoid, birefnet.rs:466-469).

  python tools/bf16_error_attrib.py [--size 1024] [--deform both|reference_cpu|deformable] [--out gpurun_out/attrib.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--deform", default="both")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.nn.functional as F
    import candle_birefnet_amd as cb

    S = args.size

    def up(x, h, w):
        return F.interpolate(x, size=(h, w), mode="bilinear", align_corners=True)

    def fuse(m, x):
        """birefnet.rs:416-454 with the backbone of model m: x1, x2, x3, x4 (context concat included)"""
        f = m.backbone.forward(x)
        fh = m.backbone.forward(up(x, S // 2, S // 2).contiguous())
        xs = [torch.cat([a, up(b, a.shape[2], a.shape[3])], 1).contiguous() for a, b in zip(f, fh)]
        x1, x2, x3, x4 = xs
        h4, w4 = x4.shape[2], x4.shape[3]
        x4 = torch.cat([up(x1, h4, w4), up(x2, h4, w4), up(x3, h4, w4), x4], 1).contiguous()
        return x1, x2, x3, x4

    def stats(y, ref):
        d = (y.double() - ref.double())
        dm = torch.sigmoid(y.double()) - torch.sigmoid(ref.double())
        return {"logit_max": float(d.abs().max()), "logit_rms": float(d.pow(2).mean().sqrt()),
                "mask_max": float(dm.abs().max()), "mask_rms": float(dm.pow(2).mean().sqrt())}

    report = {"size": S, "reference": "f32_split3 on the same box", "rows": {}}
    modes = ["reference_cpu", "deformable"] if args.deform == "both" else [args.deform]
    for dm in modes:
        cfg = cb.BiRefNetConfig(deform_mode=dm)
        w = cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42)
        vb = cb.VarBuilder.from_tensors(w)
        mf = cb.BiRefNet.new(cfg, vb, compute="f32_split3")
        mb = cb.BiRefNet.new(cfg, vb, compute="bf16")
        x = torch.from_numpy(cb.synth_input(1, S, S)).cuda()
        ref = mf.forward_logits(x)
        rows = {}
        rows["all"] = stats(mb.forward_logits(x), ref)
        ff = fuse(mf, x)
        fb = fuse(mb, x)
        # sanity: the piece-wise path in fp32 reproduces forward_logits
        sq_f = mf.squeeze_module.forward(ff[3])
        rows["piecewise_fp32_check"] = stats(mf.decoder.forward(x, ff[0], ff[1], ff[2], sq_f), ref)
        for k, (a, b) in enumerate(zip(fb, ff)):
            rows[f"backbone_feature_x{k + 1}_rel"] = float((a.double() - b.double()).abs().max() / b.double().abs().max())
        sq_fb = mf.squeeze_module.forward(fb[3])
        rows["backbone"] = stats(mf.decoder.forward(x, fb[0], fb[1], fb[2], sq_fb), ref)
        sq_b = mb.squeeze_module.forward(ff[3])
        rows["squeeze_feature_rel"] = float((sq_b.double() - sq_f.double()).abs().max() / sq_f.double().abs().max())
        rows["squeeze"] = stats(mf.decoder.forward(x, ff[0], ff[1], ff[2], sq_b), ref)
        rows["decoder"] = stats(mb.decoder.forward(x, ff[0], ff[1], ff[2], sq_f), ref)
        rows["sq+dec"] = stats(mb.decoder.forward(x, ff[0], ff[1], ff[2], sq_b), ref)
        rows["logit_abs_max_of_reference"] = float(ref.abs().max())
        report["rows"][dm] = rows
        print(f"--- {dm}, {S}x{S}, image 0 ---", flush=True)
        for k, v in rows.items():
            if isinstance(v, dict):
                print(f"  {k:22s} logits max {v['logit_max']:.3e} rms {v['logit_rms']:.3e} | mask max {v['mask_max']:.3e} rms {v['mask_rms']:.3e}", flush=True)
            else:
                print(f"  {k:22s} {v:.3e}", flush=True)
        mf.close(); mb.close()
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(report, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
