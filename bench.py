#!/usr/bin/env python3
"""bench.py — images/sec of BiRefNet::forward_logits (Swin-L, 1024x1024) on N MI355X, one process per GPU.

  python bench.py [--gpus N --steps K --warmup W]      (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" = one pass of the hot path over one batch of synthetic images already resident in HBM; the default workload is
BASELINE.json configs[1]: BiRefNetConfig::swin_l(), batch 1 per GPU, 1024x1024, fp32.  The path shards by image
(independent units, no data-path collective): every rank owns a full weight replica and its own images -> weak scaling.
torch.distributed (RCCL) is used for the timing barrier and the max-over-ranks reduction only.

Rank 0 prints ONE JSON line; see DESIGN.md §measurement for how each field is obtained.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# SURVEY.md §8(d) / BASELINE.md §2: sum of 2*M*N*K over every Linear / conv / QK^T / PV the reference executes.
GFLOP_PER_IMAGE = {1024: 2534.9, 2048: 9772.6}
GFLOP_OFFSET_MOD_1024 = 84.0          # offset + modulator convs: computed and discarded on the reference CPU path
PEAK_F32_MFMA_TFLOPS = 157.3           # MI355X_MICROARCH.md, Peak FP32 (matrix), dense
PEAK_BF16_MFMA_TFLOPS = 2500.0         # MI355X_MICROARCH.md, Peak BF16 MFMA, dense
# compute mode -> (bf16 MFMAs per fp32 product or 0 for the fp32 instruction, dtype string)
MODES = {
    "f32": (0, "f32 (v_mfma_f32_32x32x2_f32, exact fp32 operands)"),
    "f32_split3": (6, "f32 storage+accumulate; GEMM operands split error-free into 3 bf16 terms, 6 bf16 MFMAs/product (fp32-equivalent)"),
    "f32_split2": (3, "f32 storage+accumulate; GEMM operands split into 2 bf16 terms (16-bit mantissa), 3 bf16 MFMAs/product"),
    "bf16_operands": (1, "bf16 GEMM operands, f32 storage+accumulate"),
}
GEMM_FAMILIES = ("gemm_dense", "gemm_conv_nhwc", "gemm_gather_nchw", "gemm_deform_nhwc")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--deform-mode", default="reference_cpu", choices=["reference_cpu", "deformable"])
    ap.add_argument("--compute", default="f32_split3", choices=list(MODES),
                    help="arithmetic of the contraction kernels (include/birefnet_hip.h brn_dtype)")
    ap.add_argument("--also", default="f32_split2,f32", help="comma list of other compute modes to time briefly on rank 0 at N=1 ('' = none)")
    ap.add_argument("--profile-steps", type=int, default=2, help="extra steps with per-launch HIP events for the roofline block")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "off", "on"])
    ap.add_argument("--cpu-baseline-size", type=int, default=0, help="image side for the CPU oracle sample (0 = choose)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import numpy as np
    import torch
    import candle_birefnet_amd as cb

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    S, B = args.size, args.batch
    cfg = cb.BiRefNetConfig(deform_mode=args.deform_mode)                   # BiRefNetConfig::swin_l()
    weights = cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42)       # random-init weights of the real architecture
    model = cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(weights), device=local_rank, max_batch=B, max_size=(S, S), compute=args.compute)
    # rank r owns images [r*B, (r+1)*B) of the global batch (seed 1000 + global index)
    x = torch.from_numpy(cb.synth_input(B, S, S, seed0=1000 + rank * B)).cuda()

    y = None
    for _ in range(args.warmup):
        y = model.forward_logits(x)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = model.forward_logits(x)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    finite = bool(torch.isfinite(y).all().item())

    # ---- roofline of the dominant kernel (gemm_f32, all modes), per-launch HIP events on the launch stream ----
    roof = None
    stage_ms = None
    if rank == 0 and args.profile_steps > 0:
        model.set_profiling(True)
        fl = ms = by = 0.0
        launches = 0
        fam_out = {}
        for _ in range(args.profile_steps):
            model.forward_logits(x)
            st = model.last_kernel_stats()
            for k, v in st.items():
                a = fam_out.setdefault(k, {"launches": 0, "ms": 0.0, "gflop": 0.0})
                a["launches"] += v["launches"]; a["ms"] += v["ms"]; a["gflop"] += v["flop"] / 1e9
            for k in GEMM_FAMILIES:
                fl += st[k]["flop"]; ms += st[k]["ms"]; by += st[k]["bytes"]; launches += st[k]["launches"]
            stage_ms = model.last_timings()
        model.set_profiling(False)
        n = args.profile_steps
        achieved = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        npairs = MODES[args.compute][0]
        peak = PEAK_F32_MFMA_TFLOPS if npairs == 0 else PEAK_BF16_MFMA_TFLOPS / npairs
        # HBM-side bytes per launch of the gemm family: PMC counters cannot be read from inside this process; the figure is
        # the committed rocprofv3 measurement of this very command (profiles/README.md), only quoted when the config matches
        traffic, traffic_note = None, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE not collected for this configuration"
        tj = os.path.join(ROOT, "profiles", f"r01b_gemm_traffic_{args.compute.replace('f32_', '')}.json")
        if os.path.exists(tj) and (B, S, args.deform_mode) == (1, 1024, "reference_cpu"):
            t = json.load(open(tj))
            traffic = round((t["hbm_read_gb_x2corrected"] + t["hbm_write_gb"]) * 1e9 / t["gemm_family_dispatches"])
            traffic_note = (f"bytes per launch, gemm family average, from profiles/r01b_pmc_hbm_b1_1024_{args.compute.replace('f32_', '')}.csv (FETCH_SIZE x2 "
                            "gfx950 correction + WRITE_SIZE, separate passes); algorithmic bytes per launch = "
                            f"{round(by / n / max(1, launches // n))}")
        roof = {
            "bound": "mfma",
            "kernel": "gemm family: gemm_f32_kernel" if npairs == 0 else "gemm family: gemm_split_ws_kernel / gemm_split_kernel (+ gemm_f32_kernel for the NCHW-gather convs)",
            "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_note": traffic_note,
            "peak_note": ("fp32 MFMA dense peak" if npairs == 0 else
                          f"bf16 MFMA dense peak 2500 / {npairs} MFMAs per fp32 product; achieved counts ALGORITHMIC 2*M*N*K "
                          f"(= {achieved / PEAK_F32_MFMA_TFLOPS:.2f}x the fp32-MFMA peak of {PEAK_F32_MFMA_TFLOPS})"),
            "launches_per_step": launches // n, "gflop_per_step": round(fl / n / 1e9, 1), "ms_per_step": round(ms / n, 3),
            "avg_launch_ms": round(ms / max(1, launches), 4),
            "algorithmic_gbytes_per_step": round(by / n / 1e9, 2),
            "measured_over": f"{n} profiled step(s) after the timed region (HIP events bracketing every launch)",
            "families": {k: {"launches": v["launches"] // n, "ms": round(v["ms"] / n, 3), "gflop": round(v["gflop"] / n, 1)}
                         for k, v in fam_out.items()},
        }
    barrier()

    # ---- CPU baseline: the oracle (a port, not candle) on this box's host cores, rank 0 at N=1 only ----
    cpu = None
    if rank == 0 and world == 1 and args.cpu_baseline != "off":
        from oracle import oracle as orc
        cs = args.cpu_baseline_size or S
        xs = cb.synth_input(1, cs, cs, seed0=1000)
        t0c = time.perf_counter()
        ref = orc.forward_logits(orc.cfg_from(cfg), weights, xs)
        dt = time.perf_counter() - t0c
        cpu = {"value": round(1.0 / dt, 5), "unit": "images/s", "cores": orc.num_threads(), "kind": "port",
               "sample": f"1 image {cs}x{cs}, full Swin-L forward_logits through the C++ oracle (unfused fp32 restatement, OpenMP), "
                         f"{dt:.1f} s wall; host cpus={os.cpu_count()}",
               "seconds_per_image": round(dt, 2)}
        if cs == S:
            err = np.abs(y[:1].cpu().numpy().astype(np.float64) - ref)
            cpu["gpu_vs_oracle_max_abs_err"] = float(err.max())
            cpu["gpu_vs_oracle_gate_1e-3abs_or_1e-2rel"] = bool(((err <= 1e-3) | (err <= 1e-2 * np.abs(ref))).all())

    # ---- the other compute modes, briefly (rank 0, N=1): same weights, same input, 5 timed steps each ----
    others = None
    if rank == 0 and world == 1 and args.also:
        others = {}
        ref_np = ref if (cpu is not None and (args.cpu_baseline_size or S) == S) else None
        for mode in [m for m in args.also.split(",") if m and m != args.compute]:
            m2 = cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(weights), device=local_rank, max_batch=B, max_size=(S, S), compute=mode)
            for _ in range(2):
                y2 = m2.forward_logits(x)
            torch.cuda.synchronize()
            t0m = time.perf_counter()
            for _ in range(5):
                y2 = m2.forward_logits(x)
            torch.cuda.synchronize()
            dtm = (time.perf_counter() - t0m) / 5
            others[mode] = {"images_per_s": round(B / dtm, 3), "ms_per_step": round(dtm * 1e3, 3), "dtype": MODES[mode][1]}
            if ref_np is not None:
                others[mode]["gpu_vs_oracle_max_abs_err"] = float(np.abs(y2[:1].cpu().numpy().astype(np.float64) - ref_np).max())
            m2.close()

    if rank == 0:
        images = args.steps * B * world
        value = images / elapsed
        gflop_ref = GFLOP_PER_IMAGE.get(S)
        out = {
            "metric": "images/sec @1024x1024 Swin-L" if S == 1024 else f"images/sec @{S}x{S} Swin-L",
            "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": MODES[args.compute][1], "data": "synthetic",
            "config": {"workload": f"BiRefNetConfig::swin_l() forward_logits, batch {B}/GPU, {S}x{S}, fp32 (BASELINE configs[1] shape)",
                       "batch_per_gpu": B, "global_batch": B * world, "size": S, "deform_mode": args.deform_mode,
                       "compute": args.compute,
                       "parallelism": f"{world} replica(s), batch-sharded, no data-path collective",
                       "inputs": "resident in HBM (torch cuda tensors), weights: synthetic seed 42"},
            "outputs_finite": finite,
            "roofline": roof, "cpu_baseline": cpu, "other_modes": others,
        }
        if gflop_ref:
            g = gflop_ref - (GFLOP_OFFSET_MOD_1024 * (S / 1024) ** 2 if args.deform_mode == "reference_cpu" else 0.0)
            out["path"] = {"reference_gflop_per_image": round(g, 1),
                           "tflops_at_reference_count": round(value / world * g / 1e3, 2),
                           "x_of_f32_mfma_peak": round(value / world * g / 1e3 / PEAK_F32_MFMA_TFLOPS, 4)}
        if stage_ms:
            out["stage_ms_profiled"] = {k: round(v, 3) for k, v in stage_ms.items()}
        print(json.dumps(out), flush=True)

    model.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
