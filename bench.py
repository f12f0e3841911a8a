#!/usr/bin/env python3
"""bench.py — images/sec of BiRefNet::forward_logits (Swin-L) on N MI355X, one process per GPU.

  python bench.py [--gpus N --steps K --warmup W] [--config c2|c3|c4|c5]

A "step" = one pass of the hot path over one batch of synthetic images already resident in HBM.  `--config` names a
BASELINE.json configuration (default c2 = configs[1], the one `metric` is quoted on):

  c2  Swin-L 1024x1024, batch 1,  fp32 (f32_half2: fp32-equivalent arithmetic)    single-image latency, parity-graded
  c3  Swin-L 1024x1024, batch 8,  bf16 (bf16 storage + bf16 MFMA, fp32 accumulate) MFMA-saturating throughput
  c4  as c3 per GPU (8 images per GPU, weak scaling); --strong: global batch 64 split over the ranks
  c5  Swin-L 2048x2048, batch 4,  bf16

The path shards by image (independent units, no data-path collective): every rank owns a full weight replica and its own
images.  N>1: either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or — when
WORLD_SIZE is unset — this process starts the N ranks itself BEFORE anything touches HIP and relays rank 0's JSON line.
torch.distributed (RCCL) is used for the timing barrier and the max-over-ranks reduction only.

The default line (config c2, no overrides) also carries `other_configs`: at N=1 the bf16 configurations c3 and c5 in both
deform modes (`c3`, `c3_deformable`, `c5`, `c5_deformable`: own model, warm-up, timed steps bracketed by synchronize, roofline
block, error of image 0 against the committed strided golden); at N>1 a `c4` / `c4_deformable` block (8 images per GPU bf16 on every
rank, per-rank times + max).  SURVEY.md D1 assigns the deformable mode to configs 3-5 (the only mode with a gather).

Rank 0 prints ONE JSON line; see DESIGN.md §measurement for how each field is obtained.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# SURVEY.md §8(d) / BASELINE.md §2: sum of 2*M*N*K over every Linear / conv / QK^T / PV the reference executes.
GFLOP_PER_IMAGE = {1024: 2534.9, 2048: 9772.6}
GFLOP_OFFSET_MOD_1024 = 84.0          # offset + modulator convs: computed and discarded on the reference CPU path
PEAK_F32_MFMA_TFLOPS = 157.3           # MI355X_MICROARCH.md, Peak FP32 (matrix), dense
PEAK_BF16_MFMA_TFLOPS = 2500.0         # MI355X_MICROARCH.md, Peak BF16 MFMA, dense
# compute mode -> (bf16 MFMAs per product or 0 for the fp32 instruction, dtype string, bytes per activation element in HBM)
MODES = {
    "f32": (0, "f32 (v_mfma_f32_32x32x2_f32, exact fp32 operands)", 4),
    "f32_split3": (6, "f32 storage+accumulate; GEMM operands split error-free into 3 bf16 terms, 6 bf16 MFMAs/product (fp32-equivalent)", 4),
    "f32_split2": (3, "f32 storage+accumulate; GEMM operands split into 2 bf16 terms (16-bit mantissa), 3 bf16 MFMAs/product", 4),
    "f32_half2": (3, "f32 storage+accumulate; GEMM operands as two fp16 planes of the power-of-two-scaled operand (22-bit mantissa), 3 fp16 MFMAs/product "
                     "(fp32-equivalent inside the fp16 range: |GEMM input| < 8190)", 4),
    "bf16": (1, "bf16 (activations and weights stored bf16 in HBM, bf16 MFMA, f32 accumulate / LayerNorm / softmax statistics)", 2),
    "f16": (1, "fp16 (activations and weights stored fp16 in HBM, fp16 MFMA at the bf16 rate, f32 accumulate / LayerNorm / softmax statistics): the bf16 mode's "
               "graph and kernels with 3 more mantissa bits", 2),
    "bf16_dec_split2": (1, "mixed: Swin backbone as bf16 (79 % of the FLOPs), fusion / squeeze / decoder as f32_split2 on f32 maps (3 bf16 MFMAs / product); "
                           "roofline priced against the bf16 peak", 2),
}
GEMM_FAMILIES = ("gemm_dense", "gemm_conv_nhwc", "gemm_gather_nchw", "gemm_deform_nhwc")
HBM_PEAK_TBS = 8.0                     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 measured with a float4 copy)
PROFILE_ROUND = ("r04", "r03", "r02")  # committed rocprofv3 evidence, newest first (profiles/README.md)
OTHER_STEPS, OTHER_WARMUP = 20, 5      # timed steps / warm-up of each `other_configs` block (0.6 - 1.2 s regions: barrier skew at N > 1 stays < 1 %)
# BASELINE.json configs[1..4]: (images per GPU, side, compute mode, label)
CONFIGS = {
    "c2": (1, 1024, "f32_half2", "BASELINE configs[1]: Swin-L 1024x1024 batch=1 fp32 on 1xMI355X (single-image latency)"),
    "c3": (8, 1024, "bf16", "BASELINE configs[2]: Swin-L 1024x1024 batch=8 bf16 on 1xMI355X (MFMA-saturating throughput)"),
    "c4": (8, 1024, "bf16", "BASELINE configs[3]: Swin-L 1024x1024 batch=64 bf16 sharded across 8xMI355X (8 images per GPU)"),
    "c5": (4, 2048, "bf16", "BASELINE configs[4]: Swin-L 2048x2048 batch=4 bf16 on 1xMI355X (high-res)"),
}


# host threads for the CPU baseline: the GPU box gives one GPU's job a share of 16 CPUs whatever os.cpu_count() says; more
# OpenMP / torch threads than cores makes their spin-waiting barriers crawl (BRN_CPU_THREADS overrides)
HOST_THREADS = int(os.environ.get("BRN_CPU_THREADS", "0")) or max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))


def log(msg):
    """progress on stderr: stdout carries exactly one JSON line"""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_torch_worker(size, deform_mode):
    """child of the cpu_baseline leg: the torch-CPU restatement (tests/torch_ref.py) on one image, timed; prints one JSON line"""
    import tempfile
    import numpy as np
    import torch
    torch.set_num_threads(HOST_THREADS)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch_ref
    from candle_birefnet_amd.config import BiRefNetConfig
    from candle_birefnet_amd.weights import birefnet_weight_spec, synth_input, synth_weights
    cfg = BiRefNetConfig(deform_mode=deform_mode)
    weights = synth_weights(birefnet_weight_spec(cfg), seed=42)
    xs = synth_input(1, size, size, seed0=1000)
    with torch.no_grad():
        torch_ref.forward_logits(synth_input(1, 128, 128, seed0=1000), weights, cfg, torch.float32)     # warm the thread pool
        t0 = time.perf_counter()
        y = torch_ref.forward_logits(xs, weights, cfg, torch.float32)
        dt = time.perf_counter() - t0
    fd, path = tempfile.mkstemp(suffix=".npy")
    os.close(fd)
    np.save(path, y.numpy())
    print(json.dumps({"seconds": dt, "threads": torch.get_num_threads(), "torch": torch.__version__, "out": path}), flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=15)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS), help="BASELINE.json configuration (sets batch, size, compute)")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (overrides --config)")
    ap.add_argument("--size", type=int, default=0, help="image side (overrides --config)")
    ap.add_argument("--deform-mode", default="reference_cpu", choices=["reference_cpu", "deformable"])
    ap.add_argument("--compute", default="", choices=[""] + list(MODES),
                    help="arithmetic of the contraction kernels (include/birefnet_hip.h brn_dtype; overrides --config)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: the config's 8-GPU global batch (images/GPU x 8) is split over the ranks")
    ap.add_argument("--also", default=None, help="comma list of other compute modes to time briefly on rank 0 at N=1 ('' = none; default: f32_split3,f32_split2,f32,f16 for c2)")
    ap.add_argument("--profile-steps", type=int, default=2, help="extra steps with per-launch HIP events for the roofline block")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "off", "on"])
    ap.add_argument("--other-configs", default="auto", choices=["auto", "off", "on"],
                    help="time the other BASELINE configurations in the same run (auto: with the default c2 line)")
    ap.add_argument("--mask-error", default="on", choices=["on", "off"],
                    help="one extra forward() of image 0 after the timed region for the mask-space error (off: rocprofv3 passes count only the timed forwards)")
    ap.add_argument("--cpu-baseline-size", type=int, default=0, help="image side for the CPU sample (0 = the workload's side, capped at 1024)")
    ap.add_argument("--cpu-torch-worker", type=int, default=0, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# ---- N>1 without torch.distributed.run: start the ranks from a process that has not touched HIP ---------------------------
def launch_ranks(n, cmd, extra_env=None, timeout=None):
    """Start `cmd` n times (RANK = LOCAL_RANK = 0..n-1, WORLD_SIZE = n, rendezvous on 127.0.0.1), relay rank 0's stdout,
    return the worst exit code.  Never exec()s: the children are ordinary child processes."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import tempfile
    import time
    procs = []
    with tempfile.TemporaryFile() as out0:
        for r in range(n):
            env = dict(os.environ)
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                        "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
            if extra_env:
                env.update(extra_env)
            procs.append(subprocess.Popen(cmd, env=env, stdout=out0 if r == 0 else subprocess.DEVNULL))
        # a rank that dies (no such device, out of memory ...) leaves the others waiting in the rendezvous or at a barrier: watch all
        # of them, and when one fails stop the rest (exact PIDs, never a pattern) instead of hanging until somebody's timeout
        t0, rc = time.time(), 0
        while True:
            codes = [p.poll() for p in procs]
            failed = [c for c in codes if c not in (None, 0)]
            if failed or all(c is not None for c in codes) or (timeout and time.time() - t0 > timeout):
                if not failed and any(c is None for c in codes):
                    failed = [124]                                   # timed out
                rc = failed[0] if failed else 0
                break
            time.sleep(0.2)
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        out0.seek(0)
        sys.stdout.write(out0.read().decode("utf-8", "replace"))
        sys.stdout.flush()
    return rc


def streams_used(batch):
    """sub-batches = HIP streams the library runs a device-resident batch of this size on (brn_api.cpp run_model: BRN_SPLIT_STREAMS,
    default 2, at least two images per part; profiled forwards always run on one stream)"""
    try:
        want = int(os.environ.get("BRN_SPLIT_STREAMS", "2"))
    except ValueError:
        want = 2
    return max(1, min(max(1, min(want, 8)), batch // 2))


def branch_streams_used(batch):
    """auxiliary streams for independent graph branches inside one forward (brn_graph.cpp: Branch; BRN_BRANCH_STREAMS): by default on
    (5 streams: ASPP branches, image-patch convs, lateral convs) when the batch runs as one part, off with sub-batch streams"""
    env = os.environ.get("BRN_BRANCH_STREAMS")
    if env is not None:
        try:
            mask = int(env)
        except ValueError:
            mask = -1
        if mask == 0:
            return 0
        if mask > 0:
            return bin(mask & 31).count("1")
    return 5 if streams_used(batch) == 1 else 0


def golden_error(y, S, deform_mode, mask=False):
    """max |y[0] - golden| on the committed strided golden of image 0 (seed 1000) for this geometry and deform mode
    (tests/golden/make_golden.py: fp64 torch restatement at 1024^2, fp32 at 2048^2), or None when there is none.
    mask=True: y is the output of forward() (birefnet.rs:466-469: sigmoid of the logits) and is compared with sigmoid(golden logits)
    — the space north_star's "masks within 1e-3 of reference" is stated in."""
    import numpy as np
    tag = "ref" if deform_mode == "reference_cpu" else "def"
    path = os.path.join(ROOT, "tests", "golden", f"model_{S}{'' if tag == 'ref' else '_def'}.npz")
    stride = {1024: 16, 2048: 32}.get(S)
    key = f"m{S}_full_{tag}_s{stride}"
    if stride is None or not os.path.exists(path):
        return None
    k = np.load(path)
    if key not in k.files:
        return None
    yn = y[0:1, :, ::stride, ::stride].float().cpu().numpy().astype(np.float64)
    g = k[key].astype(np.float64)
    if mask:
        g = 1.0 / (1.0 + np.exp(-g))
    return float(np.abs(yn - g).max())


def profile_model(model, x, n):
    """n profiled forwards (HIP events bracketing every launch on the launch stream): per-family sums + the stage timers"""
    fam = {}
    stage_ms = None
    model.set_profiling(True)
    for _ in range(n):
        model.forward_logits(x)
        for k, v in model.last_kernel_stats().items():
            a = fam.setdefault(k, {"launches": 0, "ms": 0.0, "gflop": 0.0, "gbytes": 0.0})
            a["launches"] += v["launches"]; a["ms"] += v["ms"]; a["gflop"] += v["flop"] / 1e9; a["gbytes"] += v["bytes"] / 1e9
        stage_ms = model.last_timings()
    model.set_profiling(False)
    return fam, stage_ms


def roofline_block(fam, n, compute, traffic_key, quote_traffic):
    """roofline of the dominant kernel family (the GEMM kernels) from `n` profiled steps; `traffic` = HBM-side bytes per launch from
    the committed rocprofv3 --pmc profile of the same configuration (profiles/<round>_gemm_traffic_<traffic_key>.json)"""
    fl = sum(fam[k]["gflop"] for k in GEMM_FAMILIES) * 1e9
    ms = sum(fam[k]["ms"] for k in GEMM_FAMILIES)
    by = sum(fam[k]["gbytes"] for k in GEMM_FAMILIES) * 1e9
    launches = sum(fam[k]["launches"] for k in GEMM_FAMILIES)
    achieved = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    npairs = MODES[compute][0]
    peak = PEAK_F32_MFMA_TFLOPS if npairs == 0 else PEAK_BF16_MFMA_TFLOPS / npairs
    traffic, traffic_note = None, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE not collected for this configuration"
    if quote_traffic:
        for rnd in PROFILE_ROUND:
            tj = os.path.join(ROOT, "profiles", f"{rnd}_gemm_traffic_{traffic_key}.json")
            if os.path.exists(tj):
                t = json.load(open(tj))
                traffic = round((t["hbm_read_gb_x2corrected"] + t["hbm_write_gb"]) * 1e9 / t["gemm_family_dispatches"])
                traffic_note = (f"bytes per launch, gemm family average, from {os.path.relpath(tj, ROOT)} (rocprofv3 --pmc FETCH_SIZE x2 "
                                "gfx950 correction + WRITE_SIZE, separate passes); algorithmic bytes per launch = "
                                f"{round(by / n / max(1, launches // n))}")
                break
    roof = {
        "bound": "mfma",
        "kernel": {"f32": "gemm family: gemm_f32_kernel", "bf16": "gemm family: gemm_bf16_kernel (+ gemm_f32_kernel for the NCHW-gather convs)"}.get(
            compute, "gemm family: gemm_split_ws_kernel / gemm_split_kernel (+ gemm_f32_kernel for the NCHW-gather convs)"),
        "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_note": traffic_note,
        "peak_note": ("fp32 MFMA dense peak" if npairs == 0 else
                      "bf16 MFMA dense peak" if npairs == 1 else
                      f"{'fp16' if compute == 'f32_half2' else 'bf16'} MFMA dense peak 2500 / {npairs} MFMAs per fp32 product; achieved counts ALGORITHMIC 2*M*N*K "
                      f"(= {achieved / PEAK_F32_MFMA_TFLOPS:.2f}x the fp32-MFMA peak of {PEAK_F32_MFMA_TFLOPS})"),
        "launches_per_step": launches // n, "gflop_per_step": round(fl / n / 1e9, 1), "ms_per_step": round(ms / n, 3),
        "avg_launch_ms": round(ms / max(1, launches), 4),
        "algorithmic_gbytes_per_step": round(by / n / 1e9, 2),
        "measured_over": f"{n} profiled step(s) after the timed region (HIP events bracketing every launch; the whole batch on ONE stream, "
                         "also where the timed region runs it as sub-batches on several streams: kernel durations of kernels running alone)",
        "families": {k: {"launches": v["launches"] // n, "ms": round(v["ms"] / n, 3), "gflop": round(v["gflop"] / n, 1),
                         "gbytes": round(v["gbytes"] / n, 3)}
                     for k, v in fam.items() if not k.startswith("region_")},
    }
    # HBM view of the memory-shaped pieces BASELINE.md C5 names: the deformable gathers (algorithmic bytes of their launches / their
    # HIP-event time) and every launch of the ASPPDeformable modules (region_aspp), against the 8 TB/s HBM peak
    hbm = {}
    for key, label in (("gemm_deform_nhwc", "deform_conv_gather"), ("region_aspp", "aspp_modules")):
        v = fam.get(key)
        if v and v["ms"] > 0 and v["launches"] > 0:
            gbs = v["gbytes"] / (v["ms"] * 1e-3)
            hbm[label] = {"launches": v["launches"] // n, "ms": round(v["ms"] / n, 3), "algorithmic_gbytes": round(v["gbytes"] / n, 3),
                          "gb_per_s": round(gbs, 1), "frac_of_hbm_peak": round(gbs / (HBM_PEAK_TBS * 1e3), 4),
                          "tflops": round(v["gflop"] / v["ms"], 1)}
    if hbm:
        roof["hbm_view"] = hbm
    return roof


def main(argv=None):
    args = parse_args(argv)
    if args.cpu_torch_worker:
        return cpu_torch_worker(args.cpu_torch_worker, args.deform_mode)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # no launcher above us: become one (nothing has imported torch or touched HIP in this process)
        raise SystemExit(launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + (sys.argv[1:] if argv is None else list(argv))))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to print a mislabelled line")

    cB, cS, cmode, clabel = CONFIGS[args.config]
    B, S = args.batch or cB, args.size or cS
    compute = args.compute or cmode
    scaling = "weak"
    if args.strong:
        gb = cB * 8
        if gb % world:
            raise SystemExit(f"--strong: global batch {gb} does not divide over {world} ranks")
        B, scaling = gb // world, "strong"
    custom = (B, S, compute) != (cB, cS, cmode) and not args.strong
    default_line = args.config == "c2" and not custom and not args.strong and args.deform_mode == "reference_cpu"
    also = args.also if args.also is not None else ("f32_split3,f32_split2,f32,f16" if (args.config == "c2" and not custom) else "")
    others_on = args.other_configs == "on" or (args.other_configs == "auto" and default_line)

    import numpy as np
    import torch
    import candle_birefnet_amd as cb
    from candle_birefnet_amd.shard import shard_range

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU fallback)")
    if compute not in cb.BiRefNet.COMPUTE:
        raise SystemExit(f"compute mode {compute} is not built into this library")
    # BRN_BENCH_DIST_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (ranks share devices; RCCL refuses
    # two ranks on one device): same launcher, barriers, max-over-ranks and JSON line; the numbers of such a run mean nothing
    backend = os.environ.get("BRN_BENCH_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def all_ranks(value):
        """[value of rank 0, ..., value of rank world-1] on every rank (control plane only: one float per rank)"""
        if dist is None:
            return [float(value)]
        t = torch.zeros(world, dtype=torch.float64, device=red_dev)
        t[rank] = float(value)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(v) for v in t.cpu().tolist()]

    weights_cache = {}

    def get_weights(cfg):
        if "w" not in weights_cache:       # the synthetic weights do not depend on deform_mode (same names, same seed)
            weights_cache["w"] = cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42)   # random-init weights of the real architecture
        return weights_cache["w"]

    def timed_workload(b, size, mode, deform_mode, steps, warmup, keep=False):
        """One workload on this rank: images [start, stop) of the global batch b * world (seed 1000 + global image index), model built,
        `warmup` untimed + `steps` timed forwards bracketed by barrier + synchronize.  Returns (seconds on this rank, per-rank seconds,
        model, x, y); the model is closed unless keep."""
        cfg = cb.BiRefNetConfig(deform_mode=deform_mode)                        # BiRefNetConfig::swin_l()
        model = cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(get_weights(cfg)), device=local_rank, max_batch=b, max_size=(size, size), compute=mode)
        start, stop = shard_range(b * world, world, rank)                         # candle_birefnet_amd/shard.py: the tested partition
        x = torch.from_numpy(cb.synth_input(stop - start, size, size, seed0=1000 + start)).cuda()
        y = None
        for _ in range(warmup):
            y = model.forward_logits(x)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            y = model.forward_logits(x)
        barrier()
        elapsed = time.perf_counter() - t0
        per_rank = all_ranks(elapsed)
        return max(per_rank), per_rank, cfg, model, x, y

    if rank == 0:
        log(f"building model ({compute}, B={B}, {S}x{S}, {args.deform_mode}); warmup {args.warmup} + {args.steps} timed steps")
    elapsed, per_rank, cfg, model, x, y = timed_workload(B, S, compute, args.deform_mode, args.steps, args.warmup)
    weights = get_weights(cfg)
    finite = bool(torch.isfinite(y).all().item())

    # ---- roofline of the dominant kernel family (the GEMM kernels), per-launch HIP events on the launch stream ----
    roof = None
    stage_ms = None
    if rank == 0 and args.profile_steps > 0:
        log(f"timed region done: {elapsed / args.steps * 1e3:.3f} ms/step; profiling {args.profile_steps} step(s)")
        fam, stage_ms = profile_model(model, x, args.profile_steps)
        roof = roofline_block(fam, args.profile_steps, compute, f"{args.config}{'_deformable' if args.deform_mode == 'deformable' else ''}_{compute}",
                              not custom)
    barrier()

    # ---- CPU baseline on this box's host cores, rank 0 at N=1 only: (a) the torch-CPU restatement of the same graph
    # (library-grade GEMM / conv kernels: the closest stand-in for candle's gemm + rayon CPU path that exists here),
    # (b) the unfused C++ oracle (which is also the checker of the GPU result).  Both are ports, neither is candle. ----
    cpu = None
    ref = None
    cs = args.cpu_baseline_size or min(S, 1024)
    if rank == 0 and world == 1 and args.cpu_baseline != "off":
        from oracle import oracle as orc
        xs = cb.synth_input(1, cs, cs, seed0=1000)
        g_img = GFLOP_PER_IMAGE.get(cs, 0.0) - (GFLOP_OFFSET_MOD_1024 * (cs / 1024) ** 2 if args.deform_mode == "reference_cpu" else 0.0)
        orc.set_num_threads(HOST_THREADS)
        log(f"cpu_baseline: C++ oracle, {cs}x{cs}, {HOST_THREADS} threads")
        t0c = time.perf_counter()
        ref = orc.forward_logits(orc.cfg_from(cfg), weights, xs)
        dt_orc = time.perf_counter() - t0c
        oracle_port = {"value": round(1.0 / dt_orc, 5), "unit": "images/s", "cores": orc.num_threads(), "seconds_per_image": round(dt_orc, 2),
                       "gflops": round(g_img / dt_orc, 1) if g_img > 0 else None,
                       "impl": "oracle/brn_oracle.cpp: unfused fp32 restatement, OpenMP"}
        # (a) in a child process (own thread pool, cannot stall this process, bounded by a timeout); CPU only: it never loads HIP
        torch_port = None
        try:
            log(f"cpu_baseline: torch-CPU restatement, {cs}x{cs}, {HOST_THREADS} threads (child process, <= 300 s)")
            pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-torch-worker", str(cs), "--deform-mode", args.deform_mode],
                                capture_output=True, text=True, timeout=300, env=dict(os.environ, HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(HOST_THREADS)))
            if pr.returncode != 0:
                raise RuntimeError(pr.stderr.strip().splitlines()[-1] if pr.stderr.strip() else f"worker exit {pr.returncode}")
            tw = json.loads(pr.stdout.strip().splitlines()[-1])
            dt_t = tw["seconds"]
            torch_port = {"value": round(1.0 / dt_t, 5), "unit": "images/s", "cores": tw["threads"], "seconds_per_image": round(dt_t, 2),
                          "gflops": round(g_img / dt_t, 1) if g_img > 0 else None,
                          "impl": f"tests/torch_ref.py on torch {tw['torch']} CPU, fp32",
                          "vs_oracle_max_abs_err": float(np.abs(np.load(tw["out"]).astype(np.float64) - ref).max())}
            os.unlink(tw["out"])
        except Exception as e:                                    # the baseline is a report, never a reason to lose the line
            torch_port = {"error": f"{type(e).__name__}: {e}"}
        best = torch_port if (torch_port and "value" in torch_port and torch_port["value"] > oracle_port["value"]) else oracle_port
        cpu = {"value": best["value"], "unit": "images/s", "cores": best["cores"], "kind": "port",
               "sample": f"1 image {cs}x{cs} fp32, full Swin-L forward_logits, {best['impl']}, {best['seconds_per_image']} s wall; host cpus={os.cpu_count()}",
               "seconds_per_image": best["seconds_per_image"], "gflops": best.get("gflops"),
               "torch_cpu_restatement": torch_port, "oracle_port": oracle_port,
               "note": "neither is candle (the Rust crate cannot be built here: BASELINE.md §4); a reported baseline, not the target"}
        if cs == S:
            err = np.abs(y[:1].float().cpu().numpy().astype(np.float64) - ref)
            cpu["gpu_vs_oracle_max_abs_err"] = float(err.max())
            cpu["gpu_vs_oracle_max_rel_err"] = float((err / np.maximum(np.abs(ref), 1e-3)).max())
            cpu["gpu_vs_oracle_gate_1e-3abs_or_1e-2rel"] = bool(((err <= 1e-3) | (err <= 1e-2 * np.abs(ref))).all())

    # ---- the other compute modes, briefly (rank 0, N=1): same weights, same input, 5 timed steps each ----
    others = None
    if rank == 0 and world == 1 and also:
        others = {}
        ref_np = ref if (ref is not None and cs == S) else None
        for mode in [m for m in also.split(",") if m and m != compute]:
            log(f"other mode: {mode}")
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
                e2 = np.abs(y2[:1].float().cpu().numpy().astype(np.float64) - ref_np)
                others[mode]["gpu_vs_oracle_max_abs_err"] = float(e2.max())
                others[mode]["gpu_vs_oracle_gate_1e-3abs_or_1e-2rel"] = bool(((e2 <= 1e-3) | (e2 <= 1e-2 * np.abs(ref_np))).all())
            m2.close()
    gold_headline = golden_error(y, S, args.deform_mode) if (rank == 0 and not custom) else None
    # mask space: forward() = sigmoid(logits) (birefnet.rs:466-469) of image 0 against sigmoid(golden logits)
    gold_mask_headline = golden_error(model.forward(x[:1]), S, args.deform_mode, mask=True) if (rank == 0 and not custom and args.mask_error == "on") else None
    model.close()
    del model, x, y

    # ---- the other BASELINE configurations, driver-timed in the same run: each its own model, warm-up and timed region ----
    other_cfgs = None
    if others_on:
        other_cfgs = {}
        # (the last c3 entry: the same workload in the mixed mode — bf16 backbone, f32_split2 fusion / squeeze / decoder — the arithmetic
        # that brings the mask-space error of the reference_cpu configuration under 1e-3; reported beside c3, it does not replace it)
        plan = ([("c3", "reference_cpu", None), ("c3", "deformable", None), ("c5", "reference_cpu", None), ("c5", "deformable", None),
                 ("c3", "reference_cpu", "bf16_dec_split2"),
                 # the same configurations with fp16 instead of bf16 as the 16-bit type (compute mode f16): same kernels, bytes and speed,
                 # 3 more mantissa bits; reported beside the bf16 lines BASELINE names, they do not replace them
                 ("c3", "reference_cpu", "f16"), ("c3", "deformable", "f16"), ("c5", "reference_cpu", "f16"), ("c5", "deformable", "f16")] if world == 1 else
                [("c4", "reference_cpu", None), ("c4", "deformable", None)])
        for cname, dm, cmode_over in plan:
            oB, oS, omode, olabel = CONFIGS[cname]
            key = cname + ("_deformable" if dm == "deformable" else "")
            if cmode_over:
                omode, key = cmode_over, f"{key}_{cmode_over}"
                olabel = olabel.replace(" bf16 ", f" {cmode_over} ")
            if rank == 0:
                log(f"other config {key}: {omode}, B={oB}/GPU, {oS}x{oS}, {dm}; warmup {OTHER_WARMUP} + {OTHER_STEPS} timed steps")
            t_o, pr_o, _, m_o, x_o, y_o = timed_workload(oB, oS, omode, dm, OTHER_STEPS, OTHER_WARMUP)
            blk = None
            if rank == 0:
                g_ref = GFLOP_PER_IMAGE[oS] - (GFLOP_OFFSET_MOD_1024 * (oS / 1024) ** 2 if dm == "reference_cpu" else 0.0)
                ips = OTHER_STEPS * oB * world / t_o
                blk = {"workload": olabel + (f"; that per-GPU workload on each of {world} ranks" if world > 1 else ""),
                       "images_per_s": round(ips, 3), "ms_per_step": round(t_o / OTHER_STEPS * 1e3, 3), "steps": OTHER_STEPS, "warmup": OTHER_WARMUP,
                       "n_gpus": world, "batch_per_gpu": oB, "size": oS, "dtype": MODES[omode][1], "compute": omode, "deform_mode": dm,
                       "streams_per_gpu": streams_used(oB), "branch_streams_per_gpu": branch_streams_used(oB),
                       "outputs_finite": bool(torch.isfinite(y_o).all().item()),
                       "reference_gflop_per_image": round(g_ref, 1),
                       "whole_step_frac_of_mode_peak": round(ips / world * g_ref / 1e3 / PEAK_BF16_MFMA_TFLOPS, 4),
                       "max_abs_err_image0_vs_strided_golden": golden_error(y_o, oS, dm),
                       "max_abs_err_mask_image0": golden_error(m_o.forward(x_o[:1]), oS, dm, mask=True)}
                if world > 1:
                    blk["per_rank_ms_per_step"] = [round(v / OTHER_STEPS * 1e3, 3) for v in pr_o]
                fam_o, _ = profile_model(m_o, x_o, 1)
                blk["roofline"] = roofline_block(fam_o, 1, omode, f"{key}_{omode}", True)
            barrier()
            m_o.close()
            del m_o, x_o, y_o
            if rank == 0:
                other_cfgs[key] = blk

    if rank == 0:
        images = args.steps * B * world
        value = images / elapsed
        gflop_ref = GFLOP_PER_IMAGE.get(S)
        workload = (clabel if not custom else f"custom: Swin-L {S}x{S} batch={B}/GPU {compute}") + \
                   (f"; strong scaling: global batch {B * world} split over {world} rank(s)" if args.strong else
                    (f"; weak scaling: that per-GPU workload on each of {world} ranks, one rank per GPU" if world > 1 else ""))
        out = {
            "metric": "images/sec @1024x1024 Swin-L" if S == 1024 else f"images/sec @{S}x{S} Swin-L",
            "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": MODES[compute][1], "data": "synthetic",
            "config": {"workload": workload, "baseline_config": None if custom else args.config,
                       "batch_per_gpu": B, "global_batch": B * world, "size": S, "deform_mode": args.deform_mode,
                       "compute": compute, "streams_per_gpu": streams_used(B), "branch_streams_per_gpu": branch_streams_used(B),
                       "parallelism": f"{world} replica(s), batch-sharded (candle_birefnet_amd.shard.shard_range), no data-path collective",
                       "inputs": "resident in HBM (torch cuda tensors), weights: synthetic seed 42"},
            "outputs_finite": finite,
            "per_rank_ms_per_step": [round(v / args.steps * 1e3, 3) for v in per_rank],
            "max_abs_err_image0_vs_strided_golden": gold_headline, "max_abs_err_mask_image0": gold_mask_headline,
            "roofline": roof, "cpu_baseline": cpu, "other_modes": others, "other_configs": other_cfgs,
        }
        if gflop_ref:
            g = gflop_ref - (GFLOP_OFFSET_MOD_1024 * (S / 1024) ** 2 if args.deform_mode == "reference_cpu" else 0.0)
            peak_mode = PEAK_F32_MFMA_TFLOPS if MODES[compute][0] == 0 else PEAK_BF16_MFMA_TFLOPS / MODES[compute][0]
            out["path"] = {"reference_gflop_per_image": round(g, 1),
                           "tflops_at_reference_count": round(value / world * g / 1e3, 2),
                           "x_of_f32_mfma_peak": round(value / world * g / 1e3 / PEAK_F32_MFMA_TFLOPS, 4),
                           "whole_step_frac_of_mode_peak": round(value / world * g / 1e3 / peak_mode, 4)}
        if stage_ms:
            out["stage_ms_profiled"] = {k: round(v, 3) for k, v in stage_ms.items()}
        # LAST key of the line (a stored tail of it still carries every headline): images/s, GEMM-family roofline fraction and the
        # errors of image 0 (logits / mask = after the sigmoid) per configuration timed in this run
        summ = {(args.config if not custom else "custom") + ("_deformable" if args.deform_mode == "deformable" else ""):
                {"images_per_s": round(value, 2), "frac": roof["frac"] if roof else None, "err_logits": gold_headline, "err_mask": gold_mask_headline}}
        for k_, b_ in (other_cfgs or {}).items():
            summ[k_] = {"images_per_s": round(b_["images_per_s"], 2), "frac": b_["roofline"]["frac"],
                        "err_logits": b_["max_abs_err_image0_vs_strided_golden"], "err_mask": b_["max_abs_err_mask_image0"]}
        for k_, b_ in (others or {}).items():
            summ[f"{args.config}_mode_{k_}"] = {"images_per_s": round(b_["images_per_s"], 2)}
        out["summary"] = summ
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
