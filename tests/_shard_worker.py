"""Worker of tests/test_configs_gpu.py::test_two_processes_two_handles_bit_equal and of the launcher tests: one rank of an
N-rank batch-sharded run of the PRODUCT (not the oracle) — reads RANK / LOCAL_RANK / WORLD_SIZE as bench.py does, builds its
own model handle on device LOCAL_RANK % device_count, runs its shard of the global batch, writes <outdir>/rank<r>.npy."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    outdir, tag, gb, compute = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    rank, local, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    import candle_birefnet_amd as cb
    from candle_birefnet_amd.shard import shard_range
    import golden_cases as G
    ndev = cb.device_count()
    assert ndev > 0, "no HIP device in the worker"
    dev = local % ndev
    depths, S, _, mode = G.MODEL_CASES[tag]
    cfg = cb.BiRefNetConfig(deform_mode=mode)
    cfg.swin.depths = list(depths)
    w = cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42)
    a, b = shard_range(gb, world, rank)
    m = cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(w), device=dev, compute=compute)
    x = cb.synth_input(b - a, S, S, seed0=1000 + a)
    y = m.forward_logits(x)
    np.save(os.path.join(outdir, f"rank{rank}.npy"), y)
    m.close()
    print(f'{{"rank": {rank}, "device": {dev}, "images": [{a}, {b}]}}', flush=True)


if __name__ == "__main__":
    main()
