"""The N>1 path on CPU: world_size 2, gloo.  The path shards by image with NO data-path collective, so what must hold
is (a) the shards partition the global batch, (b) each rank generates exactly its own images, (c) per-image results do not
depend on which rank / batch they were computed in (checked with the CPU oracle), (d) the control collectives bench.py
uses (barrier, max-over-ranks) behave."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, global_batch, S, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import candle_birefnet_amd as cb
        from candle_birefnet_amd.shard import max_over_ranks, shard_range
        from oracle import oracle as O
        import golden_cases as G
        O.set_num_threads(2)
        a, b = shard_range(global_batch, world, rank)
        x = cb.synth_input(b - a, S, S, seed0=1000 + a)          # rank-local images, global seeds
        cfg, w, _ = G.model_case("m64_d2222_ref")
        y = O.forward_logits(O.cfg_from(cfg), w, x) if b > a else np.zeros((0, 1, S, S), np.float32)
        dist.barrier()
        t = max_over_ranks(1.0 + rank, dist)                     # the timing reduction of bench.py
        sums = torch.tensor([float(b - a)], dtype=torch.float64)
        dist.all_reduce(sums)                                    # control-plane only: count of images processed
        q.put((rank, a, b, y, t, float(sums.item())))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    import candle_birefnet_amd as cb
    from oracle import oracle as O
    import golden_cases as G
    world, gb, S = 2, 3, 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, gb, S, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 2), (2, 3)]       # contiguous, uneven shard goes to the first rank
    assert all(r[4] == 2.0 for r in res)                          # max over ranks of (1 + rank)
    assert all(r[5] == gb for r in res)
    cfg, w, _ = G.model_case("m64_d2222_ref")
    x_all = cb.synth_input(gb, S, S)
    y_all = O.forward_logits(O.cfg_from(cfg), w, x_all)
    y_sh = np.concatenate([r[3] for r in res], 0)
    np.testing.assert_array_equal(y_sh, y_all)                    # images are independent units: identical bits
