"""bench.py's own N-rank launcher (used when WORLD_SIZE is unset and --gpus N > 1), with stub workers: environment plumbing,
rank-0 relay, exit-code propagation, and the refusal to print a mislabelled line.  No GPU, no torch in the parent."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STUB = ("import os, json, sys; r = int(os.environ['RANK']); "
        "print(json.dumps({k: os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY')})); "
        "sys.exit(int(os.environ.get('STUB_FAIL_RANK', '-1')) == r)")


def test_launch_ranks_env_and_relay(capfd):
    import bench
    rc = bench.launch_ranks(3, [sys.executable, "-c", STUB])
    out = capfd.readouterr().out.strip().splitlines()
    assert rc == 0 and len(out) == 1                       # only rank 0's stdout is relayed: ONE JSON line reaches the driver
    env = json.loads(out[0])
    assert env["RANK"] == "0" and env["LOCAL_RANK"] == "0" and env["WORLD_SIZE"] == "3"
    assert env["MASTER_ADDR"] == "127.0.0.1" and int(env["MASTER_PORT"]) > 0 and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_launch_ranks_propagates_failure(capfd):
    import bench
    assert bench.launch_ranks(2, [sys.executable, "-c", STUB], extra_env={"STUB_FAIL_RANK": "1"}) != 0
    assert bench.launch_ranks(2, [sys.executable, "-c", STUB], extra_env={"STUB_FAIL_RANK": "0"}) != 0
    capfd.readouterr()


def test_launch_ranks_stops_the_others_when_one_rank_dies(capfd):
    """rank 1 exits with an error at once, rank 0 would wait for it forever (the rendezvous of a real run): the launcher must stop rank 0
    and return the failure within seconds instead of hanging"""
    import time
    import bench
    code = "import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(7)\ntime.sleep(600)\n"
    t0 = time.time()
    rc = bench.launch_ranks(2, [sys.executable, "-c", code])
    assert rc == 7 and time.time() - t0 < 60
    capfd.readouterr()


def test_bench_refuses_world_size_mismatch():
    """`--gpus 8` under WORLD_SIZE=2 (or --gpus 1 under WORLD_SIZE=2) must fail before any GPU work, not print n_gpus: 1"""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    for gpus in ("8", "1"):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", gpus], env=env, capture_output=True, text=True, timeout=120)
        assert p.returncode != 0 and "WORLD_SIZE=2" in (p.stderr + p.stdout) and "{" not in p.stdout


def test_bench_config_table_names_every_gpu_baseline_config():
    import bench
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    assert len(base) == 5 and sorted(bench.CONFIGS) == ["c2", "c3", "c4", "c5"]
    for key, idx in (("c2", 1), ("c3", 2), ("c4", 3), ("c5", 4)):
        B, S, mode, label = bench.CONFIGS[key]
        assert f"{S}×{S}" in base[idx] and (("bf16" in base[idx]) == (mode == "bf16")) and f"configs[{idx}]" in label
    a = bench.parse_args(["--config", "c5", "--gpus", "1"])
    assert a.config == "c5" and a.batch == 0 and a.size == 0 and a.compute == ""
