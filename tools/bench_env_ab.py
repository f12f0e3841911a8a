#!/usr/bin/env python3
"""A/B of environment switches on one box: runs bench.py once per setting (fresh process each) and prints img/s, ms, gemm TF/s.
  python tools/bench_env_ab.py "BRN_PLANES_CFG=0" "BRN_PLANES_CFG=1" "BRN_NO_P3=1" -- --config c2 --steps 10"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
for rep in range(2):
    for setting in args:
        env = dict(os.environ)
        for kv in setting.split(","):
            if "=" in kv:
                k, v = kv.split("=", 1)
                env[k] = v
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-baseline", "off", "--also", "", "--steps", "10", "--warmup", "3"] + extra,
                           env=env, capture_output=True, text=True)
        try:
            d = json.loads(p.stdout.strip().splitlines()[-1])
            r = d["roofline"]
            print(f"{setting:40s} {d['value']:8.2f} img/s {d['ms_per_step']:8.3f} ms  gemm {r['achieved']:7.1f} TF/s frac {r['frac']:.3f}  gemm ms {r['ms_per_step']:.3f}", flush=True)
        except Exception as e:
            print(setting, "FAILED", e, p.stderr[-300:], flush=True)
