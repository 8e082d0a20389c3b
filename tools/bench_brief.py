#!/usr/bin/env python3
"""Run bench.py (no CPU baseline) and print the few numbers that matter for A/B work: graphs/s, ms/step, per-class launch time."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "5", "--cpu-baseline-seconds", "0"] + sys.argv[1:],
                     capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print(out.stdout[-2000:], out.stderr[-2000:])
    sys.exit(1)
d = json.loads(line[-1])
r = d.get("roofline", {})
print(f"value {d['value']} graphs/s  {d['ms_per_step']} ms/step | wino avg {r.get('avg_launch_ms')} ms frac {r.get('frac')} | 1-stream: {r.get('measured_on', '')[60:130]}")
for k, v in d.get("other_kernels", {}).items():
    if isinstance(v, dict):
        print(f"   {k:18s} {v.get('achieved')} {v.get('unit')}  frac {v.get('frac')}  avg {v.get('avg_launch_ms')} ms x {v.get('launches')}")
