"""Pretty-print the per-launcher table of a bench.py JSON line: python tools/show_bench.py gpurun_out/b.json"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{d['value']:.1f} {d['unit']}  {d['ms_per_step']:.3f} ms/step  roofline={d.get('roofline', {}).get('frac')}")
rows = sorted(d.get("kernels", {}).items(), key=lambda kv: -kv[1]["ms_per_step"])
tot = 0.0
for k, v in rows:
    tot += v["ms_per_step"]
    extra = " ".join(f"{a}={b}" for a, b in v.items() if a not in ("calls_per_step", "ms_per_step"))
    print(f"  {v['ms_per_step'] * 1000:8.1f} us  {int(v['calls_per_step']):3d}x  {k}  {extra}")
print(f"  {tot * 1000:8.1f} us  sum")
