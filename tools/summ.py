#!/usr/bin/env python3
"""One-screen summary of bench.py JSON lines: ms per step, dominant kernel, conv-stage fraction, the top kernels."""
import json
import sys
for f in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable:", e)
        continue
    r = d["roofline"]
    print(f"{f}: {d['ms_per_step']} ms/step, {d['value']} {d['unit']}; {r['kernel']} {r['achieved']} TF (executed {r['frac_executed']}), conv stages {r['conv_stages_frac']}")
    for k, v in list(d["kernels"].items())[:14]:
        print("   ", k, v)
    for k, v in (d.get("secondary") or {}).items():
        if isinstance(v, dict) and "ms_per_step" in v:
            print(f"  secondary {k}: {v['ms_per_step']} ms, conv stages {v['roofline']['conv_stages_frac']}, parity {json.dumps(v.get('parity_vs_cpu_ref'))[:300]}")
