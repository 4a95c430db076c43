#!/usr/bin/env python3
"""Fills the R4_* placeholders of DESIGN.md section 0 from a `python bench.py` JSON line (one-off helper of round 4)."""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
s = d["secondary"]
r = d["roofline"]
rep = {
    "R4_C2_MS": f"{d['ms_per_step']:.1f}", "R4_C2_VPS": f"{d['value']:.1f}", "R4_W3_TF": f"{r['achieved']:.0f}", "R4_W3_FRAC": f"{r['frac']:.2f}",
    "R4_W3_EXEC": f"{r['frac_executed']:.2f}", "R4_W3_SHARE": f"{100 * r['time_share']:.0f} %",
    "R4_C3F16_MS": f"{s['config3_f16']['ms_per_step']:.0f}", "R4_C3F16_FRAC": f"{s['config3_f16']['roofline']['conv_stages_frac']:.3f}",
    "R4_C3F32_MS": f"{s['config3_f32']['ms_per_step']:.0f}",
    "R4_REF16_S": f"{s['reference_setting']['f16']['seconds_per_volume']:.2f}", "R4_REF32_S": f"{s['reference_setting']['f32']['seconds_per_volume']:.2f}",
    "R4_REF16_X": f"{s['reference_setting']['f16']['speedup_vs_nominal_5min']:.0f}", "R4_REF32_X": f"{s['reference_setting']['f32']['speedup_vs_nominal_5min']:.0f}",
}
t = open("DESIGN.md").read()
for k, v in rep.items():
    t = t.replace(k, v)
open("DESIGN.md", "w").write(t)
print(rep)
