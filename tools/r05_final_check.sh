#!/bin/bash
# round-5 closing run on the GPU box: the whole GPU suite, then the default bench line (gpurun_out/r05/)
mkdir -p gpurun_out/r05
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r05/t_full2.log 2>&1; tail -3 gpurun_out/r05/t_full2.log
timeout -k 10 400 python bench.py > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err; echo bench rc $?; tail -c 300 gpurun_out/r05/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
s = d["secondary"]
print("c3f16", s["config3_f16"]["ms_per_step"], s["config3_f16"]["roofline"]["conv_stages_frac"], "c3f32", s["config3_f32"]["ms_per_step"])
r = s["reference_setting"]; print("ref f16", r["f16"]["seconds_per_volume"], "f32", r["f32"]["seconds_per_volume"])
print(json.dumps(r["f16"].get("parity_vs_cpu_ref"))[:600]); print(json.dumps(r["f16"].get("consistency_vs_f32_labels")))
print(json.dumps(s["config3_f16"].get("parity_vs_cpu_ref"))[:700])
PY
