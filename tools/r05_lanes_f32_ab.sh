#!/bin/bash
# lanes A/B for config 3 fp32 on one box (the fp32 kernels are not power-limited: overlap should come back as time)
mkdir -p gpurun_out/r05
for L in 1 2 3 4 1 2; do
  python bench.py --config 3 --dtype f32 --lanes $L --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05/tmp_f32_l$L.json 2> gpurun_out/r05/tmp.err || exit 1
  python - $L <<'PY' | tee -a gpurun_out/r05/lanes_f32_ab.txt
import json, sys
d = json.loads(open(f"gpurun_out/r05/tmp_f32_l{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("config3 f32 lanes", sys.argv[1], d["ms_per_step"], "sum of kernels (single-lane profile step)", round(sum(v["ms_total"] for v in d["kernels"].values()), 1))
PY
done
