#!/bin/bash
# A/B on one box: conv3_f16_dma_kernel (one workgroup per CU, nine-tap ring) against conv3_f16_dma2_kernel (two per CU, three-tap ring)
mkdir -p gpurun_out/r05
for v in 0 1 0 1; do
  MI355_F16_DMA2=$v python bench.py --config 3 --lanes 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05/tmp_dma2_$v.json 2> gpurun_out/r05/tmp.err || exit 1
  python - $v <<'PY' | tee -a gpurun_out/r05/dma2_ab.txt
import json, sys
d = json.loads(open(f"gpurun_out/r05/tmp_dma2_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("dma2", sys.argv[1], d["ms_per_step"], d["roofline"]["conv_stages_frac"], {k: (v["ms_total"], v["tflops"]) for k, v in d["kernels"].items() if "dma" in k and "s2" not in k})
PY
done
