#!/bin/bash
# Round-end evidence run on the GPU box: kernel stats (rocprofv3 --kernel-trace --stats) and the HBM / SQ counter passes
# (each --pmc set in its own pass, never combined with tracing) of the bench commands quoted in DESIGN.md.
# usage: tools/collect_profiles.sh <tag>      (output: gpurun_out/prof_<tag>/...)
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/prof_$1; mkdir -p $O
B="python3 $R/bench.py --no-secondary --no-cpu-baseline --lanes 1"   # (one lane: per-kernel durations and counters of kernels that do not overlap)
run() { name=$1; shift; echo "== $name"; "$@" > $O/$name.log 2>&1; }
run stats_f32   rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f32 -o t -- $B --steps 10 --warmup 2
run stats_c2f16 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2f16 -o t -- $B --config 2 --dtype f16 --steps 10 --warmup 2
run stats_c3f16 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3f16 -o t -- $B --config 3 --dtype f16 --steps 3 --warmup 1
run stats_c3f32 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3f32 -o t -- $B --config 3 --dtype f32 --steps 2 --warmup 1
run fetch_f32   rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_f32 -o t -- $B --steps 1 --warmup 0
run write_f32   rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_f32 -o t -- $B --steps 1 --warmup 0
run fetch_f16   rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_f16 -o t -- $B --config 3 --dtype f16 --steps 1 --warmup 0
run write_f16   rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_f16 -o t -- $B --config 3 --dtype f16 --steps 1 --warmup 0
run fetch_c3f32 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_c3f32 -o t -- $B --config 3 --dtype f32 --steps 1 --warmup 0
run write_c3f32 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_c3f32 -o t -- $B --config 3 --dtype f32 --steps 1 --warmup 0
SQ="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
run sq_f32      rocprofv3 --pmc $SQ --output-format csv -d $O/sq_f32 -o t -- $B --steps 1 --warmup 0
run sq_f16      rocprofv3 --pmc $SQ --output-format csv -d $O/sq_f16 -o t -- $B --config 3 --dtype f16 --steps 1 --warmup 0
run sq_c3f32    rocprofv3 --pmc $SQ --output-format csv -d $O/sq_c3f32 -o t -- $B --config 3 --dtype f32 --steps 1 --warmup 0
# the sources the library that just ran was built from (bench.py compares this with the running library: traffic_stale)
cp $R/lib/libmi355_nnunet.so.digest $O/lib_digest.txt
# keep the merge-back small: the per-dispatch CSVs of the counter passes are a few MiB each
find $O -name "*agent_info.csv" -delete
ls -la $O/*/ | head -40
