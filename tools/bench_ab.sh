#!/bin/bash
# usage: tools/bench_ab.sh "<env assignments>" tag <bench args...>   (on the GPU box): bench under rocprofv3 --kernel-trace
cd /tmp && export TMPDIR=/tmp
envs="$1"; tag="$2"; shift 2
[ -n "$envs" ] && export $envs
rocprofv3 --kernel-trace -d /root/repo/gpurun_out/ab_$tag -o t -- python /root/repo/bench.py "$@" --no-cpu-baseline --no-secondary > /root/repo/gpurun_out/ab_$tag.log 2>&1
grep -o '"ms_per_step": [0-9.]*' /root/repo/gpurun_out/ab_$tag.log | head -1
