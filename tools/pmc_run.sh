#!/bin/bash
# usage: tools/pmc_run.sh "<env assignments>" tag "<counters>" cases...   (on the GPU box; counters in their own pass, no tracing)
cd /tmp && export TMPDIR=/tmp
envs="$1"; tag="$2"; ctrs="$3"; shift 3
export $envs
rocprofv3 --pmc $ctrs --output-format csv -d /root/repo/gpurun_out/pmc_$tag -o t -- python /root/repo/tools/conv_time.py "$@" > /root/repo/gpurun_out/pmc_$tag.log 2>&1
