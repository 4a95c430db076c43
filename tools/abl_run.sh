#!/bin/bash
# usage: tools/abl_run.sh "<env assignments>" tag cases...   (on the GPU box)
cd /tmp && export TMPDIR=/tmp
envs="$1"; tag="$2"; shift 2
export $envs
rocprofv3 --kernel-trace -d /root/repo/gpurun_out/abl_$tag -o t -- python /root/repo/tools/conv_time.py "$@" > /root/repo/gpurun_out/abl_$tag.log 2>&1
