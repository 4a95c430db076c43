#!/bin/bash
# usage: tools/h16_pmc.sh <probe binary> <tag>   (GPU box): FETCH_SIZE / WRITE_SIZE of the probe's launches, one pass each
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /root/repo/gpurun_out/h16pmc_$2_$c -o t -- /root/repo/tools/$1 > /root/repo/gpurun_out/h16pmc_$2_$c.log 2>&1
done
