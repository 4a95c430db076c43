#!/bin/bash
# GPU box: FETCH_SIZE of tools/dma_probe's launches (known bytes per launch) - calibrates the counter for the conv kernels'
# brick-staging access patterns (MI355X_MICROARCH.md: "other access widths are uncalibrated").  Output: gpurun_out/dma_calib/
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/dma_calib; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o t -- /root/repo/tools/dma_probe > $O/fetch.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/rdreq -o t -- /root/repo/tools/dma_probe > $O/rdreq.log 2>&1
ls -la $O/*/
