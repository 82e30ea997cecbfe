#!/bin/bash
# compile one .hip file for gfx950 and print a compact resource summary (dev helper)
f=$1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $f -o /tmp/$(basename $f .hip).o -Rpass-analysis=kernel-resource-usage 2>&1 | \
 grep -E "error|warning:|Function Name|    VGPRs:|VGPRs Spill|Occupancy|LDS Size" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | \
 awk '/Function Name/{if(l)print l; l=$3} /VGPRs:/{l=l" vgpr="$2} /Spill/{l=l" spill="$3} /Occupancy/{l=l" occ="$4} /LDS/{l=l" lds="$4} /error|warning/{print} END{print l}'
