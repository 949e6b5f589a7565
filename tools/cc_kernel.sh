#!/bin/bash
# compile one kernel file for gfx950 and print its resource usage (bring-up helper)
f=$1
cd /root/repo/av1-base_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $f.hip -o /tmp/$f.o -Rpass-analysis=kernel-resource-usage -save-temps=obj 2>&1 | grep -E "error|warning:|Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: //' | sed 's/\[-Rpass.*//'
echo "scratch ops: $(grep -c scratch_ /tmp/$f-hip-amdgcn-amd-amdhsa-gfx950.s)  asm lines: $(wc -l < /tmp/$f-hip-amdgcn-amd-amdhsa-gfx950.s)"
