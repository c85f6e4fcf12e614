#!/bin/bash
# usage: tools/kernel_resources.sh panvc3_amd/csrc/FILE.hip — VGPRs, spills, LDS and occupancy of every kernel in the file
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -x hip -c "$1" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
	| grep -E "Function Name|VGPRs:|Spill|LDS Size|Occupancy" | sed -E 's/^.*remark: +//; s/ \[-Rpass.*$//' \
	| awk '/Function Name/ {printf "\n%s", $3} !/Function Name/ {printf " | %s", $0} END {print ""}' | c++filt | sed -E 's/\(anonymous namespace\):://; s/\(pvc3::dev_entry const\*.*\)//' | cut -c1-220
