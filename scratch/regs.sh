# register / occupancy summary of one csrc file's kernels: bash scratch/regs.sh san_conv_bf16.hip [grep-pattern]
f=/root/repo/spatialalignmentnetwork_amd/csrc/$1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -x hip -c $f -o /tmp/regs.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|AGPRs:|Occupancy|SGPRs:|ScratchSize" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//' | paste - - - - - - | grep -E "error|${2:-.}"
