#!/bin/bash
# usage: scratch/spills.sh [extra hipcc flags]  -> per-kernel VGPR / spill table of the fused kernels
mkdir -p /tmp/isa && cd /tmp/isa && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -Wno-unused-value -I/root/repo/include --cuda-device-only -S -o qbnn.s "$@" /root/repo/quantised_bayesian_nets_amd/csrc/${QBNN_UNIT:-qbnn_blocks.hip} 2>&1 | grep -v "hip-link" | head
python3 - <<'PY'
import re
txt=open('/tmp/isa/qbnn.s').read()
for m in re.finditer(r'- \.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', txt, re.S):
    name=m.group(2)
    if 'chain' in name or 'down' in name or 'pp' in name or 'w16' in name or 'rs_conv' in name or 'ks_conv' in name:
        short=re.sub(r'ConvCfgI|Li|E','',name)[3:100]
        print(short, 'agpr',m.group(1),'scratch',m.group(3),'vgpr',m.group(5),'spill',m.group(6))
PY
