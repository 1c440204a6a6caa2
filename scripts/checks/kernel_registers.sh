#!/bin/bash
# Register and scratch use of the library's kernels, read from the gfx950 code object inside
# pylbl_amd/liblbl_amd.so (no GPU needed).  Usage: scripts/checks/kernel_registers.sh [pattern]
# Also prints the md5 of the device code's .text: unchanged by edits that only move host code.
set -e
LIB=${LIB:-pylbl_amd/liblbl_amd.so}
TMP=$(mktemp -d)
LLVM=/opt/rocm/lib/llvm/bin
$LLVM/llvm-objcopy --dump-section .hip_fatbin=$TMP/fatbin $LIB
$LLVM/clang-offload-bundler --unbundle --type=o --input=$TMP/fatbin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$TMP/co
$LLVM/llvm-objcopy --dump-section .text=$TMP/text $TMP/co
echo "device .text md5: $(md5sum < $TMP/text | cut -d' ' -f1)"
$LLVM/llvm-readelf --notes $TMP/co | python3 -c "
import re, sys
text = sys.stdin.read()
pattern = sys.argv[1] if len(sys.argv) > 1 else ''
for block in text.split('- .agpr_count')[1:]:
    name = re.search(r'\.name:\s+(\S+)', block)
    if not name or pattern not in name.group(1):
        continue
    get = lambda key: (re.search(r'\.' + key + r':\s+(\d+)', block) or [None, '?'])[1]
    print('%-110s vgpr %3s sgpr %3s spill(v/s) %s/%s scratch %s lds %s' % (
        name.group(1)[:110], get('vgpr_count'), get('sgpr_count'), get('vgpr_spill_count'),
        get('sgpr_spill_count'), get('private_segment_fixed_size'), get('group_segment_fixed_size')))
" "$1"
rm -rf $TMP
