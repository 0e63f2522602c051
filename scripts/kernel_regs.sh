#!/bin/bash
# dev: VGPR / SGPR / scratch of every kernel of an object file built by scripts/build_lib.sh (code-object metadata notes)
obj=${1:-build/plx_ssfm.o}
d=$(mktemp -d)
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy --dump-section .hip_fatbin=$d/fat $obj && $B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$d/fat --output=$d/co --unbundle
$B/llvm-readelf --notes $d/co | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in re.split(r'\n\s+- \.agpr_count', txt)[1:]:
    blk='.agpr_count'+blk
    def g(k):
        r=re.search(r'\.'+k+r':\s+(\S+)', blk); return r.group(1) if r else '-'
    print('%-70s vgpr %s agpr %s sgpr %s scratch %s' % (g('name')[:70], g('vgpr_count'), g('agpr_count'), g('sgpr_count'), g('private_segment_fixed_size')))
"
rm -rf $d
