#!/bin/bash
# per-kernel VGPRs / scratch / LDS of one object file (amdgpu code-object metadata): tools/kernel_resources.sh <obj> [name regex]
# e.g. tools/kernel_resources.sh uncrtaints_amd/lib/obj/pw_gemm_split_p2.o pw_gemm_split
set -e
OBJ=$1; export FILT=${2:-.}
TMP=$(mktemp -d)
LL=/opt/rocm/lib/llvm/bin
$LL/llvm-objcopy --dump-section .hip_fatbin=$TMP/fat.bin "$OBJ"
$LL/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$TMP/fat.bin --output=$TMP/dev.co
$LL/llvm-readelf --notes $TMP/dev.co | python3 -c "
import sys, re, os, subprocess
txt = sys.stdin.read()
for blk in re.split(r'\n\s*- \.agpr_count:', txt)[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s*(\S+)', blk) or [None, '?'])[1]
    name = subprocess.run(['c++filt', g('name')], capture_output=True, text=True).stdout.strip()
    if re.search(os.environ['FILT'], name):
        print(f\"{name[:90]:90s} vgpr={g('vgpr_count'):>4s} agpr={blk.split()[0]:>4s} sgpr={g('sgpr_count'):>4s} scratch={g('private_segment_fixed_size'):>5s} lds={g('group_segment_fixed_size'):>6s}\")
"
rm -rf "$TMP"
