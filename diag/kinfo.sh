#!/bin/bash
# VGPR / SGPR / scratch / LDS of the kernels in a built library: diag/kinfo.sh [lib] [name-substring]
lib=${1:-flo_amd/libflo_hip.so}; pat=${2:-chain2x}
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$lib --output=$tmp/dev.co --unbundle 2>/dev/null || \
  { python3 - "$lib" "$tmp/dev.co" <<'PY'
import sys
d=open(sys.argv[1],'rb').read()
i=d.find(b'__CLANG_OFFLOAD_BUNDLE__')
import struct
n=struct.unpack_from('<Q',d,i+24)[0]; o=i+32
for _ in range(n):
    off,size,ts=struct.unpack_from('<QQQ',d,o); o+=24; t=d[o:o+ts]; o+=ts
    if b'gfx950' in t: open(sys.argv[2],'wb').write(d[i+off:i+off+size])
PY
  }
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/dev.co | awk -v pat="$pat" '
  /\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.private_segment_fixed_size:/ {p=$2} /\.group_segment_fixed_size:/ {g=$2}
  /\.wavefront_size:/ { if (name ~ pat) print name, "vgpr", v, "sgpr", s, "scratch", p, "lds", g }'
rm -rf $tmp
