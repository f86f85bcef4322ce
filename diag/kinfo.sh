#!/bin/bash
# VGPR / SGPR / scratch / LDS of the kernels in a built library (every gfx950 code object in it):
#   diag/kinfo.sh [lib] [name-regex]
lib=${1:-flo_amd/libflo_hip.so}; pat=${2:-chain2x}
tmp=$(mktemp -d)
python3 - "$lib" "$tmp" <<'PY'
import struct, sys
d = open(sys.argv[1], 'rb').read()
i = 0; n = 0
while True:
    i = d.find(b'__CLANG_OFFLOAD_BUNDLE__', i)
    if i < 0: break
    cnt = struct.unpack_from('<Q', d, i + 24)[0]; o = i + 32
    for _ in range(cnt):
        off, size, ts = struct.unpack_from('<QQQ', d, o); o += 24; t = d[o:o + ts]; o += ts
        if b'gfx950' in t:
            open(f'{sys.argv[2]}/dev{n}.co', 'wb').write(d[i + off:i + off + size]); n += 1
    i += 24
PY
for f in $tmp/dev*.co; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $f 2>/dev/null | awk -v pat="$pat" '
    /- \.agpr_count:|^ *- \.args:/ { if (name != "" && name ~ pat) print name, "vgpr", v, "sgpr", s, "scratch", p, "lds", g; name="" }
    /\.name:/ && !/\.name: *[a-z_]*$/ {nm=$2} /\.symbol:/ {name=$2; sub(/\.kd$/, "", name)} /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2}
    /\.private_segment_fixed_size:/ {p=$2} /\.group_segment_fixed_size:/ {g=$2}
    END { if (name != "" && name ~ pat) print name, "vgpr", v, "sgpr", s, "scratch", p, "lds", g }'
done
rm -rf $tmp
