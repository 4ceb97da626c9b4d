#!/bin/bash
# tools/kres.sh <pattern>: registers / scratch / LDS of the kernels whose mangled name matches, from a device-only assembly of crsdr.hip
set -e
cd "$(dirname "$0")/../coherent-rtlsdr_amd/csrc"
OUT=${KRES_S:-/tmp/crsdr_kres.s}
if [ ! -f "$OUT" ] || [ crsdr.hip -nt "$OUT" ] || [ -n "$(find . -name '*.hpp' -newer "$OUT")" ]; then
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize --cuda-device-only -S -o "$OUT" crsdr.hip 2>/dev/null
fi
python3 - "$OUT" "$1" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2])
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)(?=\n  - \.|\Z)", txt, re.S):
    pass
# amdhsa.kernels metadata block
meta = txt[txt.rfind("amdhsa.kernels:"):]
for blk in meta.split("\n  - ")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if not name or not pat.search(name.group(1)): continue
    g = lambda k: (re.search(r"\." + k + r":\s+(\d+)", blk) or [None, "?"])[1]
    print(f"{name.group(1)[:90]:90s} vgpr {g('vgpr_count'):>4s} (spilled {g('vgpr_spill_count')}) sgpr {g('sgpr_count'):>4s} (spilled {g('sgpr_spill_count')}) scratch {g('private_segment_fixed_size'):>4s} B lds {g('group_segment_fixed_size')}")
PY
