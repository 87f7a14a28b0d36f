#!/bin/bash
# Register / scratch / LDS use of every kernel in libapgpu.so (compiles each translation unit to assembly):
#   bash tools/kernel_resources.sh > profiles/<round>/kernel_resources.csv
REPO=$(cd "$(dirname "$0")/.." && pwd)
CS=$REPO/astrophotography_amd/csrc
TMP=$(mktemp -d)
echo "translation_unit,kernel,vgpr,sgpr_spill,vgpr_spill,scratch_bytes,lds_bytes"
for src in $CS/*.hip; do
  b=$(basename $src .hip)
  EXTRA=""; case $b in stack_inst_*) EXTRA="-mllvm -disable-machine-licm";; esac     # as _build.py's STACK_TU_FLAGS
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math $EXTRA -S --cuda-device-only -I$CS -I$REPO/include $src -o $TMP/$b.s 2>/dev/null &
done
wait
for src in $CS/*.hip; do
  b=$(basename $src .hip)
  python3 - $TMP/$b.s $b <<'PY'
import re,sys
txt=open(sys.argv[1]).read()
for blk in txt.split('  - .agpr_count:')[1:]:
    g=lambda k: (re.search(r'\.%s:\s+(\S+)'%k, blk) or [None,'?'])[1]
    name=g('name')
    try:
        import subprocess
        dem=subprocess.run(['c++filt', name],capture_output=True,text=True).stdout.strip()
    except Exception:
        dem=name
    dem=dem.replace(',',';')
    print('%s,"%s",%s,%s,%s,%s,%s'%(sys.argv[2],dem[:150],g('vgpr_count'),g('sgpr_spill_count'),g('vgpr_spill_count'),g('private_segment_fixed_size'),g('group_segment_fixed_size')))
PY
done
rm -rf $TMP
