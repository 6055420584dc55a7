#!/bin/bash
# VGPR / SGPR / spill / LDS figures of every kernel in one .hip source (device-only compile for gfx950)
# usage: scripts/kernel_regs.sh pathintegralgroundstate_amd/csrc/pigs_sampler.hip [extra hipcc flags]
src=$1; shift
out=/tmp/kernel_regs_$$.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I"$(dirname "$0")/../include" \
    --offload-device-only -S "$src" -o $out "$@" || exit 1
python3 - $out <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
meta = txt[txt.rindex("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    print("%-70s vgpr %s sgpr %s spill_v %s spill_s %s lds %s scratch %s" % (g("name")[:70], g("vgpr_count"), g("sgpr_count"),
          g("vgpr_spill_count"), g("sgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
PY
[ -n "$KEEP_ASM" ] && cp $out "$KEEP_ASM"
rm -f $out
