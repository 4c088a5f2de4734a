#!/usr/bin/env python3
"""VGPR / SGPR / spill / LDS figures of every kernel in a hipcc -S listing (the amdhsa metadata at its end).
    python tools/profile/kernel_regs.py file.s [name filter]"""
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if flt not in name:
        continue
    g = lambda k: re.search(r"\." + k + r":\s+(\d+)", blk).group(1)
    print(f"{name[:90]:90s} vgpr {g('vgpr_count'):>3} spill {g('vgpr_spill_count'):>3} sgpr {g('sgpr_count'):>3} scratch {g('private_segment_fixed_size'):>4} lds {g('group_segment_fixed_size'):>6}")
