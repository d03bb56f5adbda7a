"""Register / LDS / scratch use of every kernel in a --save-temps .s file:  python scripts/analysis/kernel_resources.py file.s"""
import re
import sys

s = open(sys.argv[1]).read()
meta = s[s.rindex("amdhsa.kernels:"):]
for blk in re.split(r"\n  - ", meta)[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    print("%-70s vgpr %s agpr %s vspill %s sgpr %s sspill %s lds %s scratch %s maxwg %s" % (
        g("name")[:70], g("vgpr_count"), g("agpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"),
        g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("max_flat_workgroup_size")))
