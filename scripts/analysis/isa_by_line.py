"""Static instruction counts of one kernel by source line, from `hipcc -gline-tables-only -S` output:
    python scripts/analysis/isa_by_line.py <file.s> <kernel symbol prefix> [top N]
Prints instructions per (file, line) and per file-level function range given on the command line is left to the reader:
the table answers "which statements does the compiler spend the kernel's instructions on"."""
import collections
import re
import sys

path, sym = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
files = {}
inside = False
cur = None
cnt = collections.Counter()
fcnt = collections.Counter()
for ln in open(path):
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', ln)
    if m:
        files[int(m.group(1))] = m.group(2)
        continue
    if ln.startswith(sym + ":"):
        inside = True
        continue
    if not inside:
        continue
    if "s_endpgm" in ln:
        break
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', ln)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    if re.match(r'\s+(v_|s_|ds_|global_|scratch_|buffer_|flat_)', ln) and cur:
        cnt[cur] += 1
        fcnt[cur[0]] += 1
tot = sum(cnt.values())
print("instructions: %d" % tot)
for f, n in fcnt.most_common():
    print("  %-28s %6d  %4.1f %%" % (files.get(f, f), n, 100.0 * n / tot))
print("by line:")
for (f, l), n in cnt.most_common(top):
    print("  %-24s:%-5d %5d" % (files.get(f, f), l, n))
