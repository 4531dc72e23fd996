"""Static VALU-instruction count per source line of one kernel (hipcc -gline-tables-only -S output): where a VALU-bound kernel's
instructions come from.  Usage: python tools/isa_profile.py <asm.s> <mangled-kernel-prefix> <source.hip> [top]"""
import collections
import re
import sys

asm, name, srcf = sys.argv[1:4]
top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
txt = open(asm).read().split("\n")
src = open(srcf).read().split("\n")
inside = False
cur = None
cnt = collections.Counter()
kinds = collections.Counter()
for l in txt:
    if not inside:
        if l.startswith(name) and ":" in l.split(";")[0]:
            inside = True
        continue
    if "s_endpgm" in l:
        break
    m = re.match(r"\s*\.loc\s+\d+\s+(\d+)", l)
    if m:
        cur = int(m.group(1))
        continue
    s = l.strip()
    if not s or s.startswith((".", ";", "//")) or s.split(";")[0].strip().endswith(":"):
        continue
    op = s.split()[0]
    kinds[op.split("_")[0] + ("_f64" if "f64" in op else "")] += 1
    if op.startswith("v_"):
        cnt[cur] += 1
print(name, "VALU", sum(cnt.values()), dict(kinds))
for line, c in sorted(cnt.items(), key=lambda x: -x[1])[:top]:
    print(f"{c:5d}  L{line}: {src[line - 1].strip()[:130] if line else ''}")
