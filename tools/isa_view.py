"""Development aid: print the memory / wait / MFMA skeleton of one kernel from a hipcc -save-temps .s file.
    python tools/isa_view.py file.s <mangled-name-substring> [--all]
"""
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
names = [m.group(1) for m in re.finditer(r"^(_Z\S+):", s, re.M) if pat in m.group(1)]
if not names:
    sys.exit("no kernel matches")
name = names[0]
i = s.index(name + ":")
j = s.index("s_endpgm", i)
lines = s[i:j].split("\n")
print(name, len(lines), "lines")
allv = "--all" in sys.argv
counts = {}
for k, l in enumerate(lines):
    t = l.strip()
    op = t.split(" ")[0] if t else ""
    counts[op] = counts.get(op, 0) + 1
    if allv or re.search(r"global_load|s_waitcnt|s_barrier|v_mfma|ds_read|ds_write|s_cbranch|^\.LBB|global_store|buffer_|scratch_", t):
        print(k, t[:120])
print({k: v for k, v in sorted(counts.items(), key=lambda kv: -kv[1])[:25]})
