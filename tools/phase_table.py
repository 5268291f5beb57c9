"""Per-phase differences of the counters printed by tools/phase_c4.sh / tools/phase_lds.sh (one `== exit k` block per
phase boundary, exit 0 = the whole kernel last): python tools/phase_table.py <file> [name,name,...]"""
import re
import sys

d, cur = {}, None
for line in open(sys.argv[1]):
    m = re.match(r"== exit (\d)", line)
    if m:
        cur = int(m.group(1)); d[cur] = {}
        continue
    m = re.match(r"(\S+)\s+launches=\s*\d+ mean=\s*([\d.]+)", line)
    if m and cur is not None:
        d[cur][m.group(1)] = float(m.group(2))
order = [k for k in d if k != 0] + [0]
names = sys.argv[2].split(",") if len(sys.argv) > 2 else [f"to exit {k}" for k in order]
cols = list(d[order[0]])
print(f"{'phase':12s}" + "".join(f"{c[3:]:>18s}" for c in cols) + "   (millions per launch)")
prev = {c: 0.0 for c in cols}
for x, n in zip(order, names):
    print(f"{n:12s}" + "".join(f"{(d[x][c] - prev[c]) / 1e6:18.2f}" for c in cols))
    prev = d[x]
print(f"{'total':12s}" + "".join(f"{d[0][c] / 1e6:18.2f}" for c in cols))
