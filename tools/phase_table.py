"""Differences between consecutive 'exit' runs of tools/phase_valu.sh / tools/phase_c4.sh: the phases' own counts.

    python tools/phase_table.py gpurun_out/r03/c4_phase_counts.txt
"""
import re
import sys

t = open(sys.argv[1]).read()
blocks = re.split(r"== exit (\d+)", t)[1:]
names = sys.argv[2].split(",") if len(sys.argv) > 2 else None
prev, whole = {}, {}
for k in range(0, len(blocks), 2):
    x = int(blocks[k])
    d = {m.group(1): float(m.group(2)) for m in re.finditer(r"(SQ_\w+)\s+launches=\d+ mean=\s*([\d.]+)", blocks[k + 1])}
    if not d:
        continue
    label = names[k // 2] if names and k // 2 < len(names) else f"up to exit {x}"
    cols = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"]
    print(f"{label:28s} " + "  ".join(f"{c[3:]} {(d.get(c, 0) - prev.get(c, 0)) / 1e6:7.2f}M" for c in cols))
    prev = d
    if x == 0:
        whole = d
print(f"{'whole kernel (exit 0)':28s} " + "  ".join(f"{c[3:]} {whole.get(c, 0) / 1e6:7.2f}M" for c in cols))
