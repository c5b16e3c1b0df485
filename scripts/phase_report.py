"""Prints the phase-timer breakdown from a bench log produced with DS_PHASE_TIMERS=1 DS_PHASE_DUMP=1."""
import re, sys
names = {0: "setup", 1: "tile start: item map", 8: "tile start: pointers, bounds, tables", 10: "scatter sparse: locate + request (wave 0)", 2: "scatter dense", 3: "scan loop", 4: "select", 5: "exact",
         6: "scatter sparse: wait at the barrier", 15: "scatter sparse: wait for the quads + atomics (wave 0)", 7: "collect sparse: wait at the barrier", 13: "collect sparse: sweep (wave 0)",
         14: "collect sparse: refinement (wave 0)", 9: "zero pass", 11: "probe",
         12: "scan tail"}
line = [l for l in open(sys.argv[1]) if l.startswith("phase cycles:")][-1]
values = {int(a): int(b) for a, b in re.findall(r"(\d+)=(\d+)", line)}
total = sum(values.values())
for i, v in sorted(values.items(), key=lambda kv: -kv[1]):
    if v:
        print(f"{str(names.get(i, i)):30s} {100.0 * v / total:5.1f}%")
