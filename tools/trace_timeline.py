"""Turn a `rocprofv3 --kernel-trace` CSV into a timeline of the LAST `--window` ms of kernels: start offset, duration, gap to the
previous kernel's end (per stream-agnostic wall order), so that launch gaps and serial chains of small kernels show.
`python tools/trace_timeline.py <kernel_trace.csv> [--last N]` prints the last N kernels; `--summary` adds totals."""
import csv
import sys


def main():
    path = sys.argv[1]
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 400
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", ""), r.get("Queue_Id", "")))
    rows.sort()
    rows = rows[-last:]
    t0 = rows[0][0]
    prev_end = rows[0][0]
    busy = 0
    for s, e, name, grid, q in rows:
        gap = (s - prev_end) / 1e3
        busy += (e - s)
        n = name[5:] if name.startswith("void ") else name
        n = n.replace("(anonymous namespace)::", "").split("(")[0][:70]
        print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap:7.1f}  q{q:>3}  grid {grid:>9}  {n}")
        prev_end = max(prev_end, e)
    print(f"# window {(prev_end - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, {len(rows)} kernels")


if __name__ == "__main__":
    main()
