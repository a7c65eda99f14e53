"""Per-step kernel-time breakdown from a rocprofv3 kernel trace of bench.py (steady-state steps only).
  rocprofv3 --kernel-trace --output-format csv -d DIR -o tr -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline
  python tools/step_breakdown.py DIR/tr_kernel_trace.csv [out.csv]
Steps are delimited by the Adafactor update kernel (`af_stats`, one launch per step); the first 3 and the last delimited
intervals are dropped (warm-up / teardown)."""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:70]


def main():
    import gzip
    rows = list(csv.DictReader(gzip.open(sys.argv[1], "rt") if sys.argv[1].endswith(".gz") else open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [int(r["Start_Timestamp"]) for r in rows if "af_stats" in r["Kernel_Name"] or "k1_stats" in r["Kernel_Name"]]
    if len(marks) < 5:
        raise SystemExit("could not find the per-step Adafactor launches (af_stats)")
    # one mark per step: collapse marks closer than 20 ms
    steps = [marks[0]]
    for m in marks[1:]:
        if m - steps[-1] > 20e6:
            steps.append(m)
    lo, hi = steps[2], steps[-1]
    n = len(steps) - 3
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rows:
        t0, t1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if lo <= t0 < hi:
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += t1 - t0
    tot = sum(v[1] for v in agg.values())
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    out.write(f"# {n} steady-state steps, wall {(hi - lo) / n / 1e6:.2f} ms/step, kernel time {tot / n / 1e6:.2f} ms/step\n")
    out.write("kernel,launches_per_step,avg_us,ms_per_step,percent\n")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.write(f"{k},{v[0] / n:.1f},{v[1] / v[0] / 1e3:.1f},{v[1] / n / 1e6:.3f},{100 * v[1] / tot:.2f}\n")


if __name__ == "__main__":
    main()
