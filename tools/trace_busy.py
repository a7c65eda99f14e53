"""GPU busy time from a `rocprofv3 --kernel-trace` CSV: the union of all kernel intervals (any stream) over the steady-state window, the
idle remainder, the largest idle gaps with the kernels on either side, and the per-queue busy time.
`python tools/trace_busy.py <kernel_trace.csv> [--skip-first-frac 0.4 | --last-ms 1200]`"""
import csv
import sys


def main():
    path = sys.argv[1]
    skip = float(sys.argv[sys.argv.index("--skip-first-frac") + 1]) if "--skip-first-frac" in sys.argv else 0.4
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    t_first, t_last = rows[0][0], max(r[1] for r in rows)
    t0 = t_first + int((t_last - t_first) * skip)
    if "--last-ms" in sys.argv:                                           # steady state: the last X ms of the trace
        t0 = t_last - int(float(sys.argv[sys.argv.index("--last-ms") + 1]) * 1e6)
    rows = [r for r in rows if r[0] >= t0]
    span = max(r[1] for r in rows) - rows[0][0]
    busy, cur_s, cur_e, gaps = 0, rows[0][0], rows[0][1], []
    last_name = rows[0][2]
    for s, e, name, q in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, last_name, name))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        if e >= cur_e:
            last_name = name
    busy += cur_e - cur_s
    per_q = {}
    for s, e, name, q in rows:
        per_q[q] = per_q.get(q, 0) + (e - s)

    def short(n):
        n = n[5:] if n.startswith("void ") else n
        return n.replace("(anonymous namespace)::", "").split("(")[0][:60]
    print(f"window {span / 1e6:.2f} ms, union busy {busy / 1e6:.2f} ms ({100.0 * busy / span:.2f} %), idle {(span - busy) / 1e6:.2f} ms in {len(gaps)} gaps")
    print("kernel time per queue (ms):", {q: round(v / 1e6, 1) for q, v in sorted(per_q.items())})
    hist = {}
    for g, a, b in gaps:
        k = "<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<50us" if g < 50000 else ">=50us"
        hist[k] = (hist.get(k, (0, 0))[0] + 1, hist.get(k, (0, 0))[1] + g)
    print("gaps by size (count, total ms):", {k: (v[0], round(v[1] / 1e6, 3)) for k, v in hist.items()})
    agg = {}
    for g, a, b in gaps:
        k = (short(a), short(b))
        agg[k] = (agg.get(k, (0, 0))[0] + 1, agg.get(k, (0, 0))[1] + g)
    print("idle by (kernel before -> kernel after), top 25 by total:")
    for (a, b), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {tot / 1e6:8.3f} ms  x{n:5d}  avg {tot / n / 1e3:7.1f} us   {a}  ->  {b}")


if __name__ == "__main__":
    main()
