"""Top kernels of a rocprofv3 `--kernel-trace --stats --output-format csv` run:  python tools/kstats.py DIR/x_kernel_stats.csv [N]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in rows[:n]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"]) / 1e3:9.1f} us  total {float(r["TotalDurationNs"]) / 1e6:9.2f} ms  {float(r["Percentage"]):5.1f} %')
