"""HBM traffic of one kernel from two rocprofv3 passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950):
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d D1 -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d D2 -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
  python tools/pmc_traffic.py D1/pmc_counter_collection.csv D2/pmc_counter_collection.csv gemm_bf16_nt_256_kernel out.json
Counters are in KiB-units of 1024 B as rocprofv3 reports them; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies
128-byte requests as 64 B)."""
import csv
import json
import sys


def total(path, counter, name):
    s, ids = 0.0, set()
    for r in csv.DictReader(open(path)):
        if name in r["Kernel_Name"] and r["Counter_Name"] == counter:
            s += float(r["Counter_Value"])
            ids.add(r["Dispatch_Id"])
    return s, len(ids)


def main():
    f, nf = total(sys.argv[1], "FETCH_SIZE", sys.argv[3])
    w, nw = total(sys.argv[2], "WRITE_SIZE", sys.argv[3])
    out = {"kernel": sys.argv[3], "launches": nf, "fetch_bytes_per_launch": 2.0 * f * 1024 / max(nf, 1),
           "write_bytes_per_launch": w * 1024 / max(nw, 1),
           "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 2 --warmup 1 --no-cpu-baseline`; "
                   "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); units KiB -> bytes"}
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
