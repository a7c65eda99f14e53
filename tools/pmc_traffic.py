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
    if len(sys.argv) > 5:
        # DRAM side (round 4): TCC_EA0_RDREQ_sum = read requests leaving the L2s towards the fabric, TCC_EA0_RDREQ_DRAM_sum = those that
        # went to DRAM (the rest were served by the Infinity Cache / peers); request sizes mix 32 / 64 / 128 B, so the fraction is of
        # REQUESTS; with the optional second file (TCC_EA0_RDREQ_32B_sum, TCC_BUBBLE_sum = 128-B requests) bytes are estimated as
        # 128 * bubble + 64 * (rdreq - bubble - rd32) + 32 * rd32 (the FETCH_SIZE expression of rocprofv3 --list-avail)
        rq, n1 = total(sys.argv[5], "TCC_EA0_RDREQ_sum", sys.argv[3])
        dr, _ = total(sys.argv[5], "TCC_EA0_RDREQ_DRAM_sum", sys.argv[3])
        out["ea_read_requests_per_launch"] = rq / max(n1, 1)
        out["ea_read_requests_to_dram_per_launch"] = dr / max(n1, 1)
        out["hbm_read_fraction_of_fetch"] = dr / max(rq, 1.0)
        out["dram_read_counter"] = "TCC_EA0_RDREQ_DRAM_sum / TCC_EA0_RDREQ_sum (requests)"
        if len(sys.argv) > 6:
            r32, n2 = total(sys.argv[6], "TCC_EA0_RDREQ_32B_sum", sys.argv[3])
            bub, _ = total(sys.argv[6], "TCC_BUBBLE_sum", sys.argv[3])
            if n2:
                bytes_all = 128.0 * bub / n2 + 64.0 * (rq / max(n1, 1) - bub / n2 - r32 / n2) + 32.0 * r32 / n2
                out["ea_read_bytes_per_launch_from_request_mix"] = bytes_all
                out["dram_read_bytes_per_launch"] = bytes_all * out["hbm_read_fraction_of_fetch"]
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
