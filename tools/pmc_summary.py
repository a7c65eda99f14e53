"""Per-kernel wave-state summary of a rocprofv3 --pmc run (counter_collection.csv):
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
            SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d DIR -o pmc -- <program>
  python tools/pmc_summary.py DIR/pmc_counter_collection.csv [name filter]
parked = at s_waitcnt / s_barrier, issue-stall = wants to issue but cannot (pipe busy / dependency); SQ wave counters are
quad-cycles, MFMA busy is cycles summed over 1024 SIMDs, GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import collections
import csv
import sys


def main():
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(sys.argv[1])):
        k = r["Kernel_Name"]
        if flt not in k:
            continue
        name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
        agg[name + " grid=" + r["Grid_Size"]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
        w = a["SQ_WAVE_CYCLES"]
        if w <= 0:
            continue
        gui = a["GRBM_GUI_ACTIVE"] / 8
        print(f"{k:64s} parked {a['SQ_WAIT_ANY'] / w:.2f}  issue-stall {a['SQ_WAIT_INST_ANY'] / w:.2f}  active {a['SQ_ACTIVE_INST_ANY'] / w:.2f}"
              f"  valu {a['SQ_ACTIVE_INST_VALU'] / w:.2f}  lds-stall {a['SQ_WAIT_INST_LDS'] / w:.2f}  mfma-busy {a['SQ_VALU_MFMA_BUSY_CYCLES'] / max(gui * 1024, 1):.2f}")


if __name__ == "__main__":
    main()
