"""Registers / LDS / scratch of every gfx950 kernel in libdesta_hip.so, read from the code objects' metadata notes.
  python tools/kernel_resources.py [filter]
Scratch (`private_segment_fixed_size` > 0) in a hot kernel means hipcc parked live values in memory: round 3 lost the 128
accumulators of the 256x256 GEMM to a 528-byte stack frame that way (an epilogue helper grew past the full-unroll budget; HBM
writes per launch went 106 -> 472 MB) — tests/test_abi.py::test_hot_kernels_use_no_scratch keeps that from coming back."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "desta2.5-audio_amd", "desta", "lib", "libdesta_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_resources(lib=LIB):
    """-> {demangled-ish kernel name: dict(vgpr, sgpr, lds, scratch, spill)}"""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
                def num(key):
                    m = re.search(r"\." + key + r":\s+(\d+)", blk)
                    return int(m.group(1)) if m else 0
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                out[name] = dict(vgpr=num("vgpr_count"), sgpr=num("sgpr_count"), lds=num("group_segment_fixed_size"),
                                 scratch=num("private_segment_fixed_size"), spill=num("vgpr_spill_count"))
    return out


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    res = kernel_resources()
    for name in sorted(res):
        if flt in name:
            r = res[name]
            short = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt"), name], capture_output=True, text=True).stdout.strip() if os.path.exists(os.path.join(LLVM, "llvm-cxxfilt")) else name
            short = short.replace("(anonymous namespace)::", "").split("(")[0]
            print(f"{short[:70]:70s} vgpr {r['vgpr']:3d} sgpr {r['sgpr']:3d} lds {r['lds']:6d} scratch {r['scratch']:4d} spilled {r['spill']:3d}")


if __name__ == "__main__":
    main()
