"""Build libdesta_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python desta2.5-audio_amd/build.py [--force]

One object per csrc/*.hip (compiled in parallel), linked into desta/lib/libdesta_hip.so.  The
library links against libamdhip64.so.7 only (the SONAME torch's bundled runtime also carries, so
inside a torch process the already-loaded runtime is reused).  No torch headers are involved.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "desta", "lib")
OBJ_DIR = os.path.join(HERE, "build")
LIB = os.path.join(OUT_DIR, "libdesta_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast",
         "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-Wno-unused-result"]
FLAGS += os.environ.get("DESTA_EXTRA_HIPCC_FLAGS", "").split()      # kernel A/B experiments (-DNAME=value)
# gemm_bf16.hip: accumulators stay in VGPRs for every kernel of the file.  The one-block-per-CU ring kernel would otherwise get
# AGPR accumulators (512-register budget) and, with two MFMA groups per loop iteration, ~110 v_accvgpr copies per K-tile; the
# other kernels of the file already compile to the VGPR form (identical code with and without the flag).
FILE_FLAGS = {"gemm_bf16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode() + b"\0" + f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(ROOT, "include", "desta_hip.h"))
    os.makedirs(OUT_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    stamp = os.path.join(OBJ_DIR, "stamp")
    hdr_dig = _digest(hdrs)
    objs, todo = [], []
    for s in srcs:
        o = os.path.join(OBJ_DIR, os.path.basename(s)[:-4] + ".o")
        dig = _digest([s]) + hdr_dig + " ".join(FILE_FLAGS.get(os.path.basename(s), []))
        dfile = o + ".sha"
        objs.append(o)
        if force or not os.path.exists(o) or not os.path.exists(dfile) or open(dfile).read() != dig:
            todo.append((s, o, dfile, dig))
    if not todo and os.path.exists(LIB) and not force:
        return LIB

    def cc(job):
        s, o, dfile, dig = job
        cmd = [HIPCC, *FLAGS, *FILE_FLAGS.get(os.path.basename(s), []), "-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr[-6000:]}")
        with open(dfile, "w") as f:
            f.write(dig)
        if verbose:
            print(f"[build] {os.path.basename(s)} ok", flush=True)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(todo)))) as ex:
        list(ex.map(cc, todo))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl",
           "-Wl,-rpath,/opt/rocm/lib", "-Wl,-soname,libdesta_hip.so"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-6000:]}")
    if verbose:
        print(f"[build] linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
