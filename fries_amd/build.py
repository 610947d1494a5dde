"""Builds libfries_hip.so (hand-written HIP for gfx950 + the C ABI) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container and on the GPU box
alike.  -ffp-contract=off is part of the numerical contract: the engine reproduces the
reference's plain IEEE double arithmetic, so a*b+c must never be fused.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libfries_hip.so")
SOURCES = ["hbpp.hip", "vec.hip", "compress.hip", "pivotal.hip", "system.hip", "hh.hip", "fciqmc.hip", "driver.hip", "comm_native.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wno-unused-value", "-Wno-unused-result"] + os.environ.get("FRIES_EXTRA_HIPCC_FLAGS", "").split()      # (diagnostic builds: -DFR_SEQ_TIMING, -DFR_FKS_CHKDBG)


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(HERE, "..", "include", "fries_hip.h"))

    def compile_one(src: str) -> str:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        if force or _stale(obj, [path] + headers):
            cmd = [hipcc] + FLAGS + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    build_drivers(force)
    return LIB


DRIVER = os.path.join(HERE, "frisys_mol_hip")
REF_API_DRIVER = os.path.join(HERE, "..", "tests", "cpp", "frisys_mol_ref_api")        # a TEST consumer of include/FRIES (tests/cpp/frisys_mol_ref_api.cpp), not part of the product
FIXED_CLOCK = os.path.join(HERE, "..", "tests", "cpp", "libfixed_clock.so")
DRIVERS = {name: os.path.join(HERE, name) for name in ("frisys_mol_hip", "fciqmc_mol_hip", "frisys_hh_hip", "frifull_mol_hip", "frimulti_mol_hip")}


def build_drivers(force: bool = False) -> str:
    """The C++ command-line drivers (host side in the reference's language) linked against libfries_hip.so."""
    hdrs = [os.path.join(HERE, "..", "include", "fries_hip.h"), os.path.join(HERE, "drivers", "driver_common.hpp")]
    for name, exe in DRIVERS.items():
        src = os.path.join(HERE, "drivers", name + ".cpp")
        if force or _stale(exe, [src, LIB] + hdrs):
            cmd = ["g++", "-std=c++17", "-O2", "-o", exe, src, "-L" + HERE, "-lfries_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib",
                   "-Wl,-rpath-link,/opt/rocm/lib"]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"g++ failed on {name}.cpp:\n{r.stdout}\n{r.stderr}")
    # a test program written against the reference's own classes as this build ships them (include/FRIES/*.hpp)
    inc = os.path.join(HERE, "..", "include")
    fac_hdrs = [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(inc, "FRIES")) for f in fs]
    src = os.path.join(HERE, "..", "tests", "cpp", "frisys_mol_ref_api.cpp")
    if force or _stale(REF_API_DRIVER, [src, LIB] + hdrs + fac_hdrs):
        cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-I" + inc, "-I" + os.path.join(inc, "FRIES", "compat"), "-o", REF_API_DRIVER, src,
               "-L" + HERE, "-lfries_hip", "-Wl,-rpath,$ORIGIN/../../fries_amd", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"g++ failed on frisys_mol_ref_api.cpp:\n{r.stdout}\n{r.stderr}")
    # LD_PRELOAD shim that pins std::chrono::system_clock::now() (tests seed the reference's own driver source through it)
    src = os.path.join(HERE, "..", "tests", "cpp", "fixed_clock.cpp")
    if force or _stale(FIXED_CLOCK, [src]):
        r = subprocess.run(["g++", "-std=c++17", "-O2", "-shared", "-fPIC", "-o", FIXED_CLOCK, src], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"g++ failed on fixed_clock.cpp:\n{r.stdout}\n{r.stderr}")
    return DRIVER


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
