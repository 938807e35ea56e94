"""Build helpers for the gfx950 shared library.

``build_hip()`` compiles every ``csrc/*.hip`` with ``hipcc --offload-arch=gfx950``
into ``deep-image-matching_amd/lib/libdim_hip.so`` (in-tree, so it travels to the
GPU box with the snapshot).  ``build_emu()`` compiles the *same* sources with the
host clang against the test-only HIP emulator under ``tests/hipemu`` — that
library is loaded only by the CPU tests, never by this package.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HOST_CLANG = os.environ.get("DIM_HOST_CLANG", "/opt/rocm/lib/llvm/bin/clang++")


def _sources():
    return sorted(CSRC.glob("*.hip"))


def _digest(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def _deps():
    return list(CSRC.glob("*.h")) + list((ROOT / "include").glob("*.h"))


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s\n%s" % (" ".join(map(str, cmd)), r.stdout, r.stderr))
    return r


def _compile_all(objdir: Path, compile_cmd, srcs, tag):
    objdir.mkdir(parents=True, exist_ok=True)
    deps_digest = _digest(_deps(), tag)
    jobs = []
    objs = []
    for s in srcs:
        o = objdir / (s.stem + ".o")
        stamp = objdir / (s.stem + ".sha")
        d = _digest([s], deps_digest)
        objs.append(o)
        if o.exists() and stamp.exists() and stamp.read_text() == d:
            continue
        jobs.append((s, o, stamp, d))

    def work(j):
        s, o, stamp, d = j
        _run(compile_cmd(s, o))
        stamp.write_text(d)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(work, jobs))
    return objs, bool(jobs)


def build_hip(verbose: bool = False) -> Path:
    """hipcc --offload-arch=gfx950 build of the product library."""
    out = LIBDIR / "libdim_hip.so"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]
    objs, changed = _compile_all(
        PKG / "build" / "hip", lambda s, o: [HIPCC, *flags, "-c", str(s), "-o", str(o)], _sources(), "hip" + " ".join(flags)
    )
    if changed or not out.exists():
        LIBDIR.mkdir(exist_ok=True)
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(out), *map(str, objs)])
    if verbose:
        print("built", out)
    return out


def build_variant(name: str, defines, verbose: bool = False) -> Path:
    """The same sources with extra -D flags -> lib/libdim_hip_<name>.so (measurement builds for scripts/: A/B of a code path that is a
    compile-time choice, instrumented kernels).  Never loaded by the package."""
    out = LIBDIR / f"libdim_hip_{name}.so"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result", *defines]
    objs, changed = _compile_all(
        PKG / "build" / f"hip_{name}", lambda s, o: [HIPCC, *flags, "-c", str(s), "-o", str(o)], _sources(), "hip" + " ".join(flags)
    )
    if changed or not out.exists():
        LIBDIR.mkdir(exist_ok=True)
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(out), *map(str, objs)])
    if verbose:
        print("built", out)
    return out


RESEARCH_DEFINES = ["-DDIM_RESEARCH"]


def build_research(verbose: bool = False) -> Path:
    """lib/libdim_hip_research.so: the product sources + the default-off prototypes that lost their A/B (Winograd conv1b, 64-wide K chunks,
    double-buffered / tile-refilled GEMM blocks, round 3's feed-forward loop) and the timing probes that give wrong results by design
    (dim_tune_set keys 12-15).  Loaded only by the A/B scripts under scripts/ and by the tests of those variants — never by the package."""
    return build_variant("research", RESEARCH_DEFINES, verbose)


def build_emu(verbose: bool = False, research: bool = False) -> Path:
    """Host-clang build of the same sources against tests/hipemu (CPU tests only).  research=True: with -DDIM_RESEARCH (the prototype
    variants' CPU tests), a separate library next to the plain one."""
    emu = ROOT / "tests" / "hipemu"
    out = emu / ("libdim_hip_emu_research.so" if research else "libdim_hip_emu.so")
    flags = ["-std=c++17", "-O2", "-march=native", "-fPIC", "-ffp-contract=off", "-Wno-psabi", "-Wno-unused-value", "-I", str(emu / "include")]
    if research:
        flags += RESEARCH_DEFINES
    srcs = _sources()
    objs, changed = _compile_all(
        emu / ("build_research" if research else "build"),
        lambda s, o: [HOST_CLANG, *flags, "-x", "c++", "-c", str(s), "-o", str(o)],
        srcs,
        "emu" + " ".join(flags) + _digest([emu / "include" / "hip" / "hip_runtime.h"]),
    )
    (emu / "build").mkdir(parents=True, exist_ok=True)
    rt = emu / "build" / "hipemu_rt.o"
    rt_stamp = emu / "build" / "hipemu_rt.sha"
    d = _digest([emu / "hipemu.cpp", emu / "include" / "hip" / "hip_runtime.h"])
    if not rt.exists() or not rt_stamp.exists() or rt_stamp.read_text() != d:
        _run([HOST_CLANG, *flags, "-c", str(emu / "hipemu.cpp"), "-o", str(rt)])
        rt_stamp.write_text(d)
        changed = True
    if changed or not out.exists():
        _run([HOST_CLANG, "-shared", "-fPIC", "-o", str(out), *map(str, objs), str(rt)])
    if verbose:
        print("built", out)
    return out


if __name__ == "__main__":
    import sys

    if len(sys.argv) > 1 and sys.argv[1] == "emu":
        build_emu(True)
    else:
        build_hip(True)
