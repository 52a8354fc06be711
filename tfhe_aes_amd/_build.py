"""In-tree builds of the native libraries (no JIT cache: the .so files travel with the tree).

  tfhe_aes_amd/libfheaes.so         hipcc --offload-arch=gfx950   (HIP kernels + C ABI, the product)
  tfhe_aes_amd/libfheaes_client.so  gcc -fopenmp                   (host Client: keygen / encrypt / decrypt)
(The CPU checker under oracle/ has its own Makefile; the product never builds or loads it.)
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"

ENGINE_SO = PKG / "libfheaes.so"
CLIENT_SO = PKG / "libfheaes_client.so"

# Two translation units (csrc/ks_launch.h says why): engine.hip = everything but the key-switching kernels, compiled with LLVM's
# post-register-allocation scheduler OFF; keyswitch_tu.hip = K1 / K3 / the key-byte conversion, default schedulers.
ENGINE_SOURCES = [CSRC / "engine.hip", CSRC / "keyswitch_tu.hip"]
ENGINE_HEADERS = sorted(CSRC.glob("*.h")) + sorted(CSRC.glob("*.hpp")) + [ROOT / "include" / "fheaes.h"]
BUILD_DIR = PKG / "build"


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).exists() and Path(d).stat().st_mtime > t for d in deps)


def _run(cmd, **kw):
    res = subprocess.run(cmd, capture_output=True, text=True, **kw)
    if res.returncode != 0:
        raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(map(str, cmd)), res.stdout, res.stderr))
    return res


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found")


def _common_flags():
    # -ffp-contract=off: the canonical arithmetic only fuses where the source says fma()
    # -disable-machine-licm: the transforms' ~50 lane-independent twiddle constants are 64-bit literals moved into SGPR pairs;
    #   hoisted out of the 669-iteration loop they would all be live at once (12 SGPRs spilled to a VGPR, 20 B/lane of scratch in
    #   the blind rotation); left where they are used they cost a scalar move each, on the otherwise idle scalar unit
    return [
        "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
        "-ffp-contract=off", "-fno-fast-math", "-mllvm", "-disable-machine-licm", "-Wall",
        "-Wno-unused-function", "-I", str(ROOT / "include"), "-I", str(CSRC),
    ]


# round 6: the post-RA scheduler re-orders what the blind-rotation kernels place by hand; without it a 16,384-bit launch takes
# 197.3 instead of 200.5 ms (same box, same words; profiles/r06_k2_ablations.txt).  The key switch wants it ON (15.9 vs 16.6 ms).
NO_POST_RA_SCHED = ["-mllvm", "-enable-post-misched=false"]


def unit_flags(unit: str):
    """compile flags of one unit of the product library: "engine" (engine.hip) or "keyswitch" (keyswitch_tu.hip)"""
    if unit == "engine":
        return _common_flags() + ["-DFHEAES_SPLIT_KS"] + NO_POST_RA_SCHED
    if unit == "keyswitch":
        return _common_flags() + ["-DFHEAES_SPLIT_KS"]
    raise ValueError(unit)


def engine_flags(unit: str = "engine"):
    """flags of a SINGLE-UNIT developer build (`hipcc <these> -o lib.so csrc/engine.hip`, as tools/ablate_*.py do it): engine.hip alone
    is then a complete library, every kernel in it compiled under the scheduler setting of the product unit named (`unit="keyswitch"`
    for experiments on K1 / K3)"""
    return _common_flags() + ["-shared"] + (NO_POST_RA_SCHED if unit == "engine" else [])


def build_engine(force: bool = False, extra=()) -> Path:
    # this file is a dependency too: it holds the compile flags (a library built before a flag changed must not be reused)
    if force or _stale(ENGINE_SO, ENGINE_SOURCES + ENGINE_HEADERS + [Path(__file__)]):
        BUILD_DIR.mkdir(exist_ok=True)
        objs = []
        for src, unit in zip(ENGINE_SOURCES, ("engine", "keyswitch")):
            obj = BUILD_DIR / (src.stem + ".o")
            _run([hipcc_path()] + unit_flags(unit) + list(extra) + ["-c", "-o", str(obj), str(src)])
            objs.append(str(obj))
        _run([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(ENGINE_SO)] + objs)
    return ENGINE_SO


def build_client(force: bool = False) -> Path:
    src = CSRC / "client.c"
    if force or _stale(CLIENT_SO, [src, ROOT / "include" / "fheaes.h"]):
        _run(["gcc", "-O3", "-mavx2", "-mfma", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-fPIC",
              "-shared", "-std=gnu11", "-Wall", "-o", str(CLIENT_SO), str(src), "-lm"])
    return CLIENT_SO


def engine_source_hash() -> str:
    """sha256 over the engine's sources (csrc/*.hip, csrc/*.h, include/fheaes.h): stamps measurements (profiles/*.json)
    so that a counter summary is only ever attached to a run of the same kernels"""
    import hashlib

    h = hashlib.sha256()
    for f in sorted(ENGINE_SOURCES + ENGINE_HEADERS):
        h.update(Path(f).name.encode())
        h.update(Path(f).read_bytes())
    # ... and the compile flags (without the machine-specific include paths): the same sources under other flags are other kernels
    for unit in ("engine", "keyswitch"):
        h.update(" ".join(x for x in unit_flags(unit) if not x.startswith("/")).encode())
    return h.hexdigest()


def build_all(force: bool = False):
    return build_engine(force), build_client(force)
