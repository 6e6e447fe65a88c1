"""Builds libl3ac_hip.so (the C-ABI extension, include/l3ac_hip.h) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting ``.so`` is git-ignored
but ships to the GPU box with the repo snapshot.  Usage: ``python -m l3ac_amd.build [--force]``.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
REPO = PKG.parent
CSRC = PKG / "csrc"
# L3AC_BUILD_TAG=<tag> builds a second, independent library (libl3ac_hip_<tag>.so, objects under csrc/build_<tag>): diagnostic
# and timing-only builds (L3AC_EXTRA_HIPCC_FLAGS=-D...) next to the product library; load it with L3AC_LIB_PATH.
_TAG = os.environ.get("L3AC_BUILD_TAG", "")
OBJ_DIR = CSRC / (f"build_{_TAG}" if _TAG else "build")
LIB_PATH = PKG / (f"libl3ac_hip_{_TAG}.so" if _TAG else "libl3ac_hip.so")
ARCH = "gfx950"

SOURCES = [
    "capi.hip",
    "network.hip",
    "kernels/gemm_f32.hip",
    "kernels/gemm_split.hip",
    "kernels/gemm_split_w256.hip",
    "kernels/rows.hip",
    "kernels/first_block.hip",
    "kernels/enhance.hip",
    "kernels/attention.hip",
    "kernels/trans_stack.hip",
    "kernels/fsq.hip",
    "kernels/conv_unit_fused.hip",
    "kernels/conv_unit_split.hip",
    "kernels/conv_unit_wide.hip",
    "kernels/conv_unit_ring.hip",
    "kernels/last_block.hip",
    "kernels/bitpack.hip",
    "kernels/up_fused.hip",
]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def _newest_header() -> float:
    """Newest header anywhere under csrc/ (kernels/*.hpp hold the split, epilogue and activation math shared by most
    kernels) or include/: the fallback staleness test for objects that have no depfile yet."""
    hdrs = list(CSRC.rglob("*.hpp")) + list(CSRC.rglob("*.h")) + list((REPO / "include").glob("*.h"))
    return max(h.stat().st_mtime for h in hdrs)


def _depfile_newest(dep: Path) -> float:
    """Newest mtime among the prerequisites hipcc recorded for one object (-MD -MF); inf when one has disappeared."""
    text = dep.read_text().replace("\\\n", " ")
    newest = 0.0
    for tok in text.split(":", 1)[-1].split():
        if tok.startswith("/opt/rocm") or tok.startswith("/usr/"):
            continue  # toolchain headers: the flag stamp carries the hipcc version
        f = Path(tok)
        if not f.exists():
            return float("inf")
        newest = max(newest, f.stat().st_mtime)
    return newest


# Flags of single sources.  conv_unit_wide.hip places single-issue vector instructions one by one into the gaps behind its
# MFMAs; the SLP vectoriser would re-pack the per-element fp32 operations into v_pk_*_f32, which cost several times their
# issue slot beside an MFMA (MI355X_MICROARCH.md, 'price of one filler beside MFMAs').
PER_FILE_FLAGS = {
    "kernels/conv_unit_wide.hip": ["-fno-slp-vectorize"],
    "kernels/gemm_split_w256.hip": ["-fno-slp-vectorize"],  # the same for its in-loop operand split
    # vq_screen_kernel reduces every MFMA result on the vector unit at once: results in VGPRs, not AGPRs + 16 v_accvgpr_read
    "kernels/fsq.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
}
OPTIONAL_FLAG_FILES = {"kernels/fsq.hip"}  # flags that only affect speed: probed, dropped when hipcc does not know them


_probe_cache = {}


def _flags_supported(hipcc: str, extra) -> bool:
    """Whether this hipcc accepts `extra` (compiling an empty translation unit): optional, performance-only flags are
    dropped on toolchains that do not know them instead of failing the whole build."""
    key = (hipcc, tuple(extra))
    if key not in _probe_cache:
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            src = Path(td) / "probe.hip"
            src.write_text("#include <hip/hip_runtime.h>\n__global__ void probe() {}\n")
            r = subprocess.run([hipcc, f"--offload-arch={ARCH}", *extra, "-c", str(src), "-o", str(Path(td) / "probe.o")],
                               capture_output=True, text=True)
            base = subprocess.run([hipcc, f"--offload-arch={ARCH}", "-c", str(src), "-o", str(Path(td) / "probe0.o")],
                                  capture_output=True, text=True)
        # (if even the plain probe fails, the probe itself is broken: keep the flags and let the real compile speak)
        _probe_cache[key] = r.returncode == 0 or base.returncode != 0
    return _probe_cache[key]


def _per_file_flags(hipcc: str, src: str):
    extra = PER_FILE_FLAGS.get(src, [])
    if src in OPTIONAL_FLAG_FILES and extra and not _flags_supported(hipcc, extra):
        print(f"[build] hipcc does not accept {' '.join(extra)}: {src} is built without it (slower vq_screen_kernel)", file=sys.stderr)
        return []
    return extra


def _flag_stamp(hipcc: str, flags) -> str:
    """Identity of everything that shapes an object besides its sources: the flag list (L3AC_EXTRA_HIPCC_FLAGS included)
    and the compiler version.  A change forces a full rebuild."""
    import hashlib
    ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
    per_file = "\n".join(f"{k}: {' '.join(_per_file_flags(hipcc, k))}" for k in sorted(PER_FILE_FLAGS))  # as probed
    return hashlib.sha256(("\n".join(flags) + "\n" + per_file + "\n" + ver).encode()).hexdigest()


def build_library(force: bool = False, verbose: bool = False) -> Path:
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    hipcc = _hipcc()
    flags = [*os.environ.get("L3AC_EXTRA_HIPCC_FLAGS", "").split(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-Wall", "-Wno-unused-function",
             f"-I{REPO / 'include'}", f"-I{CSRC}"]
    hdr_time = _newest_header()
    stamp_file = OBJ_DIR / "flags.sha256"
    stamp = _flag_stamp(hipcc, flags)
    if not stamp_file.exists() or stamp_file.read_text().strip() != stamp:
        force = True  # other flags or another compiler: every object is stale
    jobs = []
    objs = []
    for src in SOURCES:
        s = CSRC / src
        o = OBJ_DIR / (src.replace("/", "_") + ".o")
        d = o.with_suffix(".d")
        objs.append(o)
        stale = force or not o.exists()
        if not stale:
            newest = max(s.stat().st_mtime, _depfile_newest(d) if d.exists() else hdr_time)
            stale = o.stat().st_mtime < newest
        if stale:
            jobs.append([hipcc, *flags, *_per_file_flags(hipcc, src), "-MD", "-MF", str(d), "-c", str(s), "-o", str(o)])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(4, os.cpu_count() or 1)) as pool:
        list(pool.map(run, jobs))
    stamp_file.write_text(stamp + "\n")
    linked = bool(jobs) or not LIB_PATH.exists()
    if linked:
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB_PATH), *map(str, objs)])
    global last_build_report
    last_build_report = {"objects": len(objs), "compiled": len(jobs), "reused": len(objs) - len(jobs), "linked": linked,
                         "forced": bool(force)}
    return LIB_PATH


last_build_report = None  # what the last build_library() call did: {"objects", "compiled", "reused", "linked", "forced"}


if __name__ == "__main__":
    path = build_library(force="--force" in sys.argv, verbose=True)
    r = last_build_report
    print(f"built {path} ({path.stat().st_size / 1e6:.1f} MB): {r['compiled']} of {r['objects']} objects compiled, {r['reused']} reused")
