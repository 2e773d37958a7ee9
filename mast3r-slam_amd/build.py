"""Build libm3slam_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python mast3r-slam_amd/build.py [--force]

One object per .hip source (compiled in parallel), linked into
mast3r-slam_amd/lib/libm3slam_hip.so.  matching.hip is compiled with
-ffp-contract=off: its results are bit-exact against the CPU oracle.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib")
LIB = os.path.join(OUT, "libm3slam_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fhip-fp32-correctly-rounded-divide-sqrt",
          "-Wall", "-Wno-unused-function"]
PER_FILE = {
    "matching.hip": ["-ffp-contract=off"],
    "attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    # MFMA results are consumed by VALU (arg-max) right away: keep them out of the AGPR file (4 v_accvgpr_read per MFMA)
    # -fno-honor-nans: scores are sums of finite products, so the arg-max needs no NaN canonicalisation (a v_max x, x
    # in front of every v_max3 that takes an MFMA result); infinities (the -inf initial maximum) stay honoured
    "fast_nn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-honor-nans"],
    # hipcc 7.2's SLP vectoriser mis-compiled the first k_track_accum (two of its 36 sums wrong at -O3, right with
    # -fno-slp-vectorize or -O1: tools/incident_r01/run.py); packed fp32 math buys these
    # float64-fold-bound kernels nothing, so it stays off for both Gauss-Newton sources
    "tracking.hip": ["-fno-slp-vectorize"],
    "gn_rays.hip": ["-fno-slp-vectorize"],
}


def _sources():
    return sorted(f for f in os.listdir(SRC) if f.endswith(".hip"))


def _digest(paths, extra=()):
    """sha256 over the CONTENT of the inputs and the command line: a checkout, a copy to the GPU box or a
    `touch` changes mtimes without changing what would be compiled (and the reverse after `git stash`)."""
    import hashlib
    h = hashlib.sha256()
    for p in sorted(os.path.realpath(x) for x in paths):
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    for e in extra:
        h.update(str(e).encode() + b"\0")
    return h.hexdigest()


def _stale(target, deps, extra=()):
    """True when `target` is missing or was built from different inputs (digest kept beside it)."""
    stamp = target + ".sha256"
    if not os.path.exists(target) or not os.path.exists(stamp):
        return True
    return open(stamp).read().strip() != _digest(deps, extra)


def _mark(target, deps, extra=()):
    with open(target + ".sha256", "w") as f:
        f.write(_digest(deps, extra))


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OUT, exist_ok=True)
    headers = [os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "m3slam.h"))
    headers += [os.path.join(HERE, "..", "include", f) for f in os.listdir(os.path.join(HERE, "..", "include"))]
    headers = sorted(set(os.path.realpath(h) for h in headers))
    jobs = []
    objs = []
    marks = []
    for s in _sources():
        src = os.path.join(SRC, s)
        obj = os.path.join(OUT, s.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [HIPCC, *COMMON, *PER_FILE.get(s, []), "-c", src, "-o", obj]
        if force or _stale(obj, [src] + headers, cmd):
            jobs.append(cmd)
            marks.append((obj, [src] + headers, cmd))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, flush=True)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    for m in marks:
        _mark(*m)
    link = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if jobs or force or _stale(LIB, objs, link):
        run(link)
        _mark(LIB, objs, link)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
