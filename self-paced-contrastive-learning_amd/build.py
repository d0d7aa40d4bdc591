"""Build libspcl_hip.so (hand-written gfx950 kernels + C ABI) in-tree with hipcc.

    python self-paced-contrastive-learning_amd/build.py [--force]

hipcc cross-compiles for gfx950 without a GPU; the .so is git-ignored but travels with the tree."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libspcl_hip.so")
OBJ = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: left on, hipcc packs adjacent f32 adds / multiplies / FMAs of the epilogues into v_pk_*_f32, which
# issue slower than the two scalar instructions they replace on gfx950 (MI355X_MICROARCH.md, price list); same arithmetic,
# same bits.  Whole step, same box, build A/B: 1147.4 -> 1134.8 us.
ARCH = os.environ.get("SPCL_BUILD_ARCH", "gfx950")  # (experiments: gfx950:xnack-)
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize"]
# experiments / debug builds: SPCL_BUILD_DEFS="-DSPCL_CONV16_STAMPS_BUILD=1 -DSPCL_FAST_WIDE_STORES=0" (use with --force)
FLAGS += os.environ.get("SPCL_BUILD_DEFS", "").split()
# The first kernel-argument dwords arrive in SGPRs at wave launch (gfx950 kernarg preload) instead of through a scalar load:
# the short one-tile workgroups of the convolutions start their first transfers a memory round trip earlier (whole step,
# same box, build A/B: 1159.6 -> 1152.6 us); supcon.hip was tuned with 14 (the large-batch sweeps).
PRELOAD = ["-mllvm", "-amdgpu-kernarg-preload-count=16"]
EXTRA = {"supcon.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=14"]}
if os.environ.get("SPCL_BUILD_NOSLP"):  # experiment: no SLP packing of adjacent f32 operations (v_pk_*_f32) in the named files
    for _f in os.environ["SPCL_BUILD_NOSLP"].split(","):
        EXTRA.setdefault(_f, []).append("-fno-slp-vectorize")


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _flags_for(src):
    """the complete hipcc flag list of one source: FLAGS + kernarg preload (a file's EXTRA entry replaces the preload
    COUNT, never drops the preload) + the file's other extras"""
    if not src.endswith(".hip"):
        return list(FLAGS)
    extra = list(EXTRA.get(src, []))
    if "kernarg-preload" in " ".join(extra + FLAGS):
        return FLAGS + extra
    return FLAGS + PRELOAD + extra


def _stamp(src):
    """what an object was compiled WITH: an object whose flags differ from the current ones is stale whatever its mtime
    (SPCL_BUILD_DEFS / SPCL_BUILD_ARCH / SPCL_BUILD_NOSLP experiments would otherwise be 'reused' by the next plain build)"""
    return hashlib.sha256(" ".join([HIPCC] + _flags_for(src)).encode()).hexdigest()


def _stamp_path(obj):
    return obj + ".flags"


def _stamp_matches(obj, src):
    try:
        return open(_stamp_path(obj)).read().strip() == _stamp(src)
    except OSError:
        return False


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(HERE, "..", "include", "spcl_hip.h"))
    jobs, objs = [], []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.rsplit(".", 1)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s, os.path.abspath(__file__)] + headers) or not _stamp_matches(o, src):
            cmd = [HIPCC] + _flags_for(src) + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", s, "-o", o]
            jobs.append((src, cmd, o))

    def run(job):
        src, cmd, o = job
        if os.path.exists(_stamp_path(o)):
            os.remove(_stamp_path(o))  # an interrupted compile must not leave an object that looks current
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode == 0:
            with open(_stamp_path(o), "w") as fh:
                fh.write(_stamp(src) + "\n")
        return src, r.returncode, r.stdout + r.stderr

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        for src, rc, out in ex.map(run, jobs):
            if verbose and out.strip():
                print(out, file=sys.stderr)
            if rc != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n{out}")
            if verbose:
                print(f"[spcl build] compiled {src}")
    linked = False
    link_stamp = hashlib.sha256(" ".join(_stamp(src) for src in _sources()).encode()).hexdigest()
    try:
        lib_current = open(LIB + ".flags").read().strip() == link_stamp
    except OSError:
        lib_current = False
    if jobs or force or _stale(LIB, objs) or not lib_current:
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
        linked = True
        with open(LIB + ".flags", "w") as fh:
            fh.write(link_stamp + "\n")
        if verbose:
            print(f"[spcl build] linked {LIB}")
    if verbose:  # a reader of the log can tell a rebuild from a reuse of objects that travelled with the tree
        print(f"[spcl build] compiled {len(jobs)} / reused {len(objs) - len(jobs)} of {len(objs)} objects; "
              f"{'linked' if linked else 'reused'} {os.path.basename(LIB)} (--force rebuilds everything)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
