"""Compile the HIP kernels + C ABI into hid-vae_amd/libhidvae_hip.so (in-tree, gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the build container; the .so is git-ignored
but travels to the GPU box with the repo snapshot.

Staleness is decided by CONTENT, not by mtime (a snapshot copy or a checkout resets mtimes in either direction): every object
carries the sha256 of (its source, every header, the flags) beside it, and the library the sha256 of (all of those, its own
bytes).  build() recompiles exactly what no longer matches -- on an unchanged tree nothing, and the recorded digests are the proof
that the shipped binary is the one these sources produce."""
import glob
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libhidvae_hip.so")
OBJ = os.path.join(HERE, "build")
STAMP = os.path.join(OBJ, "digests.json")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wno-unused-value", "-Wno-pass-failed"]


def _sha(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _file_sha(path):
    return _sha([path]) if os.path.exists(path) else None


def _load():
    try:
        with open(STAMP) as f:
            return json.load(f)
    except Exception:  # noqa: BLE001  missing or unreadable stamp: everything is stale
        return {}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "hidvae.h")]


def source_digest():
    """sha256 over every source, header and the flags: what the shipped library must have been built from"""
    return _sha(sources() + headers(), " ".join(FLAGS))


def is_current():
    st = _load()
    return st.get("sources") == source_digest() and st.get("library") is not None and st.get("library") == _file_sha(OUT)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    st = {} if force else _load()
    hdrs = headers()
    flags = " ".join(FLAGS)
    objs, procs, new_objs = [], [], {}
    for s in sources():
        o = os.path.join(OBJ, os.path.basename(s) + ".o")
        objs.append(o)
        want = _sha([s] + hdrs, flags)
        new_objs[os.path.basename(o)] = want
        have = (st.get("objects") or {}).get(os.path.basename(o))
        if force or have != want or not os.path.exists(o):
            cmd = ["hipcc"] + FLAGS + (["-x", "hip"] if s.endswith(".hip") else []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    src = source_digest()
    if force or procs or st.get("sources") != src or st.get("library") is None or st.get("library") != _file_sha(OUT):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        json.dump({"sources": src, "objects": new_objs, "library": _file_sha(OUT), "flags": FLAGS}, f, indent=1)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
