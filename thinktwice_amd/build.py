"""In-tree build of libthinktwice_hip.so (hipcc, --offload-arch=gfx950).

The library is the product's only compute path; nothing here falls back to a
CPU implementation.  `python -m thinktwice_amd.build` rebuilds it; objects are
cached per source by mtime so incremental builds take seconds.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libthinktwice_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=off",
         "-Wno-unused-result"]


# per-source extra flags.  conv_x3_pipe.hip: its K loop is a hand-placed instruction stream; the SLP vectoriser would fuse the
# operand split's f32 subtractions into v_pk_add_f32, which is slower than two plain VALU beside MFMAs (MI355X_MICROARCH.md)
EXTRA_FLAGS = {"conv_x3_pipe.hip": ["-fno-slp-vectorize"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hdrs.append(os.path.join(HERE, "..", "include", "thinktwice_hip.h"))
    hdrs.append(os.path.abspath(__file__))  # flag changes rebuild everything
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, hdr_mtime, verbose):
    obj = os.path.join(OBJ, src + ".o")
    spath = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(spath), hdr_mtime):
        return obj
    cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", spath, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return obj


def source_fingerprint():
    """sha256 over the library's sources (csrc/*, include/thinktwice_hip.h): the identity of a build, available wherever the
    tree is (the GPU box has no .git).  Profiles under profiles/ carry it (tools/summarize_pmc.py) and bench.py only quotes
    counter data whose fingerprint matches the tree it runs from."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".cpp", ".h"))]
    files.append(os.path.join(HERE, "..", "include", "thinktwice_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def build(verbose=True, force=False):
    os.makedirs(OBJ, exist_ok=True)
    # csrc/plan_thunks.inc: one thunk per stream-taking entry of the header (the plan runtime's dispatch table)
    sys.path.insert(0, os.path.join(HERE, "..", "tools"))
    import gen_plan_thunks
    gen_plan_thunks.main()
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    hdr_mtime = _deps_mtime()
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, hdr_mtime, verbose), srcs))
    if (not os.path.exists(LIB)) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    # tools/plan_host: the Python-free host of the forward (loads a plan file + weights, runs tt_encoder_fwd / tt_decoder_fwd)
    host_src = os.path.join(HERE, "..", "tools", "plan_host.cpp")
    host_bin = os.path.join(HERE, "..", "tools", "plan_host")
    if os.path.exists(host_src) and (not os.path.exists(host_bin) or
                                     os.path.getmtime(host_bin) < max(os.path.getmtime(host_src), os.path.getmtime(LIB))):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-Wno-unused-result", "-I", os.path.join(HERE, "..", "include"),
               host_src, "-L", HERE, "-lthinktwice_hip", "-Wl,-rpath,$ORIGIN/../thinktwice_amd", "-o", host_bin]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(verbose=True, force="--force" in sys.argv)
    print(LIB)
