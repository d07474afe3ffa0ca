"""Build + ctypes binding of libmdm_hip.so (the C ABI declared in include/mdm_hip.h).

The ctypes signatures are generated from the header itself, so the header is the
single source of truth for the boundary.  There is deliberately no fallback: if
the shared library is missing or a call fails, we raise.
"""
import ctypes
import os
import re
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
HEADER = os.path.join(_ROOT, "include", "mdm_hip.h")            # the drop-in boundary
DEV_HEADER = os.path.join(_ROOT, "include", "mdm_hip_dev.h")    # profiling aids (bench.py, tools/)
# MDM_HIP_LIB: load another build of the library (development: in-call A/B of two kernel versions on one GPU box)
LIB_PATH = os.environ.get("MDM_HIP_LIB") or os.path.join(_HERE, "libmdm_hip.so")
SOURCES = ["gemm_conv.hip", "norm.hip", "attention.hip", "elementwise.hip", "optim.hip", "diffusion_ops.hip"]

ABI_VERSION = int(re.search(r"#define\s+MDM_HIP_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))

_lock = threading.Lock()
_lib = None


class MdmHipError(RuntimeError):
    pass


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every HIP source for gfx950 into ml-mdm_amd/mdm_hip/libmdm_hip.so (in-tree)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in ("common.hpp", "conv_args.hpp", "attn32.hpp")]
    deps = srcs + [HEADER] + hdrs
    if not force and os.path.exists(LIB_PATH):
        if all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
            # the note is renewed only when something it records can have changed since it was written (a commit: .git/HEAD
            # or the index moved) -- not on every import: hashing every source and spawning `git describe` in each of N
            # ranks' imports was pure start-up cost
            note = LIB_PATH + ".build.json"
            marks = [os.path.join(_ROOT, ".git", "HEAD"), os.path.join(_ROOT, ".git", "index")]
            try:
                stale = any(os.path.exists(m) and os.path.getmtime(m) > os.path.getmtime(note) for m in marks)
            except OSError:
                stale = os.path.isdir(os.path.join(_ROOT, ".git"))
            if stale:
                _write_build_note(deps, refresh=True)
            return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if (not force) and os.path.exists(o) and all(os.path.getmtime(o) >= os.path.getmtime(d) for d in [s, HEADER] + hdrs):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", s, "-o", o]
        if verbose:
            print("[mdm_hip] " + " ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise MdmHipError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode(errors="replace")))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print("[mdm_hip] " + " ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise MdmHipError("link failed:\n" + r.stdout.decode(errors="replace"))
    _write_build_note(deps)
    return LIB_PATH


def _write_build_note(deps, refresh=False):
    """<lib>.build.json next to the library: what it was built from (git revision + a hash of every source it depends on).
    The .so is git-ignored but ships to the GPU box, where there is no .git: provenance() reads this note there.
    ``refresh``: the library is up to date -- only the revision is renewed, and only where there IS a repository and the
    sources still hash to what the note says (a library built from a dirty tree that was committed afterwards)."""
    import hashlib
    import json
    h = hashlib.sha256()
    for d in sorted(deps):
        h.update(os.path.basename(d).encode())
        h.update(open(d, "rb").read())
    def git(*a):
        try:
            return subprocess.run(["git", "-C", _ROOT] + list(a), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode().strip()
        except OSError:
            return ""
    note = {"git_describe": git("describe", "--always", "--dirty", "--abbrev=12"), "sources_sha256": h.hexdigest()}
    if refresh:
        try:
            old = json.load(open(LIB_PATH + ".build.json"))
        except (OSError, ValueError):
            return
        if not note["git_describe"] or old.get("sources_sha256") != note["sources_sha256"] or old.get("git_describe") == note["git_describe"]:
            return
    try:   # through a temporary file: several ranks may import at once, and a reader must never see half a note
        tmp = "%s.build.json.%d.tmp" % (LIB_PATH, os.getpid())
        with open(tmp, "w") as f:
            json.dump(note, f)
        os.replace(tmp, LIB_PATH + ".build.json")
    except OSError:
        pass


def provenance():
    """{"lib", "sha256" (of the loaded .so), "git_describe", "sources_sha256" (from the build note, if present)} -- printed by
    bench.py and __graft_entry__.smoke() so that a GPUTEST / BENCH record names the binary it ran"""
    import hashlib
    import json
    out = {"lib": os.path.relpath(LIB_PATH, _ROOT), "sha256": None, "git_describe": None, "sources_sha256": None}
    try:
        out["sha256"] = hashlib.sha256(open(LIB_PATH, "rb").read()).hexdigest()
        note = json.load(open(LIB_PATH + ".build.json"))
        out["git_describe"], out["sources_sha256"] = note.get("git_describe"), note.get("sources_sha256")
    except (OSError, ValueError):
        pass
    return out


_CTYPES = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "size_t": ctypes.c_size_t,
    "unsigned long long": ctypes.c_ulonglong,
}


def _ctype_of(decl: str):
    decl = decl.strip()
    if "*" in decl:
        base = decl.replace("const", "").split("*")[0].strip()
        if base == "char":
            return ctypes.c_char_p
        return ctypes.c_void_p
    words = decl.replace("const", "").split()[:-1]   # drop the parameter name
    return _CTYPES[" ".join(words)]


def header_prototypes(path: str = HEADER):
    """[(name, restype, [argtypes], [argnames])] for every function declared in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//.*", "", text)
    protos = []
    for m in re.finditer(r"^\s*([A-Za-z_][\w\s\*]*?)\b(mdm_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.M | re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                argnames.append(re.split(r"[\s\*]+", a)[-1])
                argtypes.append(_ctype_of(a))
        restype = ctypes.c_char_p if "char" in ret else ctypes.c_int
        protos.append((name, restype, argtypes, argnames))
    return protos


def lib():
    """The loaded library; raises MdmHipError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise MdmHipError(
                "libmdm_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "-- there is no CPU fallback for the product path." % LIB_PATH
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, restype, argtypes, _ in header_prototypes() + header_prototypes(DEV_HEADER):
            fn = getattr(handle, name)  # AttributeError if the header and the library drift
            fn.restype = restype
            fn.argtypes = argtypes
        if handle.mdm_abi_version() != ABI_VERSION:
            raise MdmHipError("libmdm_hip.so was built for ABI %d, include/mdm_hip.h declares %d: rebuild (__graft_entry__.build())"
                              % (handle.mdm_abi_version(), ABI_VERSION))
        _apply_dev_env(handle)
        _lib = handle
    return _lib


def _apply_dev_env(handle):
    """Development A/B switches: environment variables of THIS Python layer, applied once through the setters of
    include/mdm_hip_dev.h -- the C entry points themselves never look at the environment."""
    mode = os.environ.get("MDM_HIP_ATTN_BWD")
    if mode:
        handle.mdm_dev_set_attn_bwd({"split": 1, "small": 2, "small16": 3, "stream32": 4, "long16": 5}.get(mode, 0))
    if os.environ.get("MDM_HIP_SKIP_WGRAD_REDUCE") == "1":      # timing-only ablation, wrong gradients
        handle.mdm_dev_set_knob(13, 1)
    if os.environ.get("MDM_HIP_GN_CHUNK_MB"):
        handle.mdm_dev_set_gn_chunk_mb(int(os.environ["MDM_HIP_GN_CHUNK_MB"]))
    if os.environ.get("MDM_HIP_ONE_TILE_BLOCKS"):
        handle.mdm_dev_set_knob(5, int(os.environ["MDM_HIP_ONE_TILE_BLOCKS"]))
    if os.environ.get("MDM_HIP_SPLIT_FILL"):
        handle.mdm_dev_set_knob(6, int(os.environ["MDM_HIP_SPLIT_FILL"]))
    if os.environ.get("MDM_HIP_CONV_DIRECT") == "0":
        handle.mdm_dev_set_knob(7, 1)
    if os.environ.get("MDM_HIP_SPLIT_MINKT"):
        handle.mdm_dev_set_knob(9, int(os.environ["MDM_HIP_SPLIT_MINKT"]))
    if os.environ.get("MDM_HIP_SPLIT_MINSAVE"):
        handle.mdm_dev_set_knob(10, int(os.environ["MDM_HIP_SPLIT_MINSAVE"]))
    if os.environ.get("MDM_HIP_SPLIT_PER_CU"):
        handle.mdm_dev_set_knob(12, int(os.environ["MDM_HIP_SPLIT_PER_CU"]))
    if os.environ.get("MDM_HIP_DEEP_PIPE") == "0":   # no 4-stage GEMM instantiations for under-filled grids
        handle.mdm_dev_set_knob(11, 1)
    if os.environ.get("MDM_HIP_WGRAD_DIRECT") == "0":   # narrow weight gradients by the split GEMM
        handle.mdm_dev_set_knob(8, 1)


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().mdm_last_error()
        raise MdmHipError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))
