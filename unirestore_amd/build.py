"""Build libunirestore_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

The .so is written in-tree (unirestore_amd/libunirestore_hip.so) so it travels with a repo snapshot
to the GPU box; it is git-ignored.  Rebuilds only when a source is newer than the library.
"""
import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("UR_LIB_OUT") or os.path.join(HERE, "libunirestore_hip.so")      # UR_LIB_OUT / UR_EXTRA_FLAGS: A/B builds
EXTRA = os.environ.get("UR_EXTRA_FLAGS", "").split()
IGEMM_UNITS = ["igemm_v2.hip", "igemm_halo.hip", "igemm_v1a.hip", "igemm_v1b.hip", "igemm_g1.hip"]     # slowest first
# (source, extra flags, object name): every igemm instantiation unit is built once per 16-bit type
SOURCES = [(u, [f"-DUR_TU_F16={t}"], u.replace(".hip", "_f16.o" if t else "_bf16.o")) for u in IGEMM_UNITS for t in (0, 1)] + \
          [("conv_wstream.hip", [f"-DUR_TU_F16={t}"], "conv_wstream_f16.o" if t else "conv_wstream_bf16.o") for t in (0, 1)] + \
          [("attention.hip", [f"-DUR_TU_F16={t}"], "attention_f16.o" if t else "attention_bf16.o") for t in (0, 1)] + \
          [("attention_pp.hip", [f"-DUR_TU_F16={t}"], "attention_pp_f16.o" if t else "attention_pp_bf16.o") for t in (0, 1)] + \
          [("attention512.hip", [f"-DUR_TU_F16={t}"], "attention512_f16.o" if t else "attention512_bf16.o") for t in (0, 1)] + \
          [(u, [], u.replace(".hip", ".o")) for u in ("tchain.hip", "igemm.hip", "norms.hip", "elementwise.hip", "runtime.hip")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
         "-Wno-unused-result"]


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.isabs(c) and os.path.exists(c):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "unirestore_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objdir = os.path.join(HERE, "build" if not EXTRA else "build_" + "_".join(f.strip("-").replace("=", "") for f in EXTRA))
    os.makedirs(objdir, exist_ok=True)
    cc = _hipcc()

    def one(unit):
        src, extra, oname = unit
        obj = os.path.join(objdir, oname)
        cmd = [cc, *FLAGS, *EXTRA, *extra, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", obj]
        deps = [os.path.join(CSRC, src), os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "unirestore_hip.h")]
        if src.startswith("igemm") or src.startswith("conv_wstream"):
            deps += [os.path.join(CSRC, "igemm_impl.h"), os.path.join(CSRC, "igemm_asm.inc")]
        if src.startswith("attention"):
            deps += [os.path.join(CSRC, "attention_params.h"), os.path.join(CSRC, "attention_pp_asm.inc")]
        if src.startswith("tchain"):
            deps.append(os.path.join(CSRC, "tchain_asm.inc"))
        newest = max(os.path.getmtime(d) for d in deps)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > newest and os.path.exists(obj + ".usage"):
            usage.update(json.load(open(obj + ".usage")))          # unchanged since this object was built
            return obj
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        # per-kernel registers / scratch (a kernel whose accumulators land in scratch memory is 2x slower: keep it visible)
        name = None
        for line in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
            m = re.search(r"(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]): (\d+)", line)
            if m and name:
                usage.setdefault(f"{oname}:{name}", {})[m.group(1).split(" ")[0]] = int(m.group(2))
        json.dump({k: v for k, v in usage.items() if k.startswith(oname + ":")}, open(obj + ".usage", "w"))
        return obj

    usage = {}
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 8)) as ex:
        objs = list(ex.map(one, SOURCES))
    with open(os.path.join(objdir, "resource_usage.json"), "w") as f:
        json.dump(usage, f, indent=0, sort_keys=True)
    spills = {k: v["ScratchSize"] for k, v in usage.items() if v.get("ScratchSize", 0) > 0}
    if spills and verbose:
        print(f"WARNING: kernels using scratch memory: {spills}", file=sys.stderr)
    r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
