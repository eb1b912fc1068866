"""CPU-side check: the built C-ABI library loads and exports every symbol include/unirestore_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "unirestore_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ur_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from unirestore_amd import build
    lib = ctypes.CDLL(build.build(verbose=False))
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"


def test_ctypes_signatures_cover_header():
    from unirestore_amd import capi
    assert sorted(capi.SIGNATURES) == _declared()
    assert capi.lib.ur_version() >= 100
    assert ctypes.sizeof(capi.ConvDesc) % 8 == 0


def test_no_kernel_uses_scratch_memory():
    """Accumulator arrays that end up in scratch (an un-unrolled loop makes their index dynamic) halve a kernel's speed
    without failing any parity test: the build records every kernel's resource usage, this keeps scratch at zero."""
    import json
    import os
    import unirestore_amd.build as b
    b.build()
    path = os.path.join(os.path.dirname(b.LIB), "build", "resource_usage.json")
    if not os.path.exists(path):
        b.build(force=True)
    usage = json.load(open(path))
    assert len(usage) >= 40
    # The token-stationary chain kernels (csrc/tchain.hip) run ONE wave per SIMD on the whole 512-entry register file by design;
    # hipcc parks a handful of lane-constant scalars of the HEAD / TAIL kinds (lane id, address offsets, stored once at kernel
    # entry, reloaded between phases, never inside the FF loop) in scratch.  The allowance is PINNED to the measured bytes per
    # kernel and 16-bit type (hipcc 7.2): a compiler or source change that spills more - or any other kernel touching scratch at
    # all - fails here and has to be looked at (and, if it is the same kind of spill, re-pinned deliberately).
    # (round 5: the fp16 pack lost its two clamps - IEEE overflow instead of saturation - which moved the fp16 kinds' allocation:
    #  HEAD 0 -> 20 bytes, the bf16 kernel's figure, TAIL 68 -> 60; same lane-constant spills, re-pinned)
    pinned = {("tchain_head_kernel", "Lb0E"): 20, ("tchain_head_kernel", "Lb1E"): 20,
              ("tchain_tail_kernel", "Lb0E"): 52, ("tchain_tail_kernel", "Lb1E"): 60}

    def allowed(k):
        for (name, dt), v in pinned.items():
            if name in k and f"{name}I{dt}" in k:
                return v
        return 0
    bad = {k: v["ScratchSize"] for k, v in usage.items() if v.get("ScratchSize", 0) > allowed(k)}
    assert not bad, bad
    assert all(v.get("ScratchSize", 0) == 0 for k, v in usage.items() if "tchain_mlp_kernel" in k)


def test_header_is_plain_c(tmp_path):
    """include/unirestore_hip.h is the boundary a C / cgo / JNI caller would bind: it must compile as C99 on its own."""
    import os
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        import pytest
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "unirestore_hip.h"\nint main(void) { ur_conv_desc d; (void)d; return ur_version() > 0 ? 0 : 1; }\n')
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", os.path.join(root, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_ctypes_struct_layout_matches_the_header(tmp_path):
    """ur_conv_desc / ur_conv_plan are filled from Python through ctypes mirrors: every field offset and the total size must
    equal what a C compiler lays out for the header's structs."""
    import shutil
    import subprocess
    from unirestore_amd import capi
    gcc = shutil.which("gcc")
    if gcc is None:
        import pytest
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fields = [f[0] for f in capi.ConvDesc._fields_]
    lines = "".join(f'  printf("{n} %zu\\n", offsetof(ur_conv_desc, {n}));\n' for n in fields)
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "unirestore_hip.h"\nint main(void) {\n' + lines +
                   '  printf("sizeof %zu\\n", sizeof(ur_conv_desc));\n  printf("plan %zu\\n", sizeof(ur_conv_plan));\n  return 0;\n}\n')
    exe = tmp_path / "layout"
    r = subprocess.run([gcc, "-std=c99", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True).stdout.splitlines())
    for n in fields:
        assert int(out[n]) == getattr(capi.ConvDesc, n).offset, n
    assert int(out["sizeof"]) == ctypes.sizeof(capi.ConvDesc) and int(out["plan"]) == ctypes.sizeof(capi.ConvPlan)


def test_generated_asm_blocks_are_current(tmp_path, monkeypatch):
    """csrc/tchain_asm.inc, csrc/igemm_asm.inc, csrc/attention_pp_asm.inc and the timing-only variants tools/ab/*_abl{4,5}.inc are
    GENERATED (tools/gen_chain_asm.py, tools/gen_igemm_asm.py, tools/gen_attn_asm.py): the committed files must be exactly what
    the generators emit (with no ATTN_* environment switches set)."""
    import importlib.util
    import shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tools = tmp_path / "tools"
    out = tmp_path / "unirestore_amd" / "csrc"
    tools.mkdir(parents=True); out.mkdir(parents=True); (tools / "ab").mkdir()
    for f in ("gen_chain_asm.py", "gen_igemm_asm.py", "gen_attn_asm.py"):
        shutil.copy(os.path.join(root, "tools", f), tools / f)
    import sys
    sys.path.insert(0, str(tools))
    for k in [k for k in os.environ if k.startswith("ATTN_")]:
        monkeypatch.delenv(k)                                # (restored when the test ends)
    argv, sys.argv = sys.argv, ["gen"]                       # (gen_attn_asm.py takes an optional output path)
    try:
        for mod in ("gen_chain_asm", "gen_igemm_asm", "gen_attn_asm"):
            sys.modules.pop(mod, None)
            spec = importlib.util.spec_from_file_location(mod, str(tools / (mod + ".py")))
            m = importlib.util.module_from_spec(spec)
            sys.modules[mod] = m
            spec.loader.exec_module(m)
            m.main()
    finally:
        sys.argv = argv
        sys.path.remove(str(tools))
        for mod in ("gen_chain_asm", "gen_igemm_asm", "gen_attn_asm"):
            sys.modules.pop(mod, None)
    made = sorted(os.listdir(out))
    assert made == ["attention_pp_asm.inc", "igemm_asm.inc", "tchain_asm.inc"]
    for f in made:
        assert (out / f).read_text() == open(os.path.join(root, "unirestore_amd", "csrc", f)).read(), f
    made_ab = sorted(os.listdir(tools / "ab"))
    assert made_ab == ["igemm_asm_abl4.inc", "igemm_asm_abl5.inc", "tchain_asm_abl4.inc", "tchain_asm_abl5.inc"]
    for f in made_ab:
        assert (tools / "ab" / f).read_text() == open(os.path.join(root, "tools", "ab", f)).read(), f
