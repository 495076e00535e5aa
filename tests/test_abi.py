"""C-ABI surface tests that need no GPU: the HIP library builds for gfx950, loads, and exports every symbol
include/wn_abi.h declares (no compute calls).  Argument validation that happens before the first HIP call is exercised on
the PRODUCT library itself (no GPU needed); error codes that need a live handle are covered on the GPU
(tests/test_gpu_parity.py::test_abi_error_codes_on_a_live_handle); engine.py's own argument checks run on the test double."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from double_lib import double_backend, double_library
from mi355_wavenet import _abi, engine, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "wn_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wn_[a-z_]+)\s*\(", src)))


def test_header_functions_match_binding():
    assert declared_functions() == sorted(_abi.EXPORTS)


def test_hip_library_builds_and_exports_every_symbol():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "pytorch-wavenet_amd"))
    import build
    so = build.build_hip()
    out = subprocess.check_output(["nm", "-D", "--defined-only", so]).decode()
    exported = set(l.split()[-1] for l in out.splitlines() if l.strip())
    for fn in declared_functions():
        assert fn in exported, "%s not exported by %s" % (fn, so)
    # it must contain a gfx950 code object and nothing else
    bundle = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o", "--input=" + so],
                            capture_output=True, text=True)
    if bundle.returncode == 0 and bundle.stdout.strip():
        targets = [t for t in bundle.stdout.split() if "amdgcn" in t]
        assert targets and all("gfx950" in t for t in targets), targets
    lib = _abi.Library(so)  # dlopen + prototype check; no device call
    assert lib.dll.wn_abi_version() == _abi.ABI_VERSION == 5


def test_struct_sizes_match_header():
    assert ctypes.sizeof(_abi.wn_config) == 16 * 4
    assert ctypes.sizeof(_abi.wn_weight_ptrs) == 14 * 8
    assert ctypes.sizeof(_abi.wn_generate_args) == 8 + 8 + 8 + 4 + 4 + 8 * 5 + 4 + 4 + 8
    assert ctypes.sizeof(_abi.wn_info) == 8 * 4 + 4 * 8 + 14 * 4
    assert ctypes.sizeof(_abi.wn_train_layout) == 14 * 8


def _cfg(**kw):
    base = dict(synth.CONFIGS["tiny"])
    base.update(kw)
    return base


def test_error_codes_never_exceptions():
    """The product library validates its arguments before it touches the HIP runtime: these calls return codes (never
    throw, never crash) in a container without a GPU."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "pytorch-wavenet_amd"))
    import build
    d = _abi.Library(build.build_hip()).dll
    h = ctypes.c_void_p()
    assert d.wn_create(None, ctypes.byref(h)) == _abi.WN_E_BADARG
    bad = _abi.wn_config(0, 2, 16, 16, 32, 32, 256, 2, 0, 1, 0, 0, 0)
    assert d.wn_create(ctypes.byref(bad), ctypes.byref(h)) == _abi.WN_E_BADARG
    assert b"non-positive" in d.wn_last_error()
    assert d.wn_create(ctypes.byref(_abi.wn_config(3, 2, 16, 16, 32, 32, 256, 0, 0, 1, 0, 0, 0)), ctypes.byref(h)) == _abi.WN_E_BADARG
    assert d.wn_create(ctypes.byref(_abi.wn_config(30, 2, 16, 16, 32, 32, 256, 2, 0, 1, 0, 0, 0)), ctypes.byref(h)) == _abi.WN_E_UNSUPPORTED
    assert d.wn_create(ctypes.byref(_abi.wn_config(3, 2, 16, 16, 32, 32, 256, 2, 0, 1, 0, -1, 0)), ctypes.byref(h)) == _abi.WN_E_BADARG
    assert not h.value
    args = _abi.wn_generate_args()
    assert d.wn_generate(None, ctypes.byref(args)) == _abi.WN_E_BADARG
    assert d.wn_load_weights(None, None) == _abi.WN_E_BADARG
    assert d.wn_reset(None, None) == _abi.WN_E_BADARG
    assert d.wn_wait(None) == _abi.WN_E_BADARG
    assert d.wn_export_queue(None, 0, 0, None, None, None) == _abi.WN_E_BADARG
    assert d.wn_get_info(None, None) == _abi.WN_E_BADARG
    assert d.wn_prime(None, None, 1, 1, None) == _abi.WN_E_BADARG
    assert d.wn_forward(None, None, 1, 1, 1, None, None) == _abi.WN_E_BADARG
    d.wn_destroy(None)  # harmless


def test_double_mirrors_the_state_codes():
    """The host-memory test double answers the call-order errors the way the product does (checked on the GPU in
    tests/test_gpu_parity.py::test_abi_error_codes_on_a_live_handle), so host-logic tests see the same contract."""
    d = double_library().dll
    h = ctypes.c_void_p()
    ok = _abi.wn_config(3, 2, 16, 16, 32, 32, 256, 2, 0, 1, 0, 0, 0)
    assert d.wn_create(ctypes.byref(ok), ctypes.byref(h)) == 0
    args = _abi.wn_generate_args()
    assert d.wn_generate(h, ctypes.byref(args)) == _abi.WN_E_STATE  # no weights yet
    assert d.wn_load_weights(h, None) == _abi.WN_E_BADARG
    w = _abi.wn_weight_ptrs()
    assert d.wn_load_weights(h, ctypes.byref(w)) == _abi.WN_E_BADARG
    assert d.wn_export_queue(h, 99, 0, None, None, None) == _abi.WN_E_BADARG
    d.wn_destroy(h)


def test_engine_argument_validation():
    cfg = _cfg()
    W = synth.init_weights(cfg, seed=1)
    eng = engine.Engine(cfg, W, n_streams=2, **double_backend())
    with pytest.raises(ValueError):
        eng.generate(4, np.array([[1, 2, 300], [1, 2, 3]]))
    with pytest.raises(ValueError):
        eng.generate(4, np.zeros((3, 2), dtype=np.int64))
    with pytest.raises(_abi.WnError):
        engine.Engine(_cfg(kernel_size=0), W, **double_backend())


def test_product_library_is_the_only_default(monkeypatch):
    """The package must fail loudly -- not fall back -- when the HIP library is absent."""
    monkeypatch.setattr(_abi, "PRODUCT_LIB", "/nonexistent/libwn_mi355.so")
    monkeypatch.setattr(_abi, "_product", None)
    with pytest.raises(RuntimeError, match="no CPU/torch fallback"):
        _abi.load_product_library()


def test_header_is_plain_c(tmp_path):
    """include/wn_abi.h is the boundary for non-Python callers: it must compile as plain C (no C++, no HIP, no torch types)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "use_abi.c"
    src.write_text('#include "wn_abi.h"\n'
                   'int probe(void) { wn_config c; wn_generate_args a; wn_info i; wn_train_layout t; (void)c; (void)a; (void)i; (void)t;\n'
                   '                  return (int)sizeof(wn_weight_ptrs) + WN_ABI_VERSION + WN_E_STATE; }\n')
    inc = os.path.join(ROOT, "include")
    for std in ("c99", "c11"):
        subprocess.check_call([gcc, "-std=" + std, "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-I", inc, str(src)])


def test_reserved_poll_registers_belong_to_the_hand_scheduled_blocks_only(tmp_path):
    """csrc/wn_kernel_v3.h keeps request sets and the tap FIFO in v[152:167] across its inline-assembly blocks (loads into them may
    still be in flight when a block ends).  The kernels carry amdgpu_num_vgpr, so the compiler cannot allocate those registers;
    build.py verifies that on the disassembly of every library it builds and refuses to install one that breaks it.  Here: the
    installed library passes the check, and the check does reject the patterns it is there for."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "pytorch-wavenet_amd"))
    import build
    assert os.path.exists(build.objdump_path()), "llvm-objdump is part of the ROCm image: the register check must not be skipped"
    before = set(os.listdir(os.path.dirname(build.OUT)))
    assert build.check_hand_scheduled_registers(build.build_hip()) >= 2   # one kernel per form (and shape)
    assert set(os.listdir(os.path.dirname(build.OUT))) == before, "the check extracts the code object into a temporary directory only"
    cwd_before = set(os.listdir("."))
    assert build.check_hand_scheduled_registers(os.path.relpath(build.OUT)) >= 2   # (a relative path, from whatever the cwd is)
    assert set(os.listdir(".")) == cwd_before

    fake = tmp_path / "objdump.py"   # a stand-in disassembler: feeds the checker hand-written listings

    def run(body):
        listing = "0000000000001000 <_Z22wn_generate_kernel_v3mILi128EEv6WnPlan5WnRun>:\n" + "".join(
            ("0000000000002000 %s:\n" % ln) if ln.startswith("<") else ("\t%s   // 0: 0\n" % ln) for ln in body)   # ("<L3>": a branch-target label)
        (tmp_path / "listing.txt").write_text(listing)
        fake.write_text("#!/usr/bin/env python3\nimport sys, os\n"
                        "if '--offloading' in sys.argv: open('x.gfx950', 'w').close()\n"
                        "else: sys.stdout.write(open(%r).read())\n" % str(tmp_path / "listing.txt"))
        fake.chmod(0o755)
        lib = tmp_path / "dummy.so"
        lib.write_bytes(b"")
        return build.check_hand_scheduled_registers(str(lib), objdump=str(fake))

    good = ["global_load_dwordx2 v[152:153], v1, s[2:3] sc1", "v_cmp_eq_u32_e32 vcc, s5, v153", "v_add_f32_e32 v3, v3, v152",
            "global_load_dword v154, v[4:5], off sc1", "s_waitcnt vmcnt(11)", "v_mov_b32_e32 v7, v154", "v_cmp_eq_u32_e64 s[8:9], s5, v155"]
    assert run(good) == 1
    for bad in ("v_mov_b64_e32 v[156:157], s[18:19]",            # the compiler parking a value there
                "v_add_f32_e32 v152, v3, v4",                     # a reserved register as a destination
                "v_mov_b32_e32 v160, v3",
                "v_mov_b32_e32 v9, v153",                         # a FIFO take that is not directly behind one of the FIFO's waits
                "v_cmp_eq_u32_e32 vcc, v153, v9",                 # ... in the wrong operand position
                "global_load_dwordx2 v[10:11], v[152:153], off",  # ... as an address
                "global_load_dword v158, v[4:5], off",            # FIFO entries are v152-v157
                "v_mov_b32_e32 v3, v168",                         # beyond what a 768-thread workgroup leaves a lane
                "scratch_load_dword v3, off, off"):               # a spill
        with pytest.raises(RuntimeError):
            run(good + [bad])
    # rule 4: the SGPR base of a hand-scheduled load must not come out of a VALU instruction (v_readlane_b32 of a spilled pointer) less than
    # five wait states earlier -- the hazard recognizer does not look into inline assembly (round 4: memory access faults in the two-slice form)
    load = "global_load_dwordx2 v[152:153], v1, s[12:13] sc1"
    for pre in (["v_readlane_b32 s12, v128, 20", "v_readlane_b32 s13, v128, 21"],                    # 0 and 1 wait states
                ["v_readlane_b32 s13, v128, 21", "s_add_u32 s12, s12, s5", "s_addc_u32 s6, s6, 0"],     # 2 (only s12 re-written by the SALU)
                ["v_readlane_b32 s12, v128, 20", "s_nop 2"],                                         # 3
                ["v_cmp_eq_u32_e64 s[12:13], s5, v3", "s_nop 3"]):                                    # 4, any VALU write of the pair
        with pytest.raises(RuntimeError, match="SGPR operand of this memory instruction"):
            run(good + pre + [load])
    for pre in (["v_readlane_b32 s12, v128, 20", "v_readlane_b32 s13, v128, 21", "s_nop 4"],         # five wait states
                ["v_readlane_b32 s13, v128, 21", "s_nop 1", "s_mov_b32 s4, 1", "v_mov_b32_e32 v3, 0", "s_nop 0"],
                ["v_readlane_b32 s10, v128, 20", "v_readlane_b32 s11, v128, 21"],                  # another pair
                ["v_readlane_b32 s12, v128, 20", "v_readlane_b32 s13, v128, 21", "s_add_u32 s12, s12, s4", "s_addc_u32 s13, s13, 0"]):   # last written by the SALU: no hazard
        assert run(good + pre + [load]) == 1
    # ... on EVERY path to the load: the walk follows the branches that target a label in front of it
    rl = ["v_readlane_b32 s12, v128, 20", "v_readlane_b32 s13, v128, 21"]
    with pytest.raises(RuntimeError, match="SGPR operand of this memory instruction"):   # the fall-through is long enough, the taken branch is not
        run(good + rl + ["s_cbranch_vccnz L7", "s_nop 4", "v_mov_b32_e32 v3, 0", "<L7>", load])
    with pytest.raises(RuntimeError, match="SGPR operand of this memory instruction"):   # two hops
        run(good + rl + ["s_cbranch_scc1 L5", "s_nop 4", "<L5>", "s_cbranch_vccz L6", "s_nop 4", "<L6>", load])
    assert run(good + rl + ["s_nop 4", "s_cbranch_vccnz L7", "v_mov_b32_e32 v3, 0", "<L7>", load]) == 1          # five wait states in front of the branch
    assert run(good + rl + ["s_branch L9", "<L8>", load, "<L9>", "s_nop 0"]) == 1                                 # no fall-through behind s_branch, nobody jumps to L8
    with pytest.raises(RuntimeError, match="SGPR operand of this memory instruction"):   # a loop's back edge
        run(good + ["s_nop 4", "<L2>", load] + rl + ["s_cbranch_scc1 L2"])
    # ... and, round 5, for EVERY memory instruction of EVERY kernel (not only the poll sets of the wave-specialised ones): whatever an inline-assembly
    # block emits anywhere in the library is held to the rule, the compiler's own instructions pass it by construction
    other = "<_Z21wn_generate_kernel_v4ILi64EEv6WnPlan5WnRun>"
    with pytest.raises(RuntimeError, match="wn_generate_kernel_v4.*SGPR operand of this memory instruction"):
        run(good + [other, "v_readfirstlane_b32 s8, v1", "s_nop 1", "buffer_load_dwordx4 v[0:3], v4, s[8:11], 0 offen sc1"])
    with pytest.raises(RuntimeError, match="SGPR operand of this memory instruction"):          # the offset register of a buffer store
        run(good + [other, "v_readfirstlane_b32 s20, v1", "buffer_store_dwordx2 v[0:1], v4, s[8:11], s20 offen"])
    with pytest.raises(RuntimeError, match="SGPR operand of this memory instruction"):          # a plain store's base pair
        run(good + [other, "v_readlane_b32 s3, v9, 1", "s_nop 3", "global_store_dword v1, v2, s[2:3]"])
    assert run(good + [other, "v_readfirstlane_b32 s8, v1", "s_nop 4", "buffer_load_dwordx4 v[0:3], v4, s[8:11], 0 offen sc1",
                       "v_readlane_b32 s3, v9, 1", "v_mov_b32_e32 v1, 0", "v_mov_b32_e32 v2, 0", "s_nop 2", "global_store_dword v1, v2, s[2:3]",
                       "v_cmp_eq_u32_e64 s[30:31], s5, v3", "global_load_dword v5, v[6:7], off"]) == 1   # (no SGPR operand: nothing to wait for)


def test_build_compiles_each_unit_once_in_the_safe_form_and_installs_nothing_the_check_refuses(monkeypatch, tmp_path):
    """build_hip compiles every translation unit ONCE, in the source's default form (five wait states in front of every hand-scheduled load: no
    -DWN_AP_SGPR_HAZARD on the command line) -- the runtime unit with the branch-target alignment flag, the stacked-layer kernels' unit without it
    (round 6: csrc/wn_stacked_table.h) --, links them once; the disassembly check runs on the result and a library it refuses is never installed
    (nothing is left behind either)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "pytorch-wavenet_amd"))
    import build
    out = tmp_path / "libwn_mi355.so"
    monkeypatch.setattr(build, "OUT", str(out))
    monkeypatch.setattr(build, "_stale", lambda *a: True)
    compiles, links, verdicts = [], [], []

    def fake_compile_all(cmds):
        for cmd in cmds:
            compiles.append(list(cmd))
            open(cmd[cmd.index("-o") + 1], "wb").write(b"x")

    def fake_link(cmd, **kw):
        links.append(list(cmd))
        open(cmd[cmd.index("-o") + 1], "wb").write(b"x")

    def fake_check(path, objdump=None):
        v = verdicts.pop(0)
        if v:
            raise RuntimeError(v)
        return 11

    monkeypatch.setattr(build, "_compile_all", fake_compile_all)
    monkeypatch.setattr(build.subprocess, "check_call", fake_link)
    monkeypatch.setattr(build, "check_hand_scheduled_registers", fake_check)
    verdicts[:] = [None]
    assert build.build_hip(force=True) == str(out) and len(compiles) == len(build.SOURCES) == 2 and len(links) == 1 and out.exists()
    by_src = {c[c.index("-c") + 1]: c for c in compiles}
    assert sorted(by_src) == sorted(build.SOURCES)
    for src, c in by_src.items():
        assert not any(x.startswith("-DWN_AP_SGPR_HAZARD") for x in c) and "--offload-arch=gfx950" in c
        assert ("-align-all-nofallthru-blocks=6" in c) == (not src.endswith("wn_stacked.hip"))
    assert all(o in links[0] for o in (c[c.index("-o") + 1] for c in compiles)) and "-shared" in links[0]
    assert not any(os.path.exists(c[c.index("-o") + 1]) for c in compiles)   # (the objects are removed behind the link)
    compiles.clear(); links.clear(); out.unlink()
    verdicts[:] = ["k: use of a reserved poll register outside the hand-scheduled blocks in ..."]
    with pytest.raises(RuntimeError):
        build.build_hip(force=True)
    assert len(compiles) == 2 and len(links) == 1 and not out.exists() and not os.path.exists(str(out) + ".tmp")
    with open(os.path.join(ROOT, "pytorch-wavenet_amd", "csrc", "wn_kernel_v3.h")) as f:
        src = f.read()
    assert '#define WN_AP_SGPR_HAZARD "s_nop 4\\n\\t"' in src    # the source's default IS the safe form


def test_scratch_rule_covers_the_whole_library():
    """build.py rule 3, round 5: a spill in ANY kernel fails the build, except the recorded instantiations (none since round 6) -- and those only up to their
    recorded count."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "pytorch-wavenet_amd"))
    import build

    def dis(kernel, n):
        return "0000000000001000 <%s>:\n" % kernel + "\ts_nop 0\n" + "\tscratch_store_dword off, v1, off // 000: 0\n" * n + "0000000000002000 <L12>:\n\ts_endpgm\n"

    build._check_scratch_everywhere(dis("_Z11wn_fwd_gemmILi0EEv10WnGemmArgs", 0))
    with pytest.raises(RuntimeError, match="spills to scratch"):
        build._check_scratch_everywhere(dis("_Z11wn_fwd_gemmILi0EEv10WnGemmArgs", 1))
    with pytest.raises(RuntimeError, match="spills to scratch"):
        build._check_scratch_everywhere(dis("_Z17wn_bwd_layer_bf1614WnGemmArgsBf16S_", 2))
    assert build.KNOWN_SPILLS == {}   # round 6: no kernel of the library spills any more (the five recorded matrix-core instantiations got their registers)
    known = "_Z19wn_bwd_gemm_tn_bf16ILi8ELb0ELb1EEv12WnGemmTnArgs"
    with pytest.raises(RuntimeError, match="spills to scratch"):
        build._check_scratch_everywhere(dis(known, 1))
    build.KNOWN_SPILLS["wn_bwd_gemm_tn_bf16ILi8ELb0ELb1E"] = 3   # (the tolerance mechanism itself: up to the recorded count, not beyond)
    try:
        build._check_scratch_everywhere(dis(known, 3))
        with pytest.raises(RuntimeError, match="tolerated"):
            build._check_scratch_everywhere(dis(known, 4))
    finally:
        build.KNOWN_SPILLS.clear()


def test_graft_entry_build_runs():
    """The driver's "does it build" check (__graft_entry__.build()): builds the HIP library (checked: build.check_hand_scheduled_registers),
    the C oracle, loads the library through the binding and imports the facade -- on the CPU."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    ge = importlib.import_module("__graft_entry__")
    so = ge.build()
    assert os.path.basename(so) == "libwn_mi355.so"
