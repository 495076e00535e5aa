"""Builds libwn_mi355.so (the HIP/gfx950 engine behind include/wn_abi.h) in-tree.

    python pytorch-wavenet_amd/build.py            # build if stale
    python pytorch-wavenet_amd/build.py --force

hipcc cross-compiles gfx950 without a GPU.  The .so is git-ignored but travels to the GPU box with
the gpurun snapshot.  gfx950 only: no other --offload-arch, no fallback.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "mi355_wavenet", "libwn_mi355.so")
SOURCES = [os.path.join(CSRC, "wn_runtime.hip"), os.path.join(CSRC, "wn_stacked.hip")]
# translation units built WITHOUT ALIGN_FLAGS (below): the stacked-layer kernels lose 1.2-1.3 % with them (csrc/wn_stacked_table.h)
UNALIGNED_SOURCES = {os.path.join(CSRC, "wn_stacked.hip")}
DEPS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inl"))) + [os.path.join(ROOT, "include", "wn_abi.h")]


def _stale(out, deps):
    return not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def hipcc_path():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 to build the gfx950 engine)")


def objdump_path():
    """llvm-objdump of the ROCm installation whose hipcc builds the library: $LLVM_OBJDUMP, next to the hipcc that was found
    (<rocm>/bin/hipcc -> <rocm>/lib/llvm/bin), $ROCM_PATH, /opt/rocm, then PATH.  Raises with the list of places tried."""
    tried = []
    cands = [os.environ.get("LLVM_OBJDUMP")]
    try:
        rocm = os.path.dirname(os.path.dirname(os.path.realpath(hipcc_path())))
        cands += [os.path.join(rocm, "lib", "llvm", "bin", "llvm-objdump"), os.path.join(rocm, "llvm", "bin", "llvm-objdump")]
    except RuntimeError:
        pass
    if os.environ.get("ROCM_PATH"):
        cands.append(os.path.join(os.environ["ROCM_PATH"], "lib", "llvm", "bin", "llvm-objdump"))
    cands += ["/opt/rocm/lib/llvm/bin/llvm-objdump", shutil.which("llvm-objdump")]
    for c in cands:
        if not c:
            continue
        tried.append(c)
        if os.path.exists(c):
            return c
    raise RuntimeError("llvm-objdump not found (tried %s; set LLVM_OBJDUMP): cannot verify the hand-scheduled register reservation" % ", ".join(tried))


RESERVED_FIRST, RESERVED_LAST = 152, 167   # csrc/wn_kernel_v3.h: WN_V3_COMPILER_VGPRS .. the last register a 768-thread workgroup leaves a lane


def check_hand_scheduled_registers(so, objdump=None):
    """The variant-3 kernels keep loads in flight into v152-v167 across their inline-assembly blocks (input poll sets, the queue
    group's tap FIFO).  The kernels carry amdgpu_num_vgpr so that the compiler's own allocation ends below them; this check
    DISASSEMBLES the built library and raises unless (1) every instruction of those kernels that names a reserved register is one
    the blocks emit, in the operand position they emit it -- a tap-FIFO take directly behind one of the FIFO's hand-counted
    `s_waitcnt vmcnt(n)` --, (2) nothing touches a register above v167, (3) the kernels use no scratch -- and, round 5, no OTHER kernel of the library does either beyond the
    recorded matrix-core instantiations (KNOWN_SPILLS) --, (4) no memory instruction of ANY kernel reads an SGPR a VALU instruction wrote less than five wait states earlier (a spill in a persistent hot loop is a performance bug, and spill code is where an allocator would reach for "free"
    registers).  Called by build_hip(): a library that breaks the invariant is never left in place."""
    import re
    import tempfile
    objdump = objdump or objdump_path()
    if not os.path.exists(objdump):
        raise RuntimeError("llvm-objdump not found at %s: cannot verify the hand-scheduled register reservation" % objdump)
    so = os.path.abspath(so)   # (everything below runs inside a temporary directory: nothing is ever written next to `so`)
    with tempfile.TemporaryDirectory() as tmp:
        local = shutil.copy(so, os.path.join(tmp, "lib.so"))
        subprocess.check_call([objdump, "--offloading", local], cwd=tmp, stdout=subprocess.DEVNULL)
        co = [f for f in os.listdir(tmp) if "gfx950" in f]
        if len(co) != len(SOURCES) and len(co) != 1:   # one code object per translation unit (a variant built in one hipcc call with several sources has as many)
            raise RuntimeError("expected %d gfx950 code object(s) in %s, found %r" % (len(SOURCES), so, co))
        # (branch targets as labels <Ln>: rule 4 follows them; the units' disassemblies one behind the other -- every rule is per kernel)
        dis = "\n".join(subprocess.check_output([objdump, "-d", "--symbolize-operands", os.path.join(tmp, c)]).decode() for c in sorted(co))
    reserved = set(range(RESERVED_FIRST, RESERVED_LAST + 1))
    is_res = lambda tok: bool(_regs(tok) & reserved)  # noqa: E731
    seen, current, prev = 0, None, ""
    items = {}     # kernel -> its instructions and branch-target labels in address order: ("ins", text) / ("label", name)  (rule 4)
    fifo_waits = {"s_waitcnt vmcnt(%d)" % n for n in (5, 6, 10, 11)}  # WN_V3_TAP_AHEAD = 6: D - 1, D, 2 D - 2, 2 D - 1 younger operations
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            if re.fullmatch(r"L\d+", m.group(1)):   # a branch target inside the current kernel (--symbolize-operands)
                if current:
                    items[current].append(("label", m.group(1)))
                continue
            current = m.group(1)
            items[current] = []
            if "wn_generate_kernel_v3m" in current:
                seen += 1
            continue
        if not current:
            continue
        text = line.split("//")[0].strip()
        if not text:
            continue
        items[current].append(("ins", text))   # (every kernel of the library: rule 4)
        if "wn_generate_kernel_v3m" not in current:
            continue
        before, prev = prev, text
        op, _, rest = text.partition(" ")
        ops = [o.strip() for o in rest.split(",")]
        regs = _regs(text)
        if regs and max(regs) > RESERVED_LAST:
            raise RuntimeError("%s: register above v%d: %s" % (current, RESERVED_LAST, text))
        if op.startswith("scratch_"):
            raise RuntimeError("%s spills to scratch: %s" % (current, text))
        if not (regs & reserved):
            continue
        where = "%s: %s" % (current, text)
        if op == "global_load_dwordx2":      # a request set: destination pair inside the reserved range, address operands outside
            ok = re.fullmatch(r"v\[(\d+):(\d+)\]", ops[0]) and _regs(ops[0]) <= reserved and not any(is_res(o) for o in ops[1:])
        elif op == "global_load_dword":      # the tap FIFO: destination v152-v157, address outside
            ok = ops[0] in ("v152", "v153", "v154", "v155", "v156", "v157") and not any(is_res(o) for o in ops[1:])
        elif op == "v_cmp_eq_u32_e32":       # vcc = (tag == v<reserved>): reserved register as the LAST source only
            ok = is_res(ops[-1]) and not any(is_res(o) for o in ops[:-1])
        elif op == "v_cmp_eq_u32_e64":       # s[m] = (tag == v<reserved>)
            ok = is_res(ops[-1]) and not any(is_res(o) for o in ops[:-1])
        elif op == "v_add_f32_e32":          # t0 = t0 + v<reserved>: never the destination
            ok = not is_res(ops[0]) and is_res(ops[-1])
        elif op == "v_mov_b32_e32":          # FIFO take: source only, and directly behind one of the FIFO's hand-counted waits
            ok = not is_res(ops[0]) and is_res(ops[1]) and before in fifo_waits
        else:
            ok = False
        if not ok:
            raise RuntimeError("use of a reserved poll register outside the hand-scheduled blocks in " + where)
    if not seen:
        raise RuntimeError("no wn_generate_kernel_v3m kernel found in %s" % so)
    for kernel, seq in items.items():
        _check_sgpr_base_hazard(kernel, seq, reserved)
    _check_scratch_everywhere(dis)
    return seen


# Kernels that are allowed to touch scratch: kernel-name fragment -> most scratch instructions tolerated.  EMPTY since round 6: the five matrix-core
# instantiations that spilled 2-10 VGPRs at their 128-register cap (rounds 4-5: wn_fwd_gemm_bf16<*, 4, false>, wn_bwd_gemm_tn_bf16<8, false, *> -- the forms
# that convert an fp32-stored operand on its way to LDS) now run at three waves per SIMD (150 / 146-154 VGPRs, csrc/wn_forward.h launch bounds).  The
# mechanism stays: a new spill anywhere fails the build instead of going unnoticed.
KNOWN_SPILLS = {}


def _check_scratch_everywhere(dis):
    """Rule 3 for the WHOLE library: no kernel may touch scratch except the recorded matrix-core instantiations, and those not more than recorded."""
    import re
    counts, current = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            if not re.fullmatch(r"L\d+", m.group(1)):
                current = m.group(1)
            continue
        text = line.split("//")[0].strip()
        if current and text.startswith("scratch_"):
            counts[current] = counts.get(current, 0) + 1
    for kernel, n in counts.items():
        allowed = max([v for k, v in KNOWN_SPILLS.items() if k in kernel] or [0])
        if n > allowed:
            raise RuntimeError("%s spills to scratch (%d scratch instructions, %d tolerated: build.py KNOWN_SPILLS)" % (kernel, n, allowed))


def _check_sgpr_base_hazard(kernel, seq, reserved):
    """Rule 4: an SGPR that a vector MEMORY instruction reads (the base pair of a global_* access, the descriptor / offset of a buffer_* one)
    must not have been written by a VALU instruction (v_readlane_b32 of a spilled pointer, v_readfirstlane_b32, a compare, ...) within the
    last FIVE wait states on ANY path that leads to it -- a gfx9 hazard the backend resolves for its own instructions but not in front of
    inline assembly (see WN_AP_SGPR_HAZARD in wn_kernel_v3.h).  Round 4 found it on the hand-scheduled input polls of one kernel form; since
    round 5 the rule is generic: EVERY memory instruction of EVERY kernel in the library is checked (the compiler's own pass it by
    construction; whatever an inline-assembly block emits is held to the same standard without the rule having to know the block).  `seq` is
    the kernel in address order, instructions and branch-target labels; the walk goes backwards through fall-throughs and through every
    branch that targets a label it meets (each instruction one wait state, s_nop n counts n + 1), until five wait states have passed or the
    scalar unit is found to have written the registers last."""
    import re
    targets = {}   # label -> indices of the branches that jump to it
    for i, (kind, text) in enumerate(seq):
        if kind == "ins" and text.startswith(("s_cbranch", "s_branch")):
            m = re.search(r"\b(L\d+)\b", text)
            if m:
                targets.setdefault(m.group(1), []).append(i)

    def walk(i, waited, pending, load, seen_states):
        # i: index of the next item to look at (going backwards); pending: registers of the pair whose last writer has not been seen yet
        while i >= 0 and waited < 5 and pending:
            kind, text = seq[i]
            if kind == "label":
                for b in targets.get(text, ()):   # every branch to this label is a predecessor (the branch itself is looked at there)
                    key = (b, waited, frozenset(pending))
                    if key not in seen_states:
                        seen_states.add(key)
                        walk(b, waited, set(pending), load, seen_states)
                # the fall-through predecessor -- unless the instruction in front is an unconditional transfer
                k = i - 1
                while k >= 0 and seq[k][0] == "label":
                    k -= 1
                if k >= 0 and seq[k][1].split(" ")[0] in ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64"):
                    return
                i -= 1
                continue
            op, _, rest = text.partition(" ")
            dst = rest.split(",")[0].strip()
            if re.fullmatch(r"s\d+|s\[\d+:\d+\]", dst):
                d = [int(x) for x in re.findall(r"\d+", dst)]
                d = set(range(d[0], d[-1] + 1))
                if op.startswith("v_") and d & pending:
                    raise RuntimeError("%s: %s: an SGPR operand of this memory instruction is written by `%s` %d wait state(s) before it (5 needed)" % (kernel, load, text, waited))
                if op.startswith("s_"):
                    pending = pending - d   # (written by the scalar unit: no hazard, and whatever wrote it before does not matter)
            waited += int(rest) + 1 if op == "s_nop" and rest.strip().isdigit() else 1
            i -= 1

    for i, (kind, text) in enumerate(seq):
        if kind != "ins" or not text.startswith(("global_", "buffer_", "flat_", "scratch_")):
            continue
        rest = text.partition(" ")[2]
        sregs = set(int(x) for x in re.findall(r"\bs(\d+)\b", rest))
        for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", rest):
            sregs.update(range(int(a), int(b) + 1))
        if sregs:
            walk(i - 1, 0, sregs, text, set())


def _regs(text):
    import re
    regs = set(int(x) for x in re.findall(r"\bv(\d+)\b", text))
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        regs.update(range(int(a), int(b) + 1))
    return regs


# -mllvm -align-all-nofallthru-blocks=6: every basic block that is only ever entered through a branch starts on a 64-byte instruction-cache line
# (the padding in front of it is never executed).  The persistent chains are latency bound and take ~10 branches per item: a target near the end of a
# fetch window costs a second fetch.  Measured on cfg3 x 64: 1.102 -> 1.107 M samples/s, reproducible to 0.05 % (profiles/r05_instruction_trims_and_alignment.txt).
ALIGN_FLAGS = ["-mllvm", "-align-all-nofallthru-blocks=6"]


def _compile_all(cmds):
    """The translation units' compiles, side by side; raises if one fails."""
    procs = [subprocess.Popen(c) for c in cmds]
    rcs = [p.wait() for p in procs]
    for c, rc in zip(cmds, rcs):
        if rc:
            raise subprocess.CalledProcessError(rc, c)


def build_hip(force=False, verbose=False, extra_flags=()):
    """One compile per translation unit, in the form whose hand-scheduled loads carry the five wait states (the source's default WN_AP_SGPR_HAZARD): correct whatever the
    register allocator does with the polls' base pointers.  (Round 4 compiled a form without them first and let the disassembly decide; round 5
    measured both forms of the same source on one box -- 64 streams of cfg3 1.094-1.101 M with the wait states against 1.068-1.113 M without,
    single stream 19.7-21.1 k either way (the spread is the box's, not the form's) -- and dropped the second form: nothing to gain, one hazard less to
    police.)  The disassembly check still runs on every build: reserved registers, no scratch in the generation kernels, and rule 4, which the wait
    states now satisfy by construction."""
    if not force and not _stale(OUT, DEPS):
        return OUT
    tmp_out = OUT + ".tmp"
    common = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-inline-asm", *extra_flags]
    objs, cmds = [], []
    for src in SOURCES:   # one object per translation unit (their flags differ: ALIGN_FLAGS), compiled side by side, then one link
        obj = os.path.join(os.path.dirname(OUT), "." + os.path.basename(src) + ".o")
        cmds.append(common + ([] if src in UNALIGNED_SOURCES else ALIGN_FLAGS) + ["-c", src, "-o", obj])
        objs.append(obj)
        if verbose:
            print(" ".join(cmds[-1]))
    _compile_all(cmds)
    link = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp_out] + objs
    if verbose:
        print(" ".join(link))
    try:
        subprocess.check_call(link)
    finally:
        for obj in objs:
            if os.path.exists(obj):
                os.remove(obj)
    try:
        check_hand_scheduled_registers(tmp_out)   # a library that breaks the reservation is never installed
    except Exception:
        os.remove(tmp_out)
        raise
    os.replace(tmp_out, OUT)
    return OUT


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
