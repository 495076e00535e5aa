"""dev tool: build an A/B variant of the engine into tools/variants/libwn_<name>.so with the product's own recipe (fast poll form first,
the disassembly check decides; see build.py) plus -DWN_EXPERIMENT and the given -D flags.

    python tools/build_variant.py tapA -DWN_V3_TAP_AT_A=1
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-wavenet_amd"))
import build as wn_build  # noqa: E402


def main():
    args = sys.argv[1:]
    force_fast = "--force-fast" in args   # A/B runs only: keep the fast poll form whatever the hazard check says (report it)
    force_safe = "--safe" in args
    hazard = None
    if "--hazard" in args:   # e.g. --hazard 's_nop 4;s_nop 4' : the wait states in front of the hand-scheduled loads (A/B runs)
        i = args.index("--hazard")
        hazard = "".join(part.strip() + "\\n\\t" for part in args[i + 1].split(";") if part.strip())
        del args[i:i + 2]
    args = [a for a in args if a not in ("--force-fast", "--safe")]
    name, flags = args[0], args[1:]
    out = os.path.join(ROOT, "tools", "variants", "libwn_%s.so" % name)
    base = [wn_build.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function", "-Wno-inline-asm",
            "-DWN_EXPERIMENT", *flags]
    for extra in ([['-DWN_AP_SGPR_HAZARD="%s"' % hazard]] if hazard is not None else [[]] if force_safe else (['-DWN_AP_SGPR_HAZARD=""'], [])):
        subprocess.check_call(base + extra + ["-o", out] + wn_build.SOURCES)
        try:
            wn_build.check_hand_scheduled_registers(out)
            print("%s: built (%s)" % (out, " ".join(extra) if extra else "source default: safe poll form"))
            return 0
        except Exception as e:
            print("%s: %s" % (name, e))
            if force_fast and "SGPR operand of this memory instruction" in str(e):
                print("%s: kept in the FAST form regardless (A/B only)" % out)
                return 0
            if not extra or "SGPR operand of this memory instruction" not in str(e):
                os.remove(out)
                return 1
    return 1


if __name__ == "__main__":
    sys.exit(main())
