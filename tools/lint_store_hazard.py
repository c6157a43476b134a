"""ISA lint for a gfx950 store-data hazard this hipcc does not cover (tools/GFX950_NOTES.md, "store data").

A buffer_store_dwordx3/x4 whose soffset is an SGPR reads its data VGPRs over more than one cycle; a VALU instruction that
writes one of them in the next TWO issue slots can reach the register first (seen as a corrupted dword in 16 lanes of a wave,
once in ~10^5 stores).  The recognizer of this compiler inserts no wait state, so every such store in this tree is followed by
a two-wait-state s_nop that takes the stored registers as inputs - they stay live up to it.  This script compiles the .hip
files to gfx950 assembly and checks the rule on what the compiler actually emitted.

Second rule (round 5): the compiler pads ITS OWN consumers of MFMA results with the XDL -> VALU wait states (8-pass MFMA: 11), but an
INLINE-ASM instruction that reads a register an MFMA has just written gets none: it reads the accumulator before the matrix pipe has
written it back.  Found as run-to-run noise of 1e-7 in the -O1 (ASAN) build after a fold of accumulators had been rewritten as inline-asm
v_fma_f32 (repeatable, within every tolerance, at -O3).  The lint flags every instruction inside an ;;#ASMSTART / ;;#ASMEND block that
reads a VGPR written by a v_mfma fewer than MFMA_STATES issue slots earlier (another MFMA that takes it whole as C is the accumulate
chain: exempt; an s_nop n counts n + 1 slots; every other instruction 1 - conservative: an MFMA in between occupies more).  The
reverse case is checked too: an MFMA that is ITSELF inline asm (tools/experiments/wsplit: B operands in AGPRs) is unknown to the compiler, so
any instruction reading its result inside the window is flagged.  Destinations and sources in the AGPR half (a[..], v_accvgpr_read) are
tracked like VGPRs (round 6); `--selftest` feeds known-bad and known-good snippets through the rule.

usage: python tools/lint_store_hazard.py [file.hip ...]      (default: every pfnl_amd/csrc/*.hip)      exit status 1 on a hit
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
STORE = re.compile(r"^\s*buffer_store_dwordx[34]\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(\S+)")
VDST = re.compile(r"^\s*(v_\S+)\s+(v\[(\d+):(\d+)\]|v(\d+))")
NOP = re.compile(r"^\s*s_nop\s+(\d+)")


def instructions(path):
    for ln in open(path):
        s = ln.split(";")[0].rstrip()
        if not s.strip() or s.lstrip().startswith((".", "//")) or s.rstrip().endswith(":"):
            continue
        yield s


def written(ins):
    m = VDST.match(ins)
    if not m or m.group(1).startswith(("v_cmp", "v_nop")):          # (v_cmp writes vcc / an SGPR pair)
        return set()
    if m.group(5) is not None:
        return {int(m.group(5))}
    return set(range(int(m.group(3)), int(m.group(4)) + 1))


def lint(asm, need=2):
    ins = list(instructions(asm))
    hits = []
    for i, s in enumerate(ins):
        m = STORE.match(s)
        if not m or not m.group(4).startswith("s"):                  # soffset 0 / immediate: no hazard
            continue
        data = set(range(int(m.group(1)), int(m.group(2)) + 1))
        waited, j = 0, i + 1
        while waited < need and j < len(ins):
            n = NOP.match(ins[j])
            if n:
                waited += int(n.group(1)) + 1
            else:
                if written(ins[j]) & data:
                    hits.append((s.strip(), ins[j].strip(), waited))
                    break
                waited += 1
            j += 1
    return hits


MFMA = re.compile(r"^\s*v_mfma_\S+\s+([va])\[(\d+):(\d+)\]")          # destination in the VGPR or the AGPR half of the register file
VREG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")
MFMA_STATES = 11                                                   # 8-pass XDL write -> VALU read (12 for safety below 16-pass: these kernels use 8-pass MFMAs)


def raw_lines(path):
    """(instruction, inside an inline-asm block) pairs"""
    inside = False
    for ln in open(path):
        if ";;#ASMSTART" in ln:
            inside = True
            continue
        if ";;#ASMEND" in ln:
            inside = False
            continue
        s = ln.split(";")[0].rstrip()
        if not s.strip() or s.lstrip().startswith((".", "//")) or s.rstrip().endswith(":"):
            continue
        yield s, inside


def regs_read(ins):
    """Registers ("v12" / "a12": VGPRs and AGPRs) an instruction names as sources (every operand behind the first: destinations that are
    also sources - accumulate forms - are named again among them).  v_accvgpr_read_b32 v, aN reads aN."""
    ops = ins.split(None, 1)
    if len(ops) < 2:
        return set()
    parts = ops[1].split(",")
    out = set()
    for part in parts[1:]:
        for m in VREG.finditer(part):
            if m.group(4) is not None:
                out.add(m.group(4) + m.group(5))
            else:
                out.update(m.group(1) + str(r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def written_regs(ins):
    """Registers the first operand of a vector instruction names (its destination); stores, branches, waits: none."""
    t = ins.strip()
    if not t.startswith(("v_", "ds_read", "global_load", "buffer_load", "scratch_load")) or t.startswith("v_cmp"):
        return set()
    ops = t.split(None, 1)
    if len(ops) < 2:
        return set()
    first = ops[1].split(",")[0]
    out = set()
    for m in VREG.finditer(first):
        if m.group(4) is not None:
            out.add(m.group(4) + m.group(5))
        else:
            out.update(m.group(1) + str(r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def mfma_dst(ins):
    m = MFMA.match(ins)
    return {m.group(1) + str(r) for r in range(int(m.group(2)), int(m.group(3)) + 1)} if m else None


def lint_mfma_lines(ins):
    """ins: [(instruction, inside inline asm)]"""
    hits = []
    for i, (s, mfma_in_asm) in enumerate(ins):
        dst = mfma_dst(s)
        if not dst:
            continue
        waited, j = 0, i + 1
        while waited < MFMA_STATES and j < len(ins):
            t, inside = ins[j]
            n = NOP.match(t)
            if n:
                waited += int(n.group(1)) + 1
            else:
                d2 = mfma_dst(t)
                if d2 and d2 & dst:
                    break                                            # the accumulate chain (or the register is rewritten): the rule ends here
                # (an MFMA that is itself inline asm is unknown to the compiler: then EVERY reader counts)
                if (inside or mfma_in_asm) and not t.lstrip().startswith(("s_", "v_mfma")) and regs_read(t) & dst:
                    hits.append((s.strip(), t.strip(), waited))
                    break
                if not inside and not mfma_in_asm:
                    # a COMPILER-generated instruction that overwrites a result register (it got its own wait states): what is read from that
                    # register from here on is not the MFMA's result any more (round 6: v_mul_f32 v48, .. / asm v_max_f32 v48, .., v48 behind a
                    # v_mfma .. v[48:51] that had just MOVED its accumulator elsewhere)
                    dst = dst - written_regs(t)
                    if not dst:
                        break
                waited += 1
            j += 1
    return hits


def lint_mfma_asm(asm):
    return lint_mfma_lines(list(raw_lines(asm)))


def selftest():
    """Known-bad and known-good snippets (a CPU test runs this): the rule must fire on an inline-asm reader of a fresh MFMA result in the
    VGPR AND in the AGPR half (v_accvgpr_read of an AGPR accumulator), on any reader of an inline-asm MFMA, and stay quiet behind enough
    wait states, on the accumulate chain, and on compiler-generated readers of compiler-generated MFMAs."""
    A, C_ = True, False                                              # inside inline asm / compiler-generated
    bad = [
        [("v_mfma_f32_32x32x16_f16 v[0:15], v[16:19], v[20:23], v[0:15]", C_), ("v_fma_f32 v40, v3, v41, v42", A)],
        [("v_mfma_f32_32x32x16_f16 a[0:15], v[16:19], v[20:23], a[0:15]", C_), ("s_nop 3", C_), ("v_accvgpr_read_b32 v40, a7", A)],
        [("v_mfma_f32_32x32x16_f16 a[16:31], v[16:19], a[0:3], a[16:31]", A), ("v_accvgpr_read_b32 v40, a16", C_)],
        [("v_mfma_f32_32x32x16_f16 v[0:15], v[16:19], a[0:3], v[0:15]", A), ("v_add_f32 v40, v[0:1], v41", C_)],
    ]
    good = [
        [("v_mfma_f32_32x32x16_f16 v[0:15], v[16:19], v[20:23], v[0:15]", C_), ("s_nop 7", C_), ("s_nop 2", C_), ("v_fma_f32 v40, v3, v41, v42", A)],
        [("v_mfma_f32_32x32x16_f16 a[0:15], v[16:19], v[20:23], a[0:15]", C_), ("v_mfma_f32_32x32x16_f16 a[0:15], v[24:27], v[20:23], a[0:15]", C_)],
        [("v_mfma_f32_32x32x16_f16 v[0:15], v[16:19], v[20:23], v[0:15]", C_), ("v_add_f32 v40, v3, v41", C_)],
        [("v_mfma_f32_32x32x16_f16 a[0:15], v[16:19], v[20:23], a[0:15]", C_), ("v_accvgpr_read_b32 v40, a16", A)],
        # (round 6) a compiler-generated VALU has overwritten the result register: the inline-asm reader reads THAT value, not the MFMA's
        [("v_mfma_f32_16x16x32_f16 v[48:51], v[16:19], v[20:23], v[48:51]", C_), ("s_nop 3", C_), ("v_mul_f32_e32 v48, v60, v86", C_), ("v_max_f32 v48, v86, v48", A)],
    ]
    bad.append([("v_mfma_f32_16x16x32_f16 v[48:51], v[16:19], v[20:23], v[48:51]", C_), ("v_mul_f32_e32 v48, v60, v86", C_), ("v_max_f32 v40, v86, v49", A)])   # (v49 is still the MFMA's)
    for k, snip in enumerate(bad):
        assert len(lint_mfma_lines(snip)) == 1, ("bad snippet %d not flagged" % k, snip)
    for k, snip in enumerate(good):
        assert not lint_mfma_lines(snip), ("good snippet %d flagged" % k, snip)
    return 0


def main():
    if sys.argv[1:] == ["--selftest"]:
        return selftest()
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "pfnl_amd", "csrc", "*.hip")))
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            asm = os.path.join(tmp, os.path.basename(f) + ".s")
            subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-S", "--cuda-device-only", "-o", asm, f],
                           check=True, stderr=subprocess.DEVNULL)
            nstores = sum(1 for s in instructions(asm) if (m := STORE.match(s)) and m.group(4).startswith("s"))
            hits = lint(asm)
            print("%-24s %3d wide buffer stores with an SGPR offset, %d hazards" % (os.path.basename(f), nstores, len(hits)))
            for st, wr, w in hits:
                print("    %s\n      -> %s   (after %d wait states)" % (st, wr, w))
            bad += len(hits)
            h2 = lint_mfma_asm(asm)
            if h2:
                print("    %d inline-asm reader(s) of fresh MFMA results:" % len(h2))
            for mf, rd, w in h2[:8]:
                print("    %s\n      -> %s   (after %d issue slots, %d needed)" % (mf, rd, w, MFMA_STATES))
            bad += len(h2)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
