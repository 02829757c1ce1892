"""Static check of the gfx950 assembly of every kernel source: is any MFMA result read (or
overwritten) by a non-MFMA instruction fewer wait states later than the matrix pipe needs?

Why: hipcc (ROCm 7.2) covers the "XDL write VGPR -> VALU / VMEM / LDS access" hazard with s_nops
inside a basic block, but in ``conv_c64`` it left the first ``v_cvt_pk_bf16_f32`` of an accumulator
2 wait states behind an ``s_cbranch`` that followed the last MFMA (needs 11 for an 8-pass MFMA):
wrong results on the GPU, nothing at compile time (docs/KERNELS.md, ``conv_c64``).  This walks the
assembly along fall-through AND branch targets and reports every such place.

  python bench/check_mfma_hazards.py [file.hip ...]      (default: every csrc/*.hip; exit 1 on a finding)

It also fails on any kernel that needs scratch memory (``.amdhsa_private_segment_fixed_size`` > 0).

Wait states are counted the way LLVM's hazard recognizer does (one per instruction, ``s_nop N`` =
N + 1), so an intervening independent MFMA counts 1 although it occupies the pipe for its passes —
conservative.  Needed: passes + 3, with the pass counts of gfx950's double-K forms as measured
(MI355X_MICROARCH.md cycle constants: 16x16x32 16 cycles = 4 passes, 32x32x16 32 cycles = 8) —
calibrated against the s_nops hipcc itself emits inside single basic blocks of these kernels (it
leaves 8 wait states behind a 16x16x32 MFMA there).  Inline-asm bodies are ordinary lines of the
.s file and are checked like the rest.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "distributed_training_pytorch_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
LABEL = re.compile(r"^([.\w$]+):")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def need_of(mn):
    if "32x32" in mn:
        return 11
    if "16x16" in mn:
        return 7
    return 5


def parse(path):
    """-> {kernel: [(mnemonic, operand text)]} and label -> index maps."""
    kernels, cur, labels = {}, None, None
    for line in open(path):
        s = line.split(";")[0].rstrip()
        if not s.strip():
            continue
        m = LABEL.match(s)
        if m:
            name = m.group(1)
            if name.startswith("_Z") or not name.startswith("."):
                cur, labels = [], {}
                kernels[name] = (cur, labels)
            elif cur is not None:
                labels[name] = len(cur)
            continue
        if cur is None or not s.startswith("\t"):
            continue
        t = s.strip()
        if t.startswith("."):
            continue
        parts = t.split(None, 1)
        cur.append((parts[0], parts[1] if len(parts) > 1 else ""))
    return kernels


def check_kernel(name, ins, labels):
    findings = []
    for i, (mn, ops) in enumerate(ins):
        if not mn.startswith("v_mfma") and not mn.startswith("v_smfmac"):
            continue
        dst = regs_of(ops.split(",")[0])
        need = need_of(mn)
        # walk forward along every path until `need` wait states have passed
        stack, seen = [(i + 1, 0)], set()
        while stack:
            j, ws = stack.pop()
            while j < len(ins) and ws < need:
                if (j, ws) in seen:
                    break
                seen.add((j, ws))
                m2, o2 = ins[j]
                if m2.startswith("v_mfma") or m2.startswith("v_smfmac"):
                    # a later MFMA on the same registers (accumulation chain, or a new value in
                    # them) is the pipe's own business; it also ends our interest in these registers
                    if regs_of(o2.split(",")[0]) & dst:
                        break
                    ws += 1
                elif m2 == "s_nop":
                    ws += int(o2.strip() or 0) + 1
                elif m2 == "s_endpgm":
                    break
                elif m2 == "s_branch":
                    j = labels.get(o2.strip(), len(ins))
                    ws += 1
                    continue
                else:
                    if m2.startswith("s_cbranch"):
                        tgt = labels.get(o2.strip())
                        if tgt is not None:
                            stack.append((tgt, ws + 1))
                    elif regs_of(o2) & dst and not m2.startswith("s_"):
                        findings.append((name, i, mn + " " + ops, j, m2 + " " + o2, ws, need))
                        break
                    ws += 1
                j += 1
    return findings


def main():
    files = sys.argv[1:] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    bad = 0
    with tempfile.TemporaryDirectory() as td:
        defs = os.environ.get("DTP_BUILD_DEFS", "").split()

        def to_asm(f):
            out = os.path.join(td, os.path.basename(f) + ".s")
            return out, subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", *defs,
                                        "-I", CSRC, "-S", "--cuda-device-only", "-o", out, f], capture_output=True, text=True)
        # the compilations are independent: run them side by side (igemm.hip alone takes a minute)
        import concurrent.futures
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(files), os.cpu_count() or 1)) as ex:
            compiled = list(ex.map(to_asm, files))
        for f, (out, r) in zip(files, compiled):
            if r.returncode != 0:
                print("%s: does not compile to assembly\n%s" % (f, r.stderr[-2000:]))
                return 2
            n_mfma, found = 0, []
            for name, (ins, labels) in parse(out).items():
                n_mfma += sum(1 for mn, _ in ins if mn.startswith("v_mfma"))
                found += check_kernel(name, ins, labels)
            # second check on the same assembly: no kernel of the tree may need scratch memory (a register
            # spill in an epilogue lambda once cost a 256x256 tile more than the feature it came with saved)
            txt = open(out).read()
            scratch = [(m.group(1), int(m.group(2))) for m in re.finditer(
                r"\.amdhsa_kernel (\S+)\n(?:.*\n)*?\s*\.amdhsa_private_segment_fixed_size (\d+)", txt)]
            spilled = [(n, b) for n, b in scratch if b > 0]
            for n, b in spilled:
                print("   %s: %d bytes of scratch per lane (register spill)" % (n, b))
            bad += len(spilled)
            print("%-22s %5d MFMAs, %d finding(s), %d of %d kernels with scratch" % (
                os.path.basename(f), n_mfma, len(found), len(spilled), len(scratch)))
            for name, i, a, j, b, ws, need in found[:20]:
                print("   %s\n     [%d] %s\n     [%d] %s   <- %d wait states later, needs %d" % (name, i, a, j, b, ws, need))
            bad += len(found)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
