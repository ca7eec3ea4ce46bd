"""Instruction budget of one site of a sweep kernel, by functional class, from the device assembly of a -DLLDA_BUDGET_MARKS build
(csrc/device_common.hpp: LLDA_MARK puts "; @class" comment lines at the class boundaries; the marks are scheduling barriers, so the
marked build is for counting, not for timing).

    python tools/site_loop_budget.py <mangled-kernel-substring> [docs_per_wavefront]

Takes the SECOND unrolled copy of the site (between two "; @site_top" marks), attributes every instruction to the mark in front of
it, drops the classes that start with "rare_" (out-of-line or behind a rarely taken branch) into a separate table, and prices the
vector instructions with the two issue rates of profiles/r04_valu_ubench.txt (plain 32-bit VOP1/VOP2: ~1.1 ns per instruction and
SIMD with two or more wavefronts, everything else -- VOP3, packed, conversions, DPP, SDWA, lane reads, compares into SGPRs -- ~1.8 ns)."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST_NS, SLOW_NS = 1.1, 1.8


def kind(ins):
    op = ins.split()[0]
    if op.startswith("v_"):
        slow = ("_e64" in op or "_dpp" in op or "_sdwa" in op or op.startswith(("v_pk_", "v_cvt", "v_readlane", "v_writelane", "v_readfirstlane",
                "v_rcp", "v_cmp", "v_fma", "v_lshl_add", "v_lshl_or", "v_and_or", "v_or3", "v_add3", "v_bfe", "v_mad", "v_cndmask", "v_min3",
                "v_perm", "v_alignbit", "v_add_lshl", "v_mul_lo", "v_mul_hi")))
        return "valu_slow" if slow else "valu_fast"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"


def main():
    sub = sys.argv[1]
    docs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    os.makedirs("/tmp/isa_marks", exist_ok=True)
    asm = "/tmp/isa_marks/llda.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                           "-I" + os.path.join(ROOT, "include"), "-Wno-unused-function", "--cuda-device-only", "-S", "-DLLDA_BUDGET_MARKS",
                           "-o", asm, os.path.join(ROOT, "lda_thesis_amd", "csrc", "llda_gibbs.hip")], stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"_ZN.*%s.*:" % re.escape(sub), l))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    tops = [i for i, l in enumerate(body) if l.strip() == "; @site_top"]
    if len(tops) < 3:
        raise SystemExit("fewer than three '; @site_top' marks in %s" % sub)
    a, b = tops[1], tops[2]
    cls, per = "site_top", collections.OrderedDict()
    for l in body[a:b]:
        t = l.strip()
        m = re.match(r"; @(\w+)$", t)
        if m:
            cls = m.group(1)
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        per.setdefault(cls, collections.Counter())[kind(t)] += 1
    cols = ["valu_fast", "valu_slow", "salu", "s_nop", "s_waitcnt", "branch", "lds", "vmem"]
    print("kernel %s: one site iteration of a wavefront = %d sites (lines %d .. %d of the kernel's assembly)\n" % (sub, docs, a, b))
    print("| class | " + " | ".join(cols) + " | VALU | VALU issue ns | all instructions |")
    print("|---|" + "---|" * (len(cols) + 3))
    tot, rare = collections.Counter(), collections.Counter()
    for c, k in per.items():
        v = k["valu_fast"] + k["valu_slow"]
        ns = k["valu_fast"] * FAST_NS + k["valu_slow"] * SLOW_NS
        row = "| %s | " % c + " | ".join(str(k[x]) for x in cols) + " | %d | %.0f | %d |" % (v, ns, sum(k.values()))
        print(row)
        (rare if c.startswith("rare_") else tot).update(k)
    v = tot["valu_fast"] + tot["valu_slow"]
    print("| **hot path** | " + " | ".join(str(tot[x]) for x in cols) + " | **%d** | %.0f | %d |" %
          (v, tot["valu_fast"] * FAST_NS + tot["valu_slow"] * SLOW_NS, sum(tot.values())))
    print("\nper site: %.1f VALU, %.1f instructions of any kind on the hot path; rare blocks: %d VALU in the listing" %
          (v / docs, sum(tot.values()) / docs, rare["valu_fast"] + rare["valu_slow"]))


if __name__ == "__main__":
    main()
