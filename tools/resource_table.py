"""Register / LDS / spill table of every kernel in the library from `make -C lda_thesis_amd/csrc resource` output.
    make -C lda_thesis_amd/csrc resource 2>&1 | python tools/resource_table.py [--spills]"""


def main():
    import re
    import subprocess
    import sys

    txt = sys.stdin.read()
    only_spills = "--spills" in sys.argv
    for b in txt.split("Function Name:")[1:]:
        name = b.split()[0]
        def g(k):
            m = re.search(k + r":\s*(\d+)", b)
            return int(m.group(1)) if m else -1
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", dn))
        dn = re.sub(r"\(.*$", "", dn)[:84]
        row = (dn, g("TotalSGPRs"), g(r"\bVGPRs"), g("AGPRs"), g("SGPRs Spill"), g("VGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"),
               g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"))
        if only_spills and row[4] <= 0 and row[5] <= 0 and row[6] <= 0:
            continue
        print("%-86s SGPR %3d VGPR %3d AGPR %3d  spill s%3d v%3d  scratch %4d  occ %2d  LDS %6d" % row)


if __name__ == "__main__":
    main()
