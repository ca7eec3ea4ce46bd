"""Where do the SGPR spills (v_writelane / v_readlane with an immediate lane on a spill VGPR) and the scratch accesses
of a kernel sit relative to its loops?
    tools/isa_count.sh <mangled-substring>; python tools/spill_map.py [/tmp/isa/kernel.s]
Prints, per loop depth (from the compiler's "in Loop: ... Depth=n" block comments), the number of spill writes /
reloads and scratch loads / stores, then the blocks at the deepest levels that contain any."""


def main():
    import collections
    import re
    import sys

    path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/isa/kernel.s"
    lines = open(path).read().split("\n")
    wl = collections.Counter()
    for ln in lines:
        m = re.match(r"\s+v_(writelane|readlane)_b32 (\S+), (\S+), (\d+)\s*$", ln)
        if m:
            wl[m.group(2).rstrip(",") if m.group(1) == "writelane" else m.group(3).rstrip(",")] += 1
    # spill VGPRs: the targets of v_writelane with immediate lanes (data lanes are written with s-register lane selects)
    spill_v = set(v for v, n in wl.items() if v.startswith("v") and n >= 4)
    depth, block = 0, "entry"
    per_depth = collections.defaultdict(collections.Counter)
    per_block = collections.defaultdict(collections.Counter)
    block_depth = {}
    for ln in lines:
        m = re.match(r"(\.LBB\S+):\s*(;.*)?$", ln)
        if m:
            block = m.group(1)
            d = re.search(r"Depth=(\d+)", ln)
            depth = int(d.group(1)) if d else 0
            block_depth[block] = depth
            continue
        kind = None
        m = re.match(r"\s+v_writelane_b32 (\S+), \S+, \d+\s*$", ln)
        if m and m.group(1).rstrip(",") in spill_v:
            kind = "sgpr_spill"
        m = re.match(r"\s+v_readlane_b32 \S+, (\S+), \d+\s*$", ln)
        if m and m.group(1).rstrip(",") in spill_v:
            kind = "sgpr_reload"
        if re.match(r"\s+scratch_load", ln):
            kind = "scratch_load"
        if re.match(r"\s+scratch_store", ln):
            kind = "scratch_store"
        if kind:
            per_depth[depth][kind] += 1
            per_block[block][kind] += 1
    print("spill VGPRs:", sorted(spill_v))
    for d in sorted(per_depth):
        print("depth %d: %s" % (d, dict(per_depth[d])))
    maxd = max(block_depth.values()) if block_depth else 0
    print("blocks at depth >= %d with spill / scratch traffic:" % max(1, maxd))
    for b, c in per_block.items():
        if block_depth.get(b, 0) >= max(1, maxd):
            print("   %-14s depth %d  %s" % (b, block_depth[b], dict(c)))


if __name__ == "__main__":
    main()
