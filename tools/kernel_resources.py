#!/usr/bin/env python3
"""Registers, scratch and occupancy of the kernels of a HIP source, from hipcc's own resource-usage remarks
(no GPU needed): `python tools/kernel_resources.py slab.hip k_team_rows k_slab_close`.  A kernel of the one-pass
family that shows scratch, or fewer waves per SIMD than its launch needs, is a regression the ISA would show too."""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adaptive-proximal-algorithms_amd", "csrc")


def main():
    src = sys.argv[1]
    pats = sys.argv[2:]
    out = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-c",
                          os.path.join(CSRC, src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True).stderr
    for b in re.split(r"remark: [^\n]*Function Name: ", out)[1:]:
        name = b.split("\n")[0].strip()
        try:
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except FileNotFoundError:
            pass
        name = name.replace("apx::", "")
        if pats and not any(p in name for p in pats):
            continue
        g = lambda k: re.search(re.escape(k) + r": (\d+)", b).group(1)
        print("V %3s AGPR %3s S %3s scratch %4s occ %s lds %6s  %s" % (
            g("VGPRs"), g("AGPRs"), g("SGPRs"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"),
            g("LDS Size [bytes/block]"), name[:150]))


if __name__ == "__main__":
    main()
