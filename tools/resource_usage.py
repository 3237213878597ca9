#!/usr/bin/env python3
"""Table of the compiler's resource usage per kernel instantiation (registers, spills,
scratch, occupancy) from `hipcc -Rpass-analysis=kernel-resource-usage`.

  python tools/resource_usage.py [remarks.txt] > profiles/<round>/resource_usage.txt

Without an argument the library source is compiled (about two minutes) into a temp dir."""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(name):
    try:
        out = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except OSError:
        return name
    out = re.sub(r"\(rtamd_scene_desc.*", "", out)
    return out.replace("void (anonymous namespace)::", "")


def main():
    if len(sys.argv) > 1:
        text = open(sys.argv[1]).read()
    else:
        sys.path.insert(0, REPO)
        import __graft_entry__ as entry
        with tempfile.TemporaryDirectory() as tmp:
            cmd = [entry.HIPCC] + entry.HIP_FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(tmp, "lib.so"),
                                                      os.path.join(entry.CSRC, "rt_kernels.hip")]
            text = subprocess.run(cmd, capture_output=True, text=True).stderr
    keys = (("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("SGPRs", "sgpr"), ("VGPRs Spill", "vgpr_spill"),
            ("SGPRs Spill", "sgpr_spill"), (r"ScratchSize \[bytes/lane\]", "scratch_B_per_lane"),
            (r"Occupancy \[waves/SIMD\]", "waves_per_simd"), (r"LDS Size \[bytes/block\]", "lds_B"))
    print("trace_kernel<MODE (0 pixels, 1 rays, 2 tiles), COUNT (0 timed, 1 reference work, 2 executed work), KIND (0 K_ANALYTIC, 1 K_TREES, "
          "2 K_EXT, 3 K_AREA_TINY, 4 K_AREA, 5 K_AREA_FUSED, 6 K_STRAIGHT, 7 K_STRAIGHT_ONE, 8 K_PATH)>")
    print("%-44s %s" % ("kernel", " ".join("%18s" % k for _, k in keys)))
    for block in re.split(r"remark: [^\n]*Function Name: ", text)[1:]:
        name = demangle(block.split(" [-Rpass")[0].strip())
        vals = []
        for pat, _ in keys:
            m = re.search(r" " + pat + r": (\S+)", block)
            vals.append(m.group(1) if m else "?")
        print("%-44s %s" % (name, " ".join("%18s" % v for v in vals)))


if __name__ == "__main__":
    main()
