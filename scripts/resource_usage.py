#!/usr/bin/env python3
"""
Compiler resource usage of every kernel of liblynxhip: VGPR / AGPR / SGPR / scratch / LDS /
occupancy, as `hipcc -Rpass-analysis=kernel-resource-usage` reports them for gfx950.

    python scripts/resource_usage.py > profiles/rNN_kernel_resource_usage.txt

Runs on the CPU box (hipcc cross-compiles); rebuilds lynx_amd/_lib/liblynxhip.so on the way.
"""
import re
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def demangle(names):
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not tool:
        return names
    out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return out[: len(names)]


def main():
    res = subprocess.run(["bash", str(ROOT / "lynx_amd" / "csrc" / "build.sh"), "--report"], capture_output=True, text=True)
    if res.returncode != 0:
        sys.exit(res.stderr[-2000:])
    blocks = re.split(r"remark: [^\n]*Function Name: ", res.stderr)[1:]
    rows = []
    for b in blocks:
        def g(key):
            m = re.search(re.escape(key) + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        rows.append((b.split("\n")[0].strip(), g("VGPRs"), g("AGPRs"), g("SGPRs"), g("ScratchSize [bytes/lane]"),
                     g("Occupancy [waves/SIMD]"), g("LDS Size [bytes/block]")))
    names = demangle([r[0] for r in rows])
    print("# hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage (static LDS only; dynamic LDS is sized per launch)")
    print(f"{'kernel':100s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>7s}")
    for r, n in sorted(zip(rows, names), key=lambda x: x[1]):
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"\((lynx::LatticeDev|float|double|int|long|unsigned|const|lynx_).*$", "", n)
        print(f"{n[:100]:100s} {r[1]:5d} {r[2]:5d} {r[3]:5d} {r[4]:8d} {r[5]:4d} {r[6]:7d}")


if __name__ == "__main__":
    main()
