#!/usr/bin/env python3
"""Registers, scratch and occupancy of every kernel of the library, from the
compiler's own resource report (no GPU needed).

    python tools/kernel_resources.py [filter-substring] [extra hipcc flags...]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vulkan_renderer_amd.build import HIPCC_FLAGS, hipcc  # noqa: E402


def main() -> int:
    want = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else ""
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    flags = [f for f in HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    cmd = [hipcc()] + flags + extra + ["-c", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                                       "-o", "/dev/null",
                                       os.path.join(ROOT, "vulkan_renderer_amd", "csrc", "vkrhip.hip")]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: +([A-Za-z][^:]*): (.*?) \[-Rpass", line)
        if not m:
            continue
        key, value = m.group(1).strip(), m.group(2).strip()
        if key == "Function Name":
            cur = {"name": value}
            rows.append(cur)
        elif cur is not None:
            cur[key] = value
    names = subprocess.run(["c++filt"] + [r["name"] for r in rows],
                           capture_output=True, text=True).stdout.splitlines()
    print(f'{"kernel":92s} {"SGPR":>5s} {"VGPR":>5s} {"spill":>5s} {"scratch":>8s} {"occ":>4s} {"LDS":>7s}')
    for r, n in zip(rows, names):
        n = re.sub(r"\(.*\)$", "", n).replace("vkr::", "").replace("void ", "")
        if want and want not in n:
            continue
        print(f'{n[:92]:92s} {r.get("TotalSGPRs", r.get("SGPRs", "?")):>5s} {r.get("VGPRs", "?"):>5s} '
              f'{r.get("SGPRs Spill", "?") + "/" + r.get("VGPRs Spill", "?"):>5s} {r.get("ScratchSize [bytes/lane]", "?"):>8s} '
              f'{r.get("Occupancy [waves/SIMD]", "?"):>4s} {r.get("LDS Size [bytes/block]", "?"):>7s}')
    return 0


if __name__ == "__main__":
    sys.exit(main())
