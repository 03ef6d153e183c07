"""profiles/<tag>_resources.txt from a raw dump of hipcc -Rpass-analysis=kernel-resource-usage (see tools/resources.sh)."""
import re
import subprocess
import sys

raw, dst = sys.argv[1], sys.argv[2]
out = ["# Kernel resource usage of libwavehip (hipcc -Rpass-analysis=kernel-resource-usage, --offload-arch=gfx950, -O3)",
       "# source | kernel | SGPRs | VGPRs | AGPRs | scratch B/lane | waves/SIMD | SGPR spills | VGPR spills | static LDS B", ""]
K = {"sg": "TotalSGPRs", "vg": "VGPRs", "ag": "AGPRs", "sc": r"ScratchSize \[bytes/lane\]", "oc": r"Occupancy \[waves/SIMD\]",
     "ss": "SGPRs Spill", "vs": "VGPRs Spill", "ld": r"LDS Size \[bytes/block\]"}
for l in open(raw):
    src, rest = l.split(": ", 1)
    m = re.search(r"Function Name: (\S+)", rest)
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    v = {k: re.search(p + r": (\d+)", rest).group(1) for k, p in K.items()}
    out.append(f"{src:26s} {name[:64]:64s} {v['sg']:>4} {v['vg']:>4} {v['ag']:>4} {v['sc']:>4} {v['oc']:>2} {v['ss']:>4} {v['vs']:>3} {v['ld']:>6}")
open(dst, "w").write("\n".join(out) + "\n")
print(len(out) - 3, "kernels")
