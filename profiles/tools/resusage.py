"""Per-kernel register / spill table of one HIP translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python profiles/tools/resusage.py csrc/mv_gemm_ring.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Rpass-analysis=kernel-resource-usage",
       "-c", src, "-o", "/dev/null"] + sys.argv[2:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line) or re.search(r" Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z /\[\]]+?):\s+(\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
if not rows:
    print(err[-3000:])
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:110]
    print(f"{name:110s} VGPR {v.get('VGPRs', -1):3d} AGPR {v.get('AGPRs', -1):3d} SGPR {v.get('TotalSGPRs', -1):3d} "
          f"vspill {v.get('VGPRs Spill', -1):3d} sspill {v.get('SGPRs Spill', -1):3d} scratch {v.get('ScratchSize [bytes/lane]', -1)} occ {v.get('Occupancy [waves/SIMD]', -1)}")
