"""Register / spill / scratch / LDS figures of EVERY kernel instantiation of the library (hipcc -Rpass-analysis=kernel-resource-usage on each
translation unit, in parallel) as one table: python scripts/kernel_resource_table.py > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torchdriveenv_amd import build as b  # noqa: E402


def unit(src):
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + b.CFLAGS + ["-Rpass-analysis=kernel-resource-usage", "--offload-device-only", "-c", "-o", "/dev/null", src]
    return subprocess.run(cmd, capture_output=True, text=True).stderr


with ThreadPoolExecutor(8) as pool:
    text = "\n".join(pool.map(unit, b.SRC))
rows, cur = [], {}
for ln in text.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]|TotalSGPRs): (\S+)", ln)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k.split(" [")[0]] = v
assert rows, "no kernel-resource-usage remarks in the compiler's output"
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True, stdin=subprocess.DEVNULL).stdout.strip().split("\n")
print(f"# {len(rows)} kernels; hipcc flags: {' '.join(b.CFLAGS)}")
print(f"{'VGPRs':>5} {'spill':>5} {'SGPRsp':>6} {'scratch':>7} {'occ':>3} {'LDS':>6}  kernel")
for r, n in sorted(zip(rows, names), key=lambda t: t[1]):
    n = re.sub(r"\(tde_config.*", "", n).replace("void tde::", "")
    print(f"{r.get('VGPRs', '?'):>5} {r.get('VGPRs Spill', '?'):>5} {r.get('SGPRs Spill', '?'):>6} {r.get('ScratchSize', '?'):>7} {r.get('Occupancy', '?'):>3} {r.get('LDS Size', '?'):>6}  {n}")
