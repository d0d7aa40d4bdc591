"""registers / scratch / LDS / occupancy of every kernel of one source, as the compiler reports them:
    python tools/diag/kernel_resources.py conv_fast.hip [extra hipcc flags]  ->  one line per kernel"""
import re
import subprocess
import sys

CSRC = "/root/repo/self-paced-contrastive-learning_amd/csrc/"
src = sys.argv[1]
pre = "14" if src == "supcon.hip" else "16"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-mllvm",
       f"-amdgpu-kernarg-preload-count={pre}", "-Rpass-analysis=kernel-resource-usage", "-x", "hip", "-c", CSRC + src, "-o",
       "/tmp/kres.o"] + sys.argv[2:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
    for k in ("VGPRs:", "AGPRs", "ScratchSize", "Occupancy", "LDS Size", "SGPRs:"):
        m = re.search(re.escape(k) + r"\D*(\d+)", line)
        if m and cur:
            rows[cur][k.strip(":")] = int(m.group(1))
names = subprocess.run(["c++filt"] + list(rows), capture_output=True, text=True).stdout.splitlines()
for (k, v), n in sorted(zip(rows.items(), names), key=lambda t: t[1]):
    n = n.replace("void spcl::", "").split("(")[0]
    print(n, " ".join(f"{a}={b}" for a, b in v.items()))
