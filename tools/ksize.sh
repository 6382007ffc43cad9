#!/bin/bash
# code size / registers / scratch of the device kernels matching a pattern (CPU-only: hipcc cross-compiles)   usage: tools/ksize.sh 'msm|fold'
[ -n "$KSIZE_REUSE" ] || (cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -c -Rpass-analysis=kernel-resource-usage /root/repo/ark_bulletproofs_amd/csrc/arkbp.hip -o /tmp/arkbp_dev.o 2> /tmp/res.txt)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=/tmp/arkbp_dev.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/arkbp_gfx.co
/opt/rocm/lib/llvm/bin/llvm-readelf -s --wide /tmp/arkbp_gfx.co | grep FUNC | awk '{print $3, $8}' | sort -u > /tmp/ksizes.txt
python3 - "$1" <<'PY'
import re,sys,subprocess
pat=re.compile(sys.argv[1])
sizes={l.split()[1]:int(l.split()[0]) for l in open('/tmp/ksizes.txt') if len(l.split())==2}
txt=open('/tmp/res.txt').read()
seen=set()
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name=b.split()[0]
    if name in seen: continue
    seen.add(name)
    dem=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip().split('(')[0]
    if not pat.search(dem): continue
    g=lambda k: re.search(k+r": (\d+)", b).group(1)
    sc, oc, ld = g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')
    print("%-58s code %6.1f KB  VGPR %3s  scratch %4s  occ %s  LDS %s" % (dem[:58], sizes.get(name, 0) / 1024, g('VGPRs'), sc, oc, ld))
PY
