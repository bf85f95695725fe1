"""Register / scratch / occupancy table of the kernels in a hipcc -Rpass-analysis=kernel-resource-usage log: kernel_regs.py log [filter]"""
import re, sys
t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
SC = r"ScratchSize \[bytes/lane\]"; OC = r"Occupancy \[waves/SIMD\]"
for b in t.split('Function Name: ')[1:]:
    name = b.split()[0]
    if flt not in name: continue
    def g(k):
        m = re.search(k + r': (\d+)', b); return int(m.group(1)) if m else -1
    m = re.search(r'(pool_kernel|render_kernel)ILj(\d+)ELb(\d)ELi(\d+)E(?:Li(\d+)ELb(\d)|Lb(\d)ELb(\d))', name)
    m2 = re.search(r'(pool_shade_call|shade_refill_call)ILj(\d+)ELb(\d)', name)
    tag = " ".join(str(x) for x in m.groups() if x is not None) if m else (" ".join(m2.groups()) if m2 else name[:50])
    print("%-44s VGPR %4d spill %3d scratch %4d occ %d" % (tag, g('VGPRs'), g('VGPRs Spill'), g(SC), g(OC)))
