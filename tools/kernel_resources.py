"""Per-kernel register / scratch / LDS usage from a `hipcc ... -Rpass-analysis=kernel-resource-usage 2> log` build log.
usage: python tools/kernel_resources.py <log> [name-filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r'remark: Function Name: ', txt)[1:]:
    name = b.split()[0]
    if flt not in name:
        continue
    def g(k):
        m = re.search(k + r': (\d+)', b)
        return m.group(1) if m else '?'
    try:
        dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    except FileNotFoundError:
        dn = name
    dn = re.sub(r'\(.*', '', dn)
    print(f"{dn[:70]:70s} SGPR {g('TotalSGPRs'):>3} VGPR {g('VGPRs'):>3} AGPR {g('AGPRs'):>3} scratch {g(r'ScratchSize .bytes/lane.'):>4} "
          f"occ {g(r'Occupancy .waves/SIMD.')} LDS {g(r'LDS Size .bytes/block.')}")
