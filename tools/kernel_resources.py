"""Tabulate what hipcc's -Rpass-analysis=kernel-resource-usage reports for every kernel of libxmaps_hip.so:
  hipcc ... -Rpass-analysis=kernel-resource-usage 2> res.txt ; python tools/kernel_resources.py res.txt [filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
seen = set()
for b in re.split(r'remark: Function Name: ', txt)[1:]:
    name = b.split()[0]
    if name in seen:
        continue
    seen.add(name)
    g = lambda k: re.search(k + r': (\d+)', b).group(1)
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r'\(.*', '', dn)[:80]
    if flt and flt not in dn:
        continue
    print("%-82s S%4s V%4s occ%2s sspill%4s vspill%3s scratch%s" % (dn, g('TotalSGPRs'), g('VGPRs'), g(r'Occupancy \[waves/SIMD\]'),
          g('SGPRs Spill'), g('VGPRs Spill'), g(r'ScratchSize \[bytes/lane\]')))
