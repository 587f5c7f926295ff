"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage remarks (stderr of a compile) per kernel instantiation:
    hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip -o x.o 2> x.rem;  python tools/kernel_resources.py x.rem"""
import re
import subprocess
import sys


def parse(path):
    name, d = None, {}
    for line in open(path):
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            name = m.group(1)
            d[name] = {}
            continue
        m = re.search(r'remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)', line)
        if m and name:
            d[name][m.group(1)] = int(m.group(2))
    return d


if __name__ == '__main__':
    d = parse(sys.argv[1])
    names = list(d)
    dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
    for k, dn in zip(names, dem):
        v = d[k]
        dn = re.sub(r'^void ', '', dn)
        dn = re.sub(r'snsde_mfma::', '', dn)
        print(f"{dn[:110]:110s} vgpr {v.get('VGPRs')} agpr {v.get('AGPRs')} scratch {v.get('ScratchSize [bytes/lane]')} occ {v.get('Occupancy [waves/SIMD]')}")
