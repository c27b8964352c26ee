"""development: per-kernel register / scratch budget from the metadata of a device assembly listing
(hipcc ... -S --cuda-device-only -o x.s fsim.hip; see spills.sh for the flags).  usage: resources.py x.s [substring ...]"""
import re
import subprocess
import sys

t = open(sys.argv[1]).read()
md = t[t.index('amdhsa.kernels:'):]
want = sys.argv[2:] or ['env_step', 'shadow']
for k in md.split('  - .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', k).group(1)
    g = lambda f: re.search(r'\.%s:\s+(\S+)' % f, k).group(1)
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().replace('void ', '')
    if any(w in dn for w in want):
        print('%-150s vgpr %s sgpr %s scratch %s B spills v/s %s/%s lds(static) %s' % (dn[:150], g('vgpr_count'), g('sgpr_count'), g('private_segment_fixed_size'),
                                                                                   g('vgpr_spill_count'), g('sgpr_spill_count'), g('group_segment_fixed_size')))
