"""Register / scratch / LDS usage per kernel from hipcc's assembly of one .hip file (developer tool).

  python tools/kernel_regs.py geniconet_amd/csrc/icn_kernels.hip [substring]
"""
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ''
with tempfile.NamedTemporaryFile(suffix='.s') as f:
    subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-S', '--cuda-device-only', '-o', f.name, src],
                   check=True, stderr=subprocess.DEVNULL)
    text = open(f.name).read()
meta = text[text.index('amdhsa.kernels:'):]
for block in meta.split('  - .agpr_count:')[1:]:
    block = '.agpr_count:' + block
    g = {k: v for k, v in re.findall(r'\.(\w+):\s+(\S+)', block)}
    name = subprocess.run(['c++filt', g.get('name', '?')], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'\(.*', '', name)
    if pat in name:
        print('%-40s vgpr %4s agpr %4s sgpr %4s scratch %5s lds(static) %6s' % (
            name[-40:], g.get('vgpr_count'), g.get('agpr_count'), g.get('sgpr_count'), g.get('private_segment_fixed_size'),
            g.get('group_segment_fixed_size')))
