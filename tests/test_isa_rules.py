"""Rules the built library's machine code must keep (no GPU needed: the gfx950 code objects are carved out of libicn.so and disassembled).

Rule 1 (round 6, profiles/r06_stem_wgrad_race.txt): no packed fp32 instruction whose op_sel routes the HIGH dword of a source pair to the
LOW lane (e.g. `v_pk_fma_f32 ... op_sel:[0,1,0]`, what hipcc makes of `float4 += float4 * scalar`).  Inside the power-limited training
step, beside the bf16x3 weight-gradient kernel of the second stream, exactly those lanes of k_stem_wgrad differed from run to run.
csrc/Makefile therefore builds icn_kernels.hip and icn_loss.hip without packed fp32; this test covers the whole library, so that an
edit to any other file that makes the compiler emit such an instruction is caught here and not by a flaky bit-identity test."""
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'geniconet_amd', 'libicn.so')
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def code_objects(path):
    """The gfx950 code objects (ELF images) of every offload bundle embedded in the shared library."""
    blob = open(path, 'rb').read()
    out, pos = [], 0
    while True:
        at = blob.find(MAGIC, pos)
        if at < 0:
            break
        n, = struct.unpack_from('<Q', blob, at + len(MAGIC))
        q = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from('<QQQ', blob, q)
            triple = blob[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if 'gfx950' in triple and size:
                out.append(blob[at + off:at + off + size])
        pos = at + len(MAGIC)
    return out


@pytest.mark.skipif(not os.path.exists(LIB), reason='libicn.so not built')
@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason='llvm-objdump not found')
def test_no_packed_fp32_instruction_routes_a_high_dword_to_the_low_lane(tmp_path):
    objs = code_objects(LIB)
    assert len(objs) >= 4, 'expected one gfx950 code object per .hip source, found %d' % len(objs)
    bad, n_insts, kernels = [], 0, 0
    for i, obj in enumerate(objs):
        f = tmp_path / ('co%d.elf' % i)
        f.write_bytes(obj)
        asm = subprocess.run([OBJDUMP, '-d', '--mcpu=gfx950', str(f)], check=True, capture_output=True, text=True).stdout
        cur = None
        for line in asm.splitlines():
            m = re.match(r'^[0-9a-f]+ <(\S+)>:', line)
            if m:
                cur = m.group(1)
                kernels += 1
                continue
            n_insts += 1
            if re.search(r'\bv_pk_(fma|mul|add|mov)_(f32|b32)\b', line) and re.search(r'op_sel:\[[0-9,]*1', line):
                bad.append((cur, line.strip().split('//')[0].strip()))
    assert n_insts > 100000 and kernels > 50, (n_insts, kernels)     # the disassembly really happened
    assert not bad, 'packed fp32 instructions with a cross-dword op_sel (rule 1):\n' + '\n'.join('%s: %s' % b for b in bad[:20])
