#!/usr/bin/env python3
"""kernel_resources.py LIB.so [substring ...] — VGPR / SGPR / LDS / spill figures of the gfx950 kernels inside a HIP
shared library (reads the offload bundle out of .hip_fatbin, then the code object's metadata note with llvm-readelf)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def code_objects(path):
    data = open(path, 'rb').read()
    pos = 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return
        n, = struct.unpack_from('<Q', data, i + 24)
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if 'gfx' in triple and size:
                yield triple, data[i + off:i + off + size]
        pos = i + 24


def kernels(path):
    for triple, blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix='.co', delete=False) as f:
            f.write(blob)
        try:
            txt = subprocess.run([READELF, '--notes', f.name], capture_output=True, text=True).stdout
        finally:
            os.unlink(f.name)
        for e in txt.split('- .agpr_count')[1:]:
            g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, e) or [None, '?'])[1]
            yield {'name': g('name'), 'vgpr': g('vgpr_count'), 'sgpr': g('sgpr_count'), 'lds': g('group_segment_fixed_size'),
                   'spill': g('vgpr_spill_count'), 'scratch': g('private_segment_fixed_size'), 'agpr': (re.match(r':\s+(\d+)', e) or [None, '?'])[1]}


def demangle(names):
    try:
        out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout
        return out.splitlines()
    except OSError:
        return names


if __name__ == '__main__':
    ks = list(kernels(sys.argv[1]))
    names = demangle([k['name'] for k in ks])
    for k, n in zip(ks, names):
        n = re.sub(r'\(.*$', '', n).replace('void fib::', '')
        if all(s in n for s in sys.argv[2:]):
            print('%-100s vgpr %3s agpr %s sgpr %3s lds %6s spill %s scratch %s' % (n, k['vgpr'], k['agpr'], k['sgpr'], k['lds'], k['spill'], k['scratch']))
