export TMPDIR=/tmp
out=$PWD/gpurun_out/r03/sigsegv3
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/maps -- python3 tools/dbg/sas_variants.py maps > $out/maps.out 2> $out/maps.err
echo "rc=$?"
python3 - <<'PY'
import re, subprocess, os
err = open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out/r03/sigsegv3/maps.err')).read()
maps = []
for l in err.splitlines():
    m = re.match(r'([0-9a-f]+)-([0-9a-f]+) r-xp ([0-9a-f]+) \S+ \S+\s+(\S+)', l)
    if m:
        maps.append((int(m.group(1), 16), int(m.group(2), 16), int(m.group(3), 16), m.group(4)))
frames = [int(x, 16) for x in re.findall(r'@\s+(0x[0-9a-f]+)', err.split('SIGSEGV')[-1])][:14]
syms = {}
for a in frames:
    for lo, hi, off, path in maps:
        if lo <= a < hi:
            rel = a - lo + off
            if path not in syms:
                try:
                    out = subprocess.run(['nm', '-D', '--defined-only', '-n', path], capture_output=True, text=True).stdout
                    syms[path] = [(int(x.split()[0], 16), x.split()[-1]) for x in out.splitlines() if len(x.split()) == 3 and x.split()[1] in 'TtWw']
                except Exception:
                    syms[path] = []
            best = None
            for addr, name in syms[path]:
                if addr <= rel:
                    best = (addr, name)
                else:
                    break
            name = best[1] if best else '?'
            try:
                name = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()[:150]
            except Exception:
                pass
            print('%#x  %s + %#x  nearest exported: %s%+d' % (a, os.path.basename(path), rel, name, rel - best[0] if best else 0))
            break
    else:
        print('%#x  (no mapping)' % a)
PY
