"""not-gpu: the CPU restatement (oracle/fib_oracle.c) under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY 5:
sanitizers run on the CPU build; the GPU pool offers none).  `make -C oracle asan` builds the instrumented library; a
child process preloads the sanitizer runtime, loads that library and runs the unit ops, the single steps, a 64 x 64
trajectory against the golden vectors and every model on odd grids (3 x 5, 37 x 53, 5 x 3, 4 x 64)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_ubsan():
    import oracle
    libasan = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip('no libasan on this host')
    subprocess.check_call(['make', '-C', os.path.dirname(oracle.SO), 'asan'], stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=libasan, FIB_ORACLE_LIB=oracle.SO_ASAN, OMP_NUM_THREADS='2',
               ASAN_OPTIONS='detect_leaks=0:abort_on_error=1:halt_on_error=1', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'asan_worker.py')], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'all checks passed' in r.stdout
    assert 'runtime error' not in r.stderr and 'AddressSanitizer' not in r.stderr, r.stderr[-4000:]
