"""-m gpu: what surrounds the multi-tick launches (csrc/fibhip.hip) — a launch that gives up must not cost the run, a caller
that declares its series gets it launched whole, and nothing runs ahead of a caller that can write the state itself.

The reference has none of this (ionic.py:202-204: one synchronous sess.run per tick), so the yardstick is the library's own
one-launch-per-tick mode (FIBHIP_MT=0): every observation must match it bit for bit."""
import ctypes as C
import os
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _state(H, W, seed, nvar=4):
    rng = np.random.default_rng(seed)
    init = np.empty((nvar, H, W), np.float32)
    init[0] = rng.uniform(-0.02, 1.0, (H, W))
    for v in range(1, nvar):
        init[v] = rng.uniform(0, 1, (H, W))
    return init, rng.uniform(0.3, 1.0, (H, W)).astype(np.float32)


def _play(_lib, H, W, script, variant, monkeypatch, env):
    """runs `script` — ints: step(n); ('x', n): n single-tick calls; 'get': read the potential back; 'all': the whole state;
    'pace'; 'sync'; ('expect', n) — and returns (observations, stats, fallbacks, warnings)"""
    for k in ('FIBHIP_MT', 'FIBHIP_MT_FAKE_GIVEUP', 'FIBHIP_AHEAD', 'FIBHIP_MT_WAIT_MS'):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    if variant:
        monkeypatch.setenv('FIBHIP_VARIANT', variant)
    else:
        monkeypatch.delenv('FIBHIP_VARIANT', raising=False)
    init, phi = _state(H, W, 11 * H + W)
    st = _lib.Stepper(_lib.FENTON4V, H, W, 0.1, 1.3, flags=_lib.FAST)
    st.set_phase(phi)
    st.set_state(-1, init)
    out = []
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        for op in script:
            if isinstance(op, int):
                st.step(op)
            elif op == 'get':
                out.append(st.get_state(0).copy())
            elif op == 'all':
                out.append(st.get_state(-1))
            elif op == 'pace':
                st.pace(H // 4, H // 4 + 5, W // 3, W // 3 + 6, 1.0, 0.0)
            elif op == 'sync':
                st.sync()
            elif op[0] == 'x':
                for _ in range(op[1]):
                    st.step(1)
            elif op[0] == 'expect':
                st.expect(op[1])
        out.append(st.get_state(-1))
    stats, fb, tpl = st.launch_stats(), st.fallbacks(), st.ticks_per_launch()
    st.close()
    return out, stats, fb, tpl, [w for w in caught if issubclass(w.category, RuntimeWarning)]


SCRIPT = [1, 40, 'get', ('x', 10), 'get', ('x', 10), 'get', ('x', 10), 'get', 'pace', 33, ('x', 7), 'sync', ('x', 20), 'sync',
          ('x', 20), 'all', 70, 'get', ('x', 10), 'get', ('x', 10)]


@pytest.mark.parametrize('H,W,variant', [(96, 100, '10,44,25,-3'), (512, 512, None)])
@pytest.mark.parametrize('nth', [1, 2, 3, 4, 5, 6, 7, 8, 9, 11])
def test_a_launch_that_gives_up_does_not_cost_the_run(gpu_lib, monkeypatch, H, W, variant, nth):
    """FIBHIP_MT_FAKE_GIVEUP=n: the n-th multi-tick launch of the handle finds the give-up word raised in its name (what its
    tiles would write after waiting out their bound) and leaves without results, like every launch queued behind it.  The
    host finds the word at its next synchronisation, goes back to the state THAT launch started from (intact: a launch
    writes the other slab only), switches multi-tick launches off and recomputes the lost ticks one launch per tick.  Ordinary
    launches, a whole series launched at its first tick, and a run-ahead that carries a read-back are all hit by some `nth`;
    every observation must equal the one-launch-per-tick run's, and the caller is told once (RuntimeWarning)."""
    want, _, fb0, _, w0 = _play(gpu_lib, H, W, SCRIPT, variant, monkeypatch, {'FIBHIP_MT': '0'})
    assert fb0 == (0, 0) and not w0
    got, stats, fb, tpl, warned = _play(gpu_lib, H, W, SCRIPT, variant, monkeypatch, {'FIBHIP_MT_FAKE_GIVEUP': str(nth)})
    assert len(got) == len(want)
    for i, (x, y) in enumerate(zip(got, want)):
        assert np.isfinite(x).all()
        assert np.array_equal(x, y), 'observation %d differs after the recovery (max |d| %.3g)' % (i, float(np.abs(x - y).max()))
    assert fb[0] == 1, fb                                 # exactly one launch gave up; none after it (the mode is off)
    assert tpl == 1                                       # ... for good
    assert len(warned) == 1 and 'gave up' in str(warned[0].message)
    assert stats['gave_up_recovered'] == 1 and stats['ticks_recomputed_after_give_up'] == fb[1]


@pytest.mark.parametrize('nth', [2, 5, 7])
def test_beeler_reuter_launch_that_gives_up(gpu_lib, monkeypatch, nth):
    """the same with eight arrays (two 16-byte exchange cells per grid cell, the paired LDS image of two-row strips)"""
    H, W = 70, 130
    rng = np.random.default_rng(9)
    init = np.empty((8, H, W), np.float32)
    init[0] = rng.uniform(-85, 20, (H, W))
    init[1] = rng.uniform(1e-7, 1e-5, (H, W))
    for v in range(2, 8):
        init[v] = rng.uniform(0.01, 0.99, (H, W))

    def play(env):
        for k in ('FIBHIP_MT', 'FIBHIP_MT_FAKE_GIVEUP', 'FIBHIP_AHEAD'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        monkeypatch.setenv('FIBHIP_VARIANT', '5,54,21,-2')
        st = gpu_lib.Stepper(gpu_lib.BR, H, W, 0.1, 0.809, flags=gpu_lib.FAST)       # (direct gates: no table to set)
        st.set_state(-1, init)
        out = []
        with warnings.catch_warnings(record=True):
            warnings.simplefilter('always')
            for op in (1, 40, 'get', ('x', 10), 'get', ('x', 10), 'get', ('x', 10), 'get', 33, 'sync', ('x', 12)):
                if isinstance(op, int):
                    st.step(op)
                elif op == 'get':
                    out.append(st.get_state(0).copy())
                elif op == 'sync':
                    st.sync()
                else:
                    for _ in range(op[1]):
                        st.step(1)
            out.append(st.get_state(-1))
        fb, tpl = st.fallbacks(), st.ticks_per_launch()
        st.close()
        return out, fb, tpl

    want, fb0, _ = play({'FIBHIP_MT': '0'})
    got, fb, tpl = play({'FIBHIP_MT_FAKE_GIVEUP': str(nth)})
    assert fb0 == (0, 0) and fb[0] == 1 and tpl == 1
    for i, (x, y) in enumerate(zip(got, want)):
        assert np.array_equal(x, y), 'observation %d differs after the recovery' % i


@pytest.mark.parametrize('seed', range(1, 1 + int(os.environ.get('FIBTF_STRESS_SEEDS', '8'))))      # (a few hundred: a stress run)
def test_give_up_anywhere_in_random_call_sequences(gpu_lib, monkeypatch, seed):
    """whatever the caller does around the launch that gives up — single ticks, long calls, paces, probes, host writes, read-backs
    of one array or all, syncs, declared series, a new phase field — every observation equals the one-launch-per-tick run's"""
    # (stress runs: FIBTF_STRESS_GRID=512 = the benchmark's own tiling, 252 workgroups on 256 compute units; FIBTF_STRESS_MODEL=br)
    H, W = (83, 120) if not os.environ.get('FIBTF_STRESS_GRID') else (int(os.environ['FIBTF_STRESS_GRID']),) * 2
    br = os.environ.get('FIBTF_STRESS_MODEL') == 'br'
    init, phi = _state(H, W, 200 + seed, 8 if br else 4)
    if br:
        init[0] = init[0] * 100.0 - 85.0
        init[1] *= 1e-5
        init[2:] = init[2:] * 0.98 + 0.01
    nvar = 8 if br else 4

    def play(env):
        for k in ('FIBHIP_MT', 'FIBHIP_MT_FAKE_GIVEUP', 'FIBHIP_AHEAD'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        if H == 83:
            monkeypatch.setenv('FIBHIP_VARIANT', '5,54,21,-2' if br else '10,44,25,-3')
        else:
            monkeypatch.delenv('FIBHIP_VARIANT', raising=False)
        rng = np.random.default_rng(seed)
        st = gpu_lib.Stepper(gpu_lib.BR if br else gpu_lib.FENTON4V, H, W, 0.1, 0.809 if br else 1.3, flags=gpu_lib.FAST)
        st.set_phase(phi)
        st.set_state(-1, init)
        seen = []
        with warnings.catch_warnings(record=True):
            warnings.simplefilter('always')
            for _ in range(120):
                op = rng.choice(['step1', 'step1', 'step1', 'step1', 'stepn', 'series', 'pace', 'probe', 'get1', 'getall', 'set1', 'sync',
                                 'expect', 'phase'])
                if op == 'step1':
                    st.step(1)
                elif op == 'stepn':
                    st.step(int(rng.integers(0, 70)))
                elif op == 'series':                       # the reference driver's pattern: n ticks, a frame, n ticks, a frame
                    n = int(rng.integers(2, 12))
                    for _ in range(3):
                        for _ in range(n):
                            st.step(1)
                        seen.append(st.get_state(0).copy())
                elif op == 'pace':
                    r0, c0 = int(rng.integers(0, H - 4)), int(rng.integers(0, W - 4))
                    st.pace(r0, r0 + 4, c0, c0 + 4, 10.0 if br else 1.0, -90.0 if br else 0.0)
                elif op == 'probe':
                    seen.append(np.float32(st.probe(int(rng.integers(0, nvar)), int(rng.integers(0, H)), int(rng.integers(0, W)))))
                elif op == 'get1':
                    seen.append(st.get_state(int(rng.integers(0, nvar))).copy())
                elif op == 'getall':
                    seen.append(st.get_state(-1))
                elif op == 'set1':
                    v = int(rng.integers(2, nvar))
                    st.set_state(v, (st.get_state(v) * np.float32(0.999)).astype(np.float32))
                elif op == 'sync':
                    st.sync()
                elif op == 'expect':
                    st.expect(int(rng.integers(0, 40)))     # kept or broken, as the following calls happen to fall
                else:
                    st.set_phase(phi if rng.integers(0, 2) else None)
            seen.append(st.get_state(-1))
        fb, stats = st.fallbacks(), st.launch_stats()
        st.close()
        return seen, fb, stats

    want, fb0, _ = play({'FIBHIP_MT': '0'})
    untouched, fbu, free = play({})                        # (also: how many multi-tick launches this sequence has)
    assert fb0 == (0, 0) and fbu == (0, 0) and free['mt_launches'] >= 3
    for i, (x, y) in enumerate(zip(untouched, want)):      # declared series kept and broken, run-ahead, stops: nothing shows
        assert np.array_equal(x, y), 'observation %d of the untouched run differs' % i
    nth = 1 + (seed * 7 + int(os.environ.get('FIBTF_STRESS_SALT', '0'))) % max(1, int(free['mt_launches']))
    got, fb, _ = play({'FIBHIP_MT_FAKE_GIVEUP': str(nth)})
    assert fb[0] == 1, (nth, fb)
    assert len(got) == len(want)
    for i, (x, y) in enumerate(zip(got, want)):
        assert np.array_equal(x, y), 'observation %d differs (launch %d of %d gave up)' % (i, nth, free['mt_launches'])


def test_real_give_up_under_co_tenancy(gpu_lib, monkeypatch):
    """The REAL thing, with a short leash: a 512x512 handle (252 tiles that must all be resident) with a wait bound of 1 ms steps
    while another handle of this process keeps every compute unit busy with long one-launch-per-tick kernels on its own stream
    (2048x2048, not a multi-tick grid).  Tiles that become resident early wait for neighbours that are not yet: some launches give
    up for real — the kernel's own time-out, give-up word and way out — and the handle recovers; whether and how often depends
    on the dispatcher, the results must not.  (Every wait is bounded: 1 ms here, so nothing can hang.)"""
    for k in ('FIBHIP_MT', 'FIBHIP_MT_FAKE_GIVEUP', 'FIBHIP_AHEAD', 'FIBHIP_VARIANT'):
        monkeypatch.delenv(k, raising=False)
    H = W = 512
    init, phi = _state(H, W, 77)
    big, bphi = _state(2048, 2048, 78)

    def play(mt, crowd):
        if mt:
            monkeypatch.delenv('FIBHIP_MT', raising=False)
            monkeypatch.setenv('FIBHIP_MT_WAIT_MS', '1')
        else:
            monkeypatch.setenv('FIBHIP_MT', '0')
        st = gpu_lib.Stepper(gpu_lib.FENTON4V, H, W, 0.1, 1.3, flags=gpu_lib.FAST)
        st.set_phase(phi)
        st.set_state(-1, init)
        st.step(1)
        st.sync()
        other = None
        if crowd:
            other = gpu_lib.Stepper(gpu_lib.FENTON4V, 2048, 2048, 0.1, 1.3, flags=gpu_lib.FAST)
            other.set_state(-1, big)
            other.step(1)
            other.sync()
        out = []
        with warnings.catch_warnings(record=True):
            warnings.simplefilter('always')
            for rep in range(6):
                if other is not None:
                    other.step(40)                         # ~4 ms of kernels that fill the device, enqueued and left running
                st.step(64)
                out.append(st.get_state(0).copy())
                for _ in range(10):
                    st.step(1)
                out.append(st.get_state(-1))
        fb = st.fallbacks()
        if other is not None:
            other.sync()
            other.close()
        st.close()
        return out, fb

    want, _ = play(False, False)
    got, fb = play(True, True)
    print('real give-ups under co-tenancy: %d launch(es) gave up, %d tick(s) recomputed' % fb)
    for i, (x, y) in enumerate(zip(got, want)):
        assert np.isfinite(x).all()
        assert np.array_equal(x, y), 'observation %d differs (%d launches gave up)' % (i, fb[0])
    assert fb[0] in (0, 1)                                 # (the mode is off for the handle after the first)


def test_two_processes_share_the_device(gpu_lib):
    """two processes, each with a 512x512 multi-tick grid and a wait bound of 1 ms, on the one GPU: when their launches start
    together neither grid is fully resident and tiles give up FOR REAL (no test switch: the kernel's time-out, its give-up word,
    the word in host memory, the journal with several unconfirmed launches).  Whether it happens in a given run is the
    dispatcher's business (profiles/r04_two_processes_giveup.txt: it did in every run recorded); both processes must end with
    the bits of the one-launch-per-tick run either way (tools/dbg/two_processes_giveup.py checks both and says what happened)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith('FIBHIP_')}
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'dbg', 'two_processes_giveup.py')], capture_output=True, text=True,
                       timeout=300, env=env)
    print(r.stdout)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert r.stdout.count('equal to one launch per tick: True') == 2, r.stdout


def test_untouched_run_has_no_fallback(gpu_lib, monkeypatch):
    got, stats, fb, tpl, warned = _play(gpu_lib, 96, 100, SCRIPT, '10,44,25,-3', monkeypatch, {})
    want, *_ = _play(gpu_lib, 96, 100, SCRIPT, '10,44,25,-3', monkeypatch, {'FIBHIP_MT': '0'})
    assert fb == (0, 0) and tpl > 1 and not warned and stats['mt_launches'] > 0
    for x, y in zip(got, want):
        assert np.array_equal(x, y)


def test_wait_bound_is_a_property_of_the_handle(gpu_lib, monkeypatch):
    """FIBHIP_MT_WAIT_MS / fibhip_set_mt_wait_ms: the bound travels with every launch (packed into a word the kernel has
    anyway); a run under a 50 ms bound is an ordinary run"""
    got, _, fb, tpl, _ = _play(gpu_lib, 96, 100, [1, 40, 'get', ('x', 25)], '10,44,25,-3', monkeypatch, {'FIBHIP_MT_WAIT_MS': '50'})
    want, *_ = _play(gpu_lib, 96, 100, [1, 40, 'get', ('x', 25)], '10,44,25,-3', monkeypatch, {'FIBHIP_MT': '0'})
    assert fb == (0, 0) and tpl > 1
    for x, y in zip(got, want):
        assert np.array_equal(x, y)
    st = gpu_lib.Stepper(gpu_lib.FENTON4V, 64, 64, 0.1, 1.3, flags=gpu_lib.FAST)
    st.set_mt_wait_ms(10)
    with pytest.raises(gpu_lib.FibhipError):
        st.set_mt_wait_ms(-1)
    st.close()


def test_no_multi_tick_launches_under_a_cu_mask(gpu_lib, monkeypatch):
    """a process-wide CU mask takes compute units away that the device still reports: the tiles of a grid 'that fits' would not
    all be resident, so such a process never starts a multi-tick launch (it would give up after its bound and fall back)"""
    monkeypatch.setenv('FIBHIP_VARIANT', '10,44,25,-3')
    monkeypatch.delenv('FIBHIP_MT', raising=False)
    init, phi = _state(96, 100, 3)
    for var in ('HSA_CU_MASK', 'ROC_GLOBAL_CU_MASK'):
        monkeypatch.setenv(var, '0:0-127')            # (read by this library at create; the runtime of this process is up already)
        st = gpu_lib.Stepper(gpu_lib.FENTON4V, 96, 100, 0.1, 1.3, flags=gpu_lib.FAST)
        st.set_state(-1, init)
        st.step(3)
        assert st.ticks_per_launch() == 1 and st.launch_stats()['mt_launches'] == 0
        st.close()
        monkeypatch.delenv(var)
    st = gpu_lib.Stepper(gpu_lib.FENTON4V, 96, 100, 0.1, 1.3, flags=gpu_lib.FAST)
    assert st.ticks_per_launch() > 1
    st.close()


@pytest.mark.parametrize('H,W,variant', [(96, 100, '10,44,25,-3'), (512, 512, None)])
def test_a_declared_series_is_one_launch_from_the_first_time(gpu_lib, monkeypatch, H, W, variant):
    """fibhip_expect(n): the caller says how many ticks it will ask for before it looks again (IonicModel.run() does, from its
    loop bounds and frame period) — the series is ONE launch, issued at its first tick, the very first time; nothing is learnt
    from the call history.  A caller that breaks its word is stopped where it is; all observations as with one launch per tick."""
    for k in ('FIBHIP_MT', 'FIBHIP_MT_FAKE_GIVEUP', 'FIBHIP_AHEAD'):
        monkeypatch.delenv(k, raising=False)
    if variant:
        monkeypatch.setenv('FIBHIP_VARIANT', variant)
    else:
        monkeypatch.delenv('FIBHIP_VARIANT', raising=False)
    init, phi = _state(H, W, 5)
    st = gpu_lib.Stepper(gpu_lib.FENTON4V, H, W, 0.1, 1.3, flags=gpu_lib.FAST)
    st.set_phase(phi)
    st.set_state(-1, init)
    st.step(1)                                            # (plan selection happens on the first tick)
    st.sync()
    for n in (20, 7, 32):                                 # three series of lengths never seen before, each declared
        st.expect(n)
        st.time_begin()
        for _ in range(n):
            st.step(1)
        ms, launches = st.time_end()
        assert launches == 1, (n, launches)
    # a declared series longer than one launch can be: 32 ticks at a time, each launch issued when its first tick is asked for
    st.expect(70)
    st.time_begin()
    for _ in range(70):
        st.step(1)
    ms, launches = st.time_end()
    assert launches == 3, launches
    # a broken word: 20 declared, the caller looks after 5
    st.expect(20)
    for _ in range(5):
        st.step(1)
    a = st.get_state(-1)
    stats = st.launch_stats()
    assert stats['ticks'] == 1 + 20 + 7 + 32 + 70 + 5, stats
    st.close()
    monkeypatch.setenv('FIBHIP_MT', '0')
    ref = gpu_lib.Stepper(gpu_lib.FENTON4V, H, W, 0.1, 1.3, flags=gpu_lib.FAST)
    ref.set_phase(phi)
    ref.set_state(-1, init)
    ref.step(1 + 20 + 7 + 32 + 70 + 5)
    b = ref.get_state(-1)
    ref.close()
    assert np.array_equal(a, b)


def test_run_declares_its_frame_period(gpu_lib, monkeypatch):
    """IonicModel.run(im) reads a frame back every dt_per_plot sub-steps (ionic.py:206) and says so before each frame: from
    the FIRST frame on the next series is launched before the frame is waited for (no two equal series to learn from), and the
    frames are the frames of a run without any run-ahead"""
    from fib_tf_amd.fenton import Fenton4v
    from fib_tf_amd.screen import Screen

    def run(env):
        for k in ('FIBHIP_MT', 'FIBHIP_AHEAD'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m = Fenton4v({'width': 512, 'height': 512, 'dt': 0.1, 'dt_per_plot': 100, 'diff': 1.5, 'duration': 64.05, 'skip': False,
                      'cheby': True})
        m.add_hole_to_phase_field(256, 256, 30)
        m.define()
        im = Screen(512, 512, 'test')
        frames = []
        im.imshow = lambda image: frames.append(np.array(image, copy=True))
        for _ in m.run(im, block=False):
            pass
        stats = m._stepper.launch_stats()
        final = np.stack([m._State[n].eval() for n in m.VAR_NAMES])
        return frames, stats, final

    f0, s0, x0 = run({'FIBHIP_AHEAD': '0', 'FIBHIP_MT': '0'})
    f1, s1, x1 = run({})
    assert len(f0) == len(f1) == 7                         # frames after ticks 0, 10, ..., 60
    for a, b in zip(f0, f1):
        assert np.array_equal(a, b)
    assert np.array_equal(x0, x1)
    # 64 ticks: the first tick on its own, then the frame's run-ahead launches of 10 ticks each, then the last 3
    assert s1['mt_launches'] <= 8 and s1['mt_ticks'] >= 60, s1


def _hip():
    """the HIP runtime libfibhip.so itself runs on (a second copy of the runtime in this process — torch bundles one — would
    see no device): the libamdhip64 already mapped into the process"""
    with open('/proc/self/maps') as f:
        paths = sorted({line.split()[-1] for line in f if 'libamdhip64' in line})
    hip = None
    for path in paths:                                    # (an earlier test may have imported torch: its copy sees no device here)
        cand, n = C.CDLL(path), C.c_int(0)
        if cand.hipGetDeviceCount(C.byref(n)) == 0 and n.value > 0:
            hip = cand
            break
    assert hip is not None, paths
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    return hip


def test_nothing_runs_ahead_of_a_caller_that_can_write_the_state(gpu_lib, monkeypatch):
    """fibhip_state_ptr hands out a raw device pointer 'the caller may write through at any time'.  A launch that ran ahead of
    the caller would read the state BEFORE such a write (or race it): once the pointer is out, nothing runs ahead any more —
    series are launched when their ticks have been asked for — and a write between a read-back and the next tick counts."""
    for k in ('FIBHIP_MT', 'FIBHIP_AHEAD'):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv('FIBHIP_VARIANT', '10,44,25,-3')
    H, W = 96, 100
    init, phi = _state(H, W, 21)
    hip = _hip()
    patch = np.full((W,), 0.75, np.float32)

    def play(expose, mt):
        if mt:
            monkeypatch.delenv('FIBHIP_MT', raising=False)
        else:
            monkeypatch.setenv('FIBHIP_MT', '0')
        st = gpu_lib.Stepper(gpu_lib.FENTON4V, H, W, 0.1, 1.3, flags=gpu_lib.FAST)
        st.set_phase(phi)
        st.set_state(-1, init)
        seen, launches_at_get = [], []
        for rep in range(6):
            st.step(10)                                    # (one call: launched at once, nothing is pending at the read-back)
            l0 = st.launch_stats()['launches']
            seen.append(st.get_state(0).copy())
            launches_at_get.append(st.launch_stats()['launches'] - l0)
            if expose and rep >= 1:
                # the caller writes row 40 of the potential itself, through the raw pointer, between the frame and the next tick
                _, ptr = st.state_buf(0)
                assert hip.hipMemcpy(ptr + 40 * W * 4, patch.ctypes.data, W * 4, 1) == 0
            elif rep >= 1:
                u = st.get_state(0).copy()
                u[40] = patch
                st.set_state(0, u)
        seen.append(st.get_state(-1))
        st.close()
        return seen, launches_at_get

    a, la = play(True, True)
    b, _ = play(False, False)                              # the same writes through set_state, one launch per tick
    for i, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x, y), 'observation %d: a write through the raw pointer was lost or raced' % i
    # after the pointer was handed out (behind the second frame) no read-back launches anything any more
    assert la[2:] == [0, 0, 0, 0], la
    # (the control: the same calls with set_state instead of the raw pointer DO run ahead — the read-back launches the next series)
    _, lc = play(False, True)
    assert lc[0] == 0 and sum(lc[1:]) >= 3, lc


def test_no_run_ahead_on_caller_owned_slabs(gpu_lib, monkeypatch):
    """a handle on caller-owned slabs (fibhip_desc.ext_slab: device memory the caller can write between two calls, as
    fib_tf_amd/sharded.py's torch tensors are) never gets a launch ahead of its calls: a read-back after equal series
    launches nothing"""
    for k in ('FIBHIP_MT', 'FIBHIP_AHEAD'):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv('FIBHIP_VARIANT', '10,44,25,-3')
    H, W = 96, 100
    init, phi = _state(H, W, 22)
    hip = _hip()
    slabs = [C.c_void_p(), C.c_void_p()]
    for p in slabs:
        assert hip.hipMalloc(C.byref(p), 4 * H * W * 4) == 0
        assert hip.hipMemset(p, 0, 4 * H * W * 4) == 0
    st = gpu_lib.Stepper(gpu_lib.FENTON4V, H, W, 0.1, 1.3, flags=gpu_lib.FAST, ext_slabs=(slabs[0].value, slabs[1].value))
    st.set_phase(phi)
    st.set_state(-1, init)
    def series(st):
        at_get = []
        for rep in range(5):
            st.step(10)
            l0 = st.launch_stats()['launches']
            st.get_state(0)
            at_get.append(st.launch_stats()['launches'] - l0)
        return at_get
    at_get = series(st)
    a = st.get_state(-1)
    assert st.ticks_per_launch() > 1                       # several ticks per launch: yes; ahead of the caller: never
    st.close()
    for p in slabs:
        hip.hipFree(p)
    assert at_get == [0] * 5, at_get
    own = gpu_lib.Stepper(gpu_lib.FENTON4V, H, W, 0.1, 1.3, flags=gpu_lib.FAST)     # the control: library-owned slabs do run ahead
    own.set_phase(phi)
    own.set_state(-1, init)
    ctl = series(own)
    b = own.get_state(-1)
    own.close()
    assert ctl[0] == 0 and sum(ctl[1:]) >= 3, ctl
    assert np.array_equal(a, b)
    monkeypatch.setenv('FIBHIP_MT', '0')
    ref = gpu_lib.Stepper(gpu_lib.FENTON4V, H, W, 0.1, 1.3, flags=gpu_lib.FAST)
    ref.set_phase(phi)
    ref.set_state(-1, init)
    ref.step(50)
    assert np.array_equal(a, ref.get_state(-1))
    ref.close()
