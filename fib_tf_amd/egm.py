"""egm — the two-electrode electrogram recorder the reference uses to measure conduction velocity
(siravan/fib_tf `egm.py:5-12,37-47`; results in `diff_conduction_velcoty.dat`).

An "electrode" is a Gaussian-weighted mean of `image()` around a pixel.  `record()` samples two of them every
millisecond while the model runs; `delay_ms()` / `conduction_velocity()` turn the two traces into a velocity
(pixels per millisecond — the reference does not record its length unit per pixel)."""
import numpy as np


def create_mask(model, x, y, radius):
    """float32 [height, width] Gaussian exp(-(dist/radius)^2) centred at column x, row y (egm.py:5-12)"""
    cols, rows = np.meshgrid(np.arange(model.width), np.arange(model.height))
    d = np.hypot(cols - x, rows - y)
    return np.exp(-np.square(d / radius)).astype(np.float32)


def record(model, mask1, mask2, im=None, every_ms=1.0, on_tick=None):
    """run the model to its `duration`, sampling mean(image()*mask) of both electrodes every `every_ms`
    (egm.py:41-47).  `on_tick(i)` is called first on every tick (fire S2 there).  Returns float64 [n, 2]."""
    stride = max(1, int(round(every_ms / (model.dt * model.dt_per_step))))
    rows = []
    for i in model.run(im):
        if on_tick is not None:
            on_tick(i)
        if i % stride == 0:
            frame = model.image()
            rows.append([np.mean(frame * mask1), np.mean(frame * mask2)])
    return np.asarray(rows, dtype=np.float64)


def _upstroke_time(trace, level):
    """first upward crossing of `level`, linearly interpolated, in samples; None if it never happens"""
    above = trace >= level
    idx = np.flatnonzero(~above[:-1] & above[1:])
    if idx.size == 0:
        return None
    k = int(idx[0])
    return k + (level - trace[k]) / (trace[k + 1] - trace[k])


def delay_ms(traces, every_ms=1.0, frac=0.5):
    """activation delay between the two electrodes: each trace's first crossing of `frac` of its own swing"""
    t = []
    for col in (0, 1):
        x = traces[:, col]
        lo, hi = float(x.min()), float(x.max())
        at = _upstroke_time(x, lo + frac * (hi - lo))
        if at is None:
            raise ValueError('electrode %d never activated' % (col + 1))
        t.append(at)
    return (t[1] - t[0]) * every_ms


def conduction_velocity(traces, distance_px, every_ms=1.0, frac=0.5):
    """pixels per millisecond between two electrodes `distance_px` apart along the propagation direction"""
    return distance_px / delay_ms(traces, every_ms, frac)


def main():
    """the reference's own protocol (egm.py:15-50): Beeler-Reuter 512x512, obstacle, S2, electrodes 30 px apart"""
    from .br import BeelerReuter
    from .screen import Screen
    config = {'width': 512, 'height': 512, 'dt': 0.1, 'dt_per_plot': 10, 'diff': 1.0, 'duration': 3000,
              'skip': False, 'cheby': True, 'timeline': False, 'timeline_name': 'timeline_br.json',
              'save_graph': False}
    model = BeelerReuter(config)
    model.add_hole_to_phase_field(150, 256, 50)
    model.define()
    model.add_pace_op('s2', 'luq', 10.0)
    im = Screen(model.height, model.width, 'Beeler-Reuter Model')
    s2 = model.millisecond_to_step(300)
    m1, m2 = create_mask(model, 300 + 15, 256, 5), create_mask(model, 300 - 15, 256, 5)
    out = record(model, m1, m2, im, on_tick=lambda i: model.fire_op('s2') if i == s2 else None)
    np.savetxt('test.dat', out)


if __name__ == '__main__':
    main()
