"""playcube — replay a `cube.npy` written by the example drivers (`np.save('cube', cube)`, frames of
`image()*phase`; siravan/fib_tf `fenton.py:179-187`, `playcube.py:1-15`).

The reference loops the frames in an SDL2 window until a key is pressed.  The headless Screen has no
keyboard, so this plays `loops` passes (default 1) and can write every frame as a PNG:

    python -m fib_tf_amd.playcube cube.npy --png frames/f%04d.png"""
import argparse
from time import sleep

import numpy as np

from .screen import Screen


def play(cube, title='reentry!', loops=1, delay=0.025, png_pattern=None, screen=None):
    """show every frame of `cube` ([n, h, w], values 0..1) `loops` times; returns the Screen"""
    x = np.load(cube) if isinstance(cube, str) else np.asarray(cube)
    if x.ndim != 3:
        raise ValueError('cube must be [frames, height, width], got shape %s' % (x.shape,))
    n, h, w = x.shape
    sc = screen if screen is not None else Screen(h, w, title, png_pattern=png_pattern)
    i = 0
    while i < n * loops and not sc.peek():
        sc.imshow(x[i % n, :, :])
        if delay:
            sleep(delay)
        i += 1
    return sc


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('cube', nargs='?', default='cube.npy')
    ap.add_argument('--loops', type=int, default=1)
    ap.add_argument('--delay', type=float, default=0.025)
    ap.add_argument('--png', default=None, help='printf-style pattern: write every shown frame as a PNG')
    a = ap.parse_args(argv)
    sc = play(a.cube, loops=a.loops, delay=a.delay, png_pattern=a.png)
    print('%d frames shown' % sc.count)
    sc.destroy()


if __name__ == '__main__':
    main()
