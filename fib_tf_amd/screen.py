"""Headless stand-in for the reference's SDL2 `Screen` (siravan/fib_tf `screen.py:58-374`): the object
`IonicModel.run(im)` paints a frame into every `dt_per_plot` sub-steps (`ionic.py:206-215`).  No window is
opened; frames are kept (optionally written as 8-bit greyscale PNGs with the standard library only), so
that drivers written against the reference — `im.imshow(image)`, `im.wait()`, `im.save(name)` — run on a
GPU box without a display."""
import struct
import zlib

import numpy as np


def write_png_grey(path, img):
    """[H, W] array in 0..1 -> 8-bit greyscale PNG"""
    a = (np.clip(np.asarray(img, np.float32), 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
    h, w = a.shape
    raw = b''.join(b'\x00' + a[r].tobytes() for r in range(h))

    def chunk(tag, data):
        c = struct.pack('>I', len(data)) + tag + data
        return c + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)

    with open(path, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 0, 0, 0, 0)) +
                chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


class Screen:
    def __init__(self, height, width, title='fib_tf_amd', keep=0, png_pattern=None):
        """keep: number of most recent frames retained in `self.frames` (0 = only the last one);
        png_pattern: e.g. 'frame_%05d.png' to write every frame"""
        self.height, self.width, self.title = height, width, title
        self.keep, self.png_pattern = keep, png_pattern
        self.frames, self.count, self.last = [], 0, None
        self.texts = []

    def imshow(self, image):
        """accepts what the reference's imshow accepts: a [H, W] float image in 0..1 (screen.py:255-289)"""
        img = np.array(image, dtype=np.float32)
        assert img.shape == (self.height, self.width), (img.shape, (self.height, self.width))
        self.last = img
        if self.keep:
            self.frames.append(img)
            del self.frames[:-self.keep]
        if self.png_pattern:
            write_png_grey(self.png_pattern % self.count, img)
        self.count += 1

    def draw_text(self, text, x=0, y=0, **kw):
        self.texts.append((x, y, text))

    def plot(self, *a, **kw):
        pass

    def peek(self):
        return False                                # no pending window events

    def wait(self):
        pass                                        # nothing to block on without a window

    def save(self, name):
        if self.last is not None:
            write_png_grey(name, self.last)

    def destroy(self):
        pass                                        # no window to close (screen.py:376-380)

    def __bool__(self):
        return True
