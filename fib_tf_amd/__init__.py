"""fib_tf_amd — MI355X-native explicit time-stepper for 2D cardiac reaction-diffusion with the
config-dict / IonicModel / define() / run() API of siravan/fib_tf.  Python host code over a ctypes
C ABI (include/fibhip.h) into hand-written HIP kernels for gfx950; no TensorFlow, no CPU fallback.

    from fib_tf_amd.fenton import Fenton4v
    from fib_tf_amd.br import BeelerReuter
    from fib_tf_amd.court import Courtemanche
"""
from . import _lib                              # noqa: F401
from .ionic import IonicModel                   # noqa: F401
from .fenton import Fenton4v                    # noqa: F401
from .br import BeelerReuter                    # noqa: F401
from .court import Courtemanche                 # noqa: F401

__all__ = ['IonicModel', 'Fenton4v', 'BeelerReuter', 'Courtemanche']
