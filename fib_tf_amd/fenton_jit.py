"""fenton_jit — the reference's `fenton_jit.py` is `fenton_simple.py` with `solve` wrapped in an XLA JIT scope
(`fenton_jit.py:128-135`) and another trace-file name; the arithmetic is identical, and so is the kernel here."""
from .fenton_simple import Fenton4vSimple


class Fenton4vJIT(Fenton4vSimple):
    def __init__(self, props):
        super().__init__(props)
        if 'timeline_name' not in props:
            self.timeline_name = 'timeline_jit.json'
