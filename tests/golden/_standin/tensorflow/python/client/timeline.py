"""placeholder so `from tensorflow.python.client import timeline` resolves; never called."""
