"""package marker for the stand-in (see ../__init__.py)"""
