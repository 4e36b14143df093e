"""placeholder for the reference's SDL2 viewer module (imported by court.py, never used by the golden generator)."""


class Screen:
    pass
