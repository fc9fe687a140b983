"""
Stand-in for the third-party ``biotite`` package (absent from this image, no network).

OUR code, not reference code: it provides exactly the handful of names that
springcraft's hot path touches, so that the reference can be
*imported in the build container* by ``oracle/make_golden.py`` to generate golden vectors.
It is never imported by the product package, by tests, or on the GPU box.
"""
