"""Minimal stand-in for the `gym` package (absent in this image).

Only used by tools/make_golden.py so that /root/reference can be imported in
the build container to produce golden vectors.  Not part of the product.
"""


class Env:
    metadata = {}

    def step(self, action):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError
