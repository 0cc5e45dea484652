from . import base, functional, layer, neuron, surrogate  # noqa: F401
