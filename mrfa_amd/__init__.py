"""mrfa_amd: MI355X-native (gfx950) implementation of the MRFA dense-motion + refinement + generator hot path."""
__version__ = "0.1.0"
