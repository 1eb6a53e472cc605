"""Test helper: run the product engine on CPU memory through the C-ABI emulator (oracle/capi_emulator.py)."""
import contextlib

from mrfa_amd import hip
from oracle.capi_emulator import Emulator


@contextlib.contextmanager
def emulated_hip():
    old_lib, old_stream = hip._lib, hip.stream_ptr
    hip._lib = Emulator()
    hip.stream_ptr = lambda: 0
    try:
        yield
    finally:
        hip._lib, hip.stream_ptr = old_lib, old_stream
