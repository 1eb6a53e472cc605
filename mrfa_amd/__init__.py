"""mrfa_amd: MI355X-native (gfx950) implementation of the MRFA dense-motion + refinement + generator hot path."""
import os as _os
import sys as _sys

# hipGraph replays (mrfa_amd/graph.py): ROCm 7.2's "graph packet capture" fast path does not reliably order memcpy / memset
# nodes against kernel nodes on replay (wrong results from the second replay on; see the module docstring of graph.py).
# The flag is read when the HIP runtime initialises, i.e. at the first device call -- import mrfa_amd before that.
_FLAG = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
_flag_before = _os.environ.get(_FLAG)
_torch = _sys.modules.get("torch")
_hip_was_up = bool(_torch is not None and getattr(_torch, "cuda", None) is not None and _torch.cuda.is_initialized())
_os.environ.setdefault(_FLAG, "0")
# True: the HIP runtime was already initialised when this package was imported and the flag was not "0" at that time, so the
# setdefault above came too late.  mrfa_amd.graph refuses to capture in that state (graph_replay_safe()) instead of replaying
# graphs whose memcpy / memset nodes may run out of order.
GRAPH_REPLAY_UNSAFE = _hip_was_up and _flag_before != "0"
if GRAPH_REPLAY_UNSAFE:
    import warnings as _warnings
    _warnings.warn(f"mrfa_amd was imported after the HIP runtime had initialised with {_FLAG}={_flag_before!r}: hipGraph capture "
                   f"(GraphedTrainStep / GraphedForward / Animator(graph=True)) is disabled; export {_FLAG}=0 before the first "
                   f"device call, or import mrfa_amd before touching the GPU", RuntimeWarning, stacklevel=2)


def graph_replay_safe(what: str = "hipGraph capture") -> None:
    """Raise unless graph replays can be trusted in this process: the flag must have been "0" when the HIP runtime started (see
    above), and must still be "0" (a caller that reset it before the runtime initialised is caught here too)."""
    if GRAPH_REPLAY_UNSAFE or _os.environ.get(_FLAG) != "0":
        raise RuntimeError(f"{what}: {_FLAG} was {(_flag_before if GRAPH_REPLAY_UNSAFE else _os.environ.get(_FLAG))!r} when the HIP "
                           f"runtime initialised; on ROCm 7.2 replayed graphs then mis-order memcpy / memset nodes (results are wrong "
                           f"from the second replay on).  Export {_FLAG}=0 before the first device call or import mrfa_amd first; "
                           f"eager launches (no graph) are unaffected.")


__version__ = "0.2.0"
