"""mrfa_amd: MI355X-native (gfx950) implementation of the MRFA dense-motion + refinement + generator hot path."""
import os as _os

# hipGraph replays (mrfa_amd/graph.py): ROCm 7.2's "graph packet capture" fast path does not reliably order memcpy / memset
# nodes against kernel nodes on replay (wrong results from the second replay on; see the module docstring of graph.py).
# The flag is read when the HIP runtime initialises, i.e. at the first device call -- import mrfa_amd before that.
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

__version__ = "0.1.0"
