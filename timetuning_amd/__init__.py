"""TimeTuning's training hot path on MI355X (gfx950): hand-written HIP kernels behind a C ABI (include/timetuning_hip.h) and the Python
mirror of the reference's modules."""
import os as _os

# ROCm 7.2 replays a hipGraph from AQL packets it captured when the graph was instantiated (DEBUG_CLR_GRAPH_PACKET_CAPTURE, on by default).
# TimeT's captured training step (TimeT.enable_step_graph: ~600 kernel nodes at BASELINE C2, a few memset nodes) does NOT execute
# reliably on that path: with the host more than three replays ahead of the device, or after large synchronous device-to-host copies
# between replays, a later replay computes a different step (round 6: bench.py's loss 3.56 - 3.79 by build against the eager step's
# 4.27454; tools/graph_vs_eager.py, tools/probes/graph_repeat2.py).  With the packet capture off the same graph is, bit for
# bit, the eager step in every regime measured - at the same speed (C2 7.48 ms either way).  The runtime reads the flag once, when it
# initialises (the first HIP call of the process - torch.cuda.is_available() is one): this package therefore switches it off on import,
# and TimeT.enable_step_graph() refuses to capture when somebody asked for it to stay on.
GRAPH_FLAG = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"


def _runtime_already_up() -> bool:
    """True when this process has visibly initialised the HIP runtime before this import (torch's lazy CUDA init has run): what is set
    below then comes too late for the runtime to read.  (A bare ``torch.cuda.is_available()`` initialises the runtime too and cannot be
    seen from here: import this package first - bench.py, tests/conftest.py, __graft_entry__.py and the driver do.)"""
    import sys

    t = sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:   # noqa: BLE001
        return False


_LATE = _os.environ.get(GRAPH_FLAG) != "0" and _runtime_already_up()
_os.environ.setdefault(GRAPH_FLAG, "0")


# The step's independent launch chains run on up to three HIP streams (engine.TWO_STREAMS: C2 / C3 -8 %).  HIP streams are multiplexed onto
# GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order, and two streams that land on ONE queue run in order, not beside each
# other: in a process that also holds an RCCL communicator (its streams come first) the side streams landed on the compute stream's queue
# - measured on a one-rank communicator: 8.76 ms per C2 step with the default 4, not one kernel overlapping another in the trace, against
# 7.45 with 8 (one stream: 7.91; no communicator: 7.10 either way).  Read by the runtime when it initialises, like the flag above.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def step_graph_safe() -> bool:
    """True when the process runs hipGraphs without the AQL packet capture (the only mode TimeT's step graph is verified in)."""
    return _os.environ.get(GRAPH_FLAG) == "0" and not _LATE
