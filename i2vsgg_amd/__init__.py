"""i2vsgg_amd -- MI355X-native hot path of Ego-J/I2VSGG.

Import side effect (before the HIP runtime initialises, which happens at the first HIP call of the process):
``DEBUG_CLR_GRAPH_PACKET_CAPTURE=0``.  ROCm 7.2's runtime replays HIP graphs through pre-built AQL packet batches; on the
legacy default stream that path loses the order between a graph's nodes and the stream's other work while a second
stream is busy (DESIGN.md section 5 has the bisect and the A/B runs; measured cost of the generic path on the 330-node
step graph: none).  A value the user has set is left alone, and so is a process whose HIP runtime is already up.
Whether the runtime READ the value cannot be known from here (``torch.cuda.is_available()`` initialises it without torch
saying so), so ``train.replay_graph`` does not rely on it: a replay asked for on the default stream always runs on a private
stream between two event edges."""
import os
import sys


def _hip_already_up():
    t = sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:
        return False


if "DEBUG_CLR_GRAPH_PACKET_CAPTURE" not in os.environ and not _hip_already_up():
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
