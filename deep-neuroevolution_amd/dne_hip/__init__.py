"""dne_hip -- MI355X-native drop-in for the hot path of es_distributed (uber-research/deep-neuroevolution).

Host-side mirror of the reference's Python seams (run_master / run_worker, SharedNoiseTable,
policies.Policy.rollout) over the C ABI of libdne_hip.so (include/dne_hip.h).  No CPU fallback.
"""
from . import _lib  # noqa: F401
