"""agplace_amd -- MI355X-native aerial-ground embedding + retrieval hot path of AGPlace.

Python host side mirroring the reference's module interfaces (same class names, constructor
arguments, attribute names, state_dict keys, forward signatures) above the C ABI of
libagplace_hip.so (include/agplace_hip.h).  See DESIGN.md.
"""
from .options import Options, get_options, set_options, from_reference_opt  # noqa: F401

__version__ = "0.1.0"
