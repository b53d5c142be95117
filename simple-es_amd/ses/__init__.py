"""ses -- Python host layer over libses_hip.so (the gfx950 hot path of simple-es)."""
from ._lib import (ENV_CARTPOLE, ENV_LUNARLANDER, ENV_NONE, ENV_SIMPLE_SPREAD, HIDDEN, MODE_EPISODIC, MODE_FIXED_LENGTH, SesError, load)  # noqa: F401
from .device import HipES, destroy_stream, exclusive_stream, param_count  # noqa: F401
