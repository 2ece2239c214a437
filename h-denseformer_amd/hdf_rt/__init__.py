"""Runtime of the MI355X-native H-DenseFormer hot path (ctypes over libhdf_hip.so)."""
from ._lib import BF16, F32, EXPORTS, HdfError, LIB_PATH, lib  # noqa: F401
