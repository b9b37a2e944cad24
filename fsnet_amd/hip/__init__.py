"""ctypes binding of libfsnet_hip.so (the C ABI declared in include/fsnet_hip.h)."""
from .lib import lib, load_library, stream_ptr, FsError, check  # noqa: F401
