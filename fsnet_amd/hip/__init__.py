"""ctypes binding of libfsnet_hip.so (the C ABI declared in include/fsnet_hip.h)."""
from .binding import lib, load_library, stream_ptr, FsError, check  # noqa: F401
