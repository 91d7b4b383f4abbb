"""ctypes binding of libmi_ddpm.so (the C ABI in include/mi_ddpm.h) and tensor-level wrappers."""
from .lib import load_library, library_path  # noqa: F401
