"""CPU-side checks of the C-ABI library: it builds, loads and exports every declared entry point."""
import ctypes
import os

from bagel_amd import _lib


def test_library_exports_every_declared_symbol():
    _lib.build()
    assert os.path.exists(_lib.LIB_PATH)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _lib.declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/bagel_hip.h but not exported"
    assert lib.bagel_abi_version() == 1


def test_argument_validation_needs_no_gpu():
    """Entry points validate shapes before touching the device, so errors are reportable on a CPU box."""
    lib = _lib.lib()
    rc = lib.bagel_gemm_bf16(None, ctypes.c_long(0), None, ctypes.c_long(0), None, ctypes.c_long(0), 1, 1, 1, None, None,
                             ctypes.c_long(0), 0, None)
    assert rc != 0 and b"null operand" in lib.bagel_last_error()
