"""ctypes binding of libolympic_hip.so (the HIP kernels behind include/olympic_hip.h).

There is NO CPU fallback: if the shared library is missing or the machine has no gfx950
device, importing works but the first use raises OlyError - loudly.
"""
import ctypes as C
import os

from . import _abi

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("OLYMPIC_HIP_LIB") or os.path.join(_PKG_DIR, "lib", "libolympic_hip.so")


class OlyError(RuntimeError):
    pass


_lib = None
_ROCTX = os.environ.get("OLY_ROCTX", "0") == "1"


def lib():
    """Load the HIP library once; bind every symbol of the header."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OlyError(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950). olympic_hip has no CPU fallback.")
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:  # e.g. libamdhip64 not found
            raise OlyError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in _abi.SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        if int(L.oly_abi_version()) != _abi.ABI_VERSION:
            raise OlyError(f"{LIB_PATH} exports ABI version {int(L.oly_abi_version())}, this binding mirrors version "
                           f"{_abi.ABI_VERSION} of include/olympic_hip.h: rebuild the library")
        _lib = L
    return _lib


def check(ctx, rc, what):
    if rc != 0:
        L = lib()
        detail = L.oly_last_error(ctx).decode() if ctx else ""
        raise OlyError(f"{what}: {L.oly_strerror(rc).decode()} ({rc}) {detail}")


def ptr(t):
    """Device (or host) address of a torch tensor / numpy array / None."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


class Context:
    """Owns one oly_ctx (one per process and device)."""

    def __init__(self, device=None):
        import torch
        if not torch.cuda.is_available():
            raise OlyError("no HIP device visible: olympic_hip needs an MI355X (gfx950); "
                           "there is no CPU fallback")
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", int(device) if not isinstance(device, torch.device)
                                   else (device.index or 0))
        self._h = C.c_void_p()
        check(None, lib().oly_create(C.byref(self._h), self.device.index), "oly_create")
        self._keep = []

    @property
    def handle(self):
        return self._h

    def stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def call(self, name, *args):
        if _ROCTX:                                   # OLY_ROCTX=1: a roctx range per entry point (rocprofv3 --marker-trace)
            import torch
            torch.cuda.nvtx.range_push(name)
            try:
                rc = getattr(lib(), name)(self._h, *args)
            finally:
                torch.cuda.nvtx.range_pop()
        else:
            rc = getattr(lib(), name)(self._h, *args)
        check(self._h, rc, name)

    def close(self):
        if self._h:
            lib().oly_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipTimer:
    """HIP-event timer on an explicit stream (torch.cuda.Event only sees torch's stream)."""

    def __init__(self):
        self.a, self.b = C.c_void_p(), C.c_void_p()
        L = lib()
        check(None, L.oly_event_create(C.byref(self.a)), "oly_event_create")
        check(None, L.oly_event_create(C.byref(self.b)), "oly_event_create")

    def start(self, stream):
        check(None, lib().oly_event_record(self.a, stream), "oly_event_record")

    def stop(self, stream):
        check(None, lib().oly_event_record(self.b, stream), "oly_event_record")

    def elapsed_ms(self):
        L = lib()
        check(None, L.oly_event_sync(self.b), "oly_event_sync")
        ms = C.c_float()
        check(None, L.oly_event_elapsed_ms(self.a, self.b, C.byref(ms)), "oly_event_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            lib().oly_event_destroy(self.a)
            lib().oly_event_destroy(self.b)
        except Exception:
            pass
