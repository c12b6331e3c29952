"""Shared by the A/B tools: the caller-owned K-split workspace of the persistent GEMMs (ABI 7: tt_linear_ksplit_workspace_bytes / _init)."""
import ctypes as C

import torch


def ksplit_ws(lib, stream):
    """Allocates and initialises one workspace for `lib` on the current device; returns (tensor, data_ptr, bytes)."""
    lib.tt_linear_ksplit_workspace_bytes.restype = C.c_size_t
    lib.tt_linear_ksplit_workspace_bytes.argtypes = []
    lib.tt_linear_ksplit_workspace_init.restype = C.c_int
    lib.tt_linear_ksplit_workspace_init.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    nb = int(lib.tt_linear_ksplit_workspace_bytes())
    t = torch.empty(nb, dtype=torch.uint8, device="cuda")
    assert lib.tt_linear_ksplit_workspace_init(t.data_ptr(), nb, stream) == 0
    return t, t.data_ptr(), nb
