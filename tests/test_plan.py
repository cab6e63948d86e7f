"""CPU-side checks of the launch-plan boundary (include/pmn_hip.h: pmn_plan_*): recording touches no device, so the host logic --
what gets recorded, by which thread, in which order, the error behaviour -- runs here without a GPU.  The replay side is
tests/test_plan_gpu.py."""
import ctypes
import threading

import pytest
import torch


def _plan(L):
    p = ctypes.c_void_p()
    assert L.pmn_plan_create(ctypes.byref(p)) == 0 and p.value
    return p


def test_recording_appends_launches_instead_of_launching():
    from patchmatchnet_amd import _lib
    L = _lib.lib()
    p = _plan(L)
    assert L.pmn_plan_launch(p, None) == -1  # not recorded yet
    assert L.pmn_plan_begin(p) == 0
    assert L.pmn_plan_begin(p) == -1  # one recording per thread at a time
    # fake device addresses: nothing is dereferenced or launched while recording (there is no GPU here)
    assert L.pmn_nchw_to_nhwc(0x1000, 0x2000, 1, 4, 4, 4, None) == 0
    assert L.pmn_nchw_to_nhwc(None, 0x2000, 1, 4, 4, 4, None) == -1  # argument checks as usual; a refused call records nothing
    assert L.pmn_normalize_depth(0x1000, 0x2000, 0x3000, 2, 100, 0x4000, None) == 0
    assert L.pmn_confidence(0x1000, 1, 8, 4, 4, 8, 8, 0x2000, None, None) == 0
    assert L.pmn_plan_end(p) == 0
    assert L.pmn_plan_end(p) == -1 and L.pmn_plan_begin(p) == -1  # a plan is recorded once
    assert L.pmn_plan_count(p) == 3
    names = [L.pmn_plan_kernel_name(p, i).decode() for i in range(3)]
    assert "nchw_to_nhwc_kernel" in names[0] and "normalize_depth_kernel" in names[1] and "confidence" in names[2], names
    assert L.pmn_plan_kernel_name(p, 3) is None
    assert L.pmn_plan_launch(p, None) == -3  # no device here: the launch itself fails, loudly
    assert L.pmn_plan_destroy(p) == 0
    assert L.pmn_plan_count(None) == -1 and L.pmn_plan_destroy(None) == -1


def test_recording_is_per_thread():
    """Only the thread between pmn_plan_begin and pmn_plan_end records; eval.py's other threads keep launching for real (here: they
    fail for want of a device instead of being appended)."""
    from patchmatchnet_amd import _lib
    L = _lib.lib()
    p = _plan(L)
    assert L.pmn_plan_begin(p) == 0
    seen = {}

    def other():
        seen["rc"] = L.pmn_nchw_to_nhwc(0x1000, 0x2000, 1, 4, 4, 4, None)
        q = _plan(L)
        seen["own"] = (L.pmn_plan_begin(q), L.pmn_nchw_to_nhwc(0x1000, 0x2000, 1, 4, 4, 4, None), L.pmn_plan_end(q), L.pmn_plan_count(q))
        L.pmn_plan_destroy(q)

    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert L.pmn_plan_end(p) == 0
    assert L.pmn_plan_count(p) == 0  # the other thread's call was not recorded into this plan ...
    assert seen["rc"] == -3            # ... it tried to launch (and there is no device)
    assert seen["own"] == (0, 0, 0, 1)  # while that thread's own recording worked
    L.pmn_plan_destroy(p)


def test_recording_pass_refuses_operators_outside_the_library():
    """PlannedForward's guard: while a forward is recorded, ATen operators that launch kernels raise (they would run once, on
    uninitialised data, and be missing from every replay); views and allocations pass."""
    from patchmatchnet_amd import PmnError
    from patchmatchnet_amd.graph import _LibraryLaunchesOnly
    x = torch.arange(24, dtype=torch.float32).reshape(2, 3, 4)
    with _LibraryLaunchesOnly():
        y = x.permute(0, 2, 1)[:, 1:].unsqueeze(0).view(1, 2, 3, 3)
        e = torch.empty((2, 3), dtype=torch.float32)
        z = torch.empty_like(x)
        assert x.contiguous() is x and x.float() is x and x.detach().shape == x.shape
        for bad in (lambda: x + 1, lambda: y.contiguous(), lambda: x.double(), lambda: torch.zeros(3), lambda: z.copy_(x),
                    lambda: torch.cat([x, x]), lambda: x / x):
            with pytest.raises(PmnError, match="outside"):
                bad()
    assert (x + 1).sum() > 0 and e.shape == (2, 3)  # outside the mode everything is as usual
