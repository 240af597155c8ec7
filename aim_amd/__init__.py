"""aim_amd -- MI355X-native drop-in for AIM's per-pair alignment path (NW / SWG / WFA / WFA-adaptive).

Only the hot path lives here: csrc/ (hand-written gfx950 HIP kernels + the C-ABI of include/aim_hip.h),
host/ (the C host program that keeps the reference CLI), capi.py / engine.py (Python mirror of the
reference's host-side interface) and launch.py (the run-*-pim-*.py counterpart).
"""
__all__ = ["capi", "engine", "build", "launch"]
