"""shineon-virtual-tryon_amd: the ShineOn try-on hot path (WarpModel + UnetMaskModel) on MI355X.

Import as `shineon_virtual_tryon_amd` (the repo-root shim maps the hyphenated directory name).
Nothing here falls back to a CPU implementation: without libshineon_hip.so every operator raises.
"""
from ._lib import LIB_PATH, HipKernelError, HipLibraryMissing, lib  # noqa: F401

__all__ = ["lib", "LIB_PATH", "HipKernelError", "HipLibraryMissing"]
