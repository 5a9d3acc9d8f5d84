"""Arena: one placement draw for the device workspaces of everything built inside it (no reference counterpart:
numpy / cupy allocate per call; include/rcfm.h rcfm_arena_*, profiles/r04_k_placement.md for why it matters)."""

from radiocore._internal.hip import Arena

__all__ = ["Arena"]
