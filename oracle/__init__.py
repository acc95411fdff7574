"""CPU restatement of the reference's reconstruction path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
nothing under dryv_amd/ does. PARITY UNPINNED (see dryv_oracle.c header and DESIGN.md).
"""
from .oracle import build, load, reconstruct, decode_mb, residual4x4, residual8x8, get_qpc, deblock  # noqa: F401
