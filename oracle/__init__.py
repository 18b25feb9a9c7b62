"""CPU oracle for the RAFFT fold hot path - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (rafft_amd) never does.
"""
from .oracle import *  # noqa: F401,F403
