"""CPU oracle — TEST INFRASTRUCTURE, NOT PRODUCT.  PARITY UNPINNED (see cpu_ref.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
