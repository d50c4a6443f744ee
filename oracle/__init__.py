"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference's algorithms for the hot path, used as the checker in tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product (sgrl_amd/) never imports this package.
"""
