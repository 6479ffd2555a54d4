"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference hot path.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package.  PARITY UNPINNED.
"""
