"""CPU oracle for the SMART exact-matching hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (smart_amd) never does.
"""
