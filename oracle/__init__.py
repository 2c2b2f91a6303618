"""CPU oracle for the CMDIAD hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package, and only as the checker / reported baseline.  The product (cmdiad_amd/) never
imports it and has no CPU fallback.
"""
