"""CPU oracle for the MPN Groth16 hot path - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product path (bazuka_amd/, libbzk.so) must never import, link or call it.
"""
