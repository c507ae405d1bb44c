"""TEST INFRASTRUCTURE ONLY — CPU oracle for the rocoder hot path.

`oracle.cbind` wraps the plain-C restatement (oracle/rocoder_oracle.c); `oracle.oracle_np`
is an independent numpy-f64 twin used to pin the C code and to generate golden fixtures.
Nothing under rocoder_amd/ may import this package: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg do. Parity against the real rocoder binary is UNPINNED
(see rocoder_oracle.h).
"""
