"""CPU oracle of the SVTyper likelihood hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  ``svtyper_amd`` (the product) never does.

Two restatements of the same reference lines:

* ``oracle.c_oracle``  -- ctypes front-end of ``svt_oracle.c`` (plain C, fast enough for
  million-unit batches and for the ``cpu_baseline`` timing).
* ``oracle.py_oracle`` -- pure-Python loops, statement-for-statement after the reference,
  for small cases; it cross-checks the C file and anchors the golden vectors.

Parity status: pinned against golden vectors generated from the imported reference
(tests/golden/make_golden.py); see tests/test_oracle_golden.py.
"""
