"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the MAFED per-step hot path.

Nothing under ``mafed_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and only as the
checker / reported CPU baseline -- never as the thing shipped or measured as the product.

Parity status: PINNED by golden vectors generated in the authoring container from the
reference's own classes (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``).  The reference
repository itself holds no tests or fixtures for this path.
"""
