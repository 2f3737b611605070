"""CPU oracle for the masked-diffusion pretraining step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker.  The product path
(``src/`` on top of the HIP library) never routes through this package.
"""
