"""CPU oracle package -- TEST INFRASTRUCTURE ONLY.

Nothing under ``mageslam_amd/`` may import this package; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do (as the checker).
"""
