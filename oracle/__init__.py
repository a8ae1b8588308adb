"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference hot path used as checkers.  Nothing under
``pathnet_amd/`` may import this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do.
"""
