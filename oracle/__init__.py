"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the BasicVSR hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / reported CPU baseline -- never as the thing shipped.
"""
