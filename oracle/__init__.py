"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the openai/blocksparse BlocksparseMatMul hot path.

Nothing under ``oracle/`` is imported by the product package ``blocksparse_amd``.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it, and only as the
checker / the timed CPU baseline -- never as a compute fallback.
"""
