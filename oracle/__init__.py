"""CPU oracle for the BISCUIT MC-dropout tile-inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU
baseline -- never as the thing shipped.  ``biscuit_amd`` must not import it.

Pinning status
--------------
* consumer side (``biscuit/threshold.py``): PINNED.  ``oracle/make_consumer_golden.py``
  imports the reference's own ``threshold.py`` / ``utils.py`` in the build
  container and captures inputs + outputs as fixtures under ``tests/golden/``.
* producer side (Slideflow / Keras arithmetic, not vendored in the reference,
  ``requirements.txt:1,5``): **PARITY UNPINNED**.  The reference holds no golden
  vector, no test and no importable implementation of the network, the MC loop
  or the normaliser, so ``oracle/xception_ref.py`` restates the published
  Keras-Xception / Slideflow algorithm from the reference's call sites
  (``biscuit/hp.py:3-23``, ``results.py:250-258``, ``biscuit/utils.py:19-28``).
"""
