"""CPU oracle for the CLIP-ViP contrastive hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product path: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and there only as the checker.

* ``clipvip_oracle`` -- a functional, plain-PyTorch (CPU, fp32/fp64) restatement of
  the reference algorithm, every function citing the reference file:line it
  follows.  Pinned against the real reference (imported unmodified from
  ``/root/reference`` by ``ref_import``) through the fixtures under
  ``tests/golden/`` produced by ``tests/golden/make_golden.py``.
* ``ref_import`` -- imports the unmodified reference modules when
  ``/root/reference`` exists (build container only; never on the GPU box).
"""
