"""CPU oracle for the plane-sweep cost-volume hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, function by function, the arithmetic of the reference's
depth-inference hot path (fdarmon/wild_deep_mvs: ``models/MVSNet``,
``models/VisMVSNet``, ``models/CVP_MVSNet``) on the CPU, in fp32, with the same
ATen primitives the reference itself calls (``grid_sample``, ``conv3d``,
``conv_transpose3d``, ``batch_norm``, ``softmax`` ...).  PyTorch is the
reference's only arithmetic dependency (``requirements.txt:5`` pins torch 1.4;
the flags used are stable through the 2.10 build in this image) and it is
present here, so no third-party algorithm has to be re-derived; the one
primitive whose semantics matter most, the zero-padded ``align_corners=True``
bilinear gather, is additionally restated from first principles in numpy
(``oracle/sampling.py``) and cross-checked against ``grid_sample``.

Rules:
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import anything from here, and only as the checker;
  * nothing in ``wild_deep_mvs_amd/`` imports it -- the product path fails
    loudly when the HIP library is missing instead of falling back;
  * parity pin: the reference ships no tests, golden vectors or fixtures
    (SURVEY.md section 4), so the oracle is pinned against outputs of the
    reference itself, generated in the build container by
    ``tests/golden/gen_golden.py`` (which imports ``/root/reference``) and
    committed as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks
    every stage boundary against them.

Every function cites the reference file:line it follows.
"""
