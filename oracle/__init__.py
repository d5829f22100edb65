"""CPU oracle for the NequIP message-passing hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this package; nothing under ``nequip_amd/`` does.  It restates, with plain PyTorch CPU ops in the
reference's own op order (materialised gather -> one einsum chain per path -> cat -> scatter_add_ ->
autograd forces), the path that ``BASELINE.json:north_star`` names.  Each function cites the reference
file:line (mir-group/nequip v0.19.0 under /root/reference) or the SURVEY.md Appendix-A item it follows.

PARITY: PARTLY PINNED.
* Pinned against the reference's own code: the rows nequip implements itself in plain PyTorch -- edge vectors
  (``nn/utils.py:68-118``), length normaliser + Bessel x polynomial cutoff (``nn/embedding/_edge.py:18-151``,
  ``cutoffs.py:17-27``), ``ScalarMLPFunction`` (``nn/mlp.py:81-268``), ``AvgNumNeighborsNorm``, ``PerTypeScaleShift``,
  ``AtomwiseReduce``, ``ForceStressOutput`` (forces / virial / stress).  ``tests/golden/make_reference_golden.py``
  imports those modules from /root/reference (with inert stand-ins for the uninstalled e3nn / training stack, which
  they never call at run time) and commits their inputs, outputs and gradients as ``tests/golden/ref_*.npz``;
  ``tests/test_reference_golden.py`` checks this oracle, the host mirrors and the HIP kernels against them.
* UNPINNED: everything computed BY ``e3nn>=0.6.0,<0.7.0`` (``pyproject.toml:22``) -- real spherical harmonics, the
  Clebsch-Gordan ``uvu`` tensor product, ``o3.Linear``, ``FullyConnectedTensorProduct``, ``Gate``.  e3nn is an
  un-vendored dependency, absent from /root/reference and not installable here, and the reference ships no golden
  vectors / known-answer values for it.  Its semantics (real Wigner-3j construction, spherical-harmonic
  convention, path/linear normalisation, Gate constants) are restated from the published algorithm and validated by
  algebraic identities, against ``sympy.physics`` Clebsch-Gordan values and by the reference's own property suite
  (equivariance, finite-difference forces, cutoff smoothness) in ``tests/`` -- but not against e3nn itself.  A global
  sign per path or a normalisation constant that differed from e3nn would be invisible to those checks.
"""
