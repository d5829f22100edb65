"""CPU oracle for the NequIP message-passing hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this package; nothing under ``nequip_amd/`` does.  It restates, with plain PyTorch CPU ops in the
reference's own op order (materialised gather -> one einsum chain per path -> cat -> scatter_add_ ->
autograd forces), the path that ``BASELINE.json:north_star`` names.  Each function cites the reference
file:line (mir-group/nequip v0.19.0 under /root/reference) or the SURVEY.md Appendix-A item it follows.

PARITY UNPINNED: the arithmetic of this path lives in ``e3nn>=0.6.0,<0.7.0`` (``pyproject.toml:22``),
an un-vendored dependency that is absent from /root/reference and not installable here, and the
reference ships no golden vectors / known-answer values for it.  The e3nn semantics
(real Wigner-3j construction, spherical-harmonic convention, path/linear normalisation, Gate constants)
are restated from its published algorithm; they are validated by algebraic identities, against
``sympy.physics`` Clebsch-Gordan values, and by the reference's own property suite (equivariance,
finite-difference forces, cutoff smoothness) in ``tests/`` -- but not against e3nn itself.  A global
sign per path or a normalisation constant that differed from e3nn would be invisible to those checks.
"""
