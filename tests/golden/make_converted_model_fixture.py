#!/usr/bin/env python3
"""A model that the REFERENCE's builder produced and the REFERENCE's `modify` converted with `enable_NequipAMD_full`, saved so
that the GPU box (which has no reference tree) can evaluate it: `tests/golden/converted_reference_model.pt`.

    python tests/golden/make_converted_model_fixture.py      # needs /root/reference

The build and the conversion are exactly those of tests/test_reference_full_modifier.py (nequip's real `NequIPGNNModel`
builder and `modify`, e3nn served by functional stand-ins).  The converted chain consists of nequip_amd modules only; the two
reference CONTAINERS around it (`SequentialGraphNetwork`, `GraphModel`: no arithmetic) cannot be unpickled without nequip,
so the chain's modules -- the very objects the modifier produced, not rebuilt ones -- are re-seated in this package's
containers before saving.  tests/test_converted_reference_model.py (GPU) compares its energy and forces with the oracle.
"""
import os
import sys

sys.dont_write_bytecode = True  # (the reference tree is read-only: no __pycache__ under /root/reference)
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..")))


def main():
    import pytest  # noqa: F401  (the fixture generator of the test module is reused as is)
    import test_reference_full_modifier as T

    gen = T.ref.__wrapped__() if hasattr(T.ref, "__wrapped__") else T.ref.__pytest_wrapped__.obj()
    ref = next(gen)
    try:
        from nequip_amd.integrations import nequip_full
        from nequip_amd.nn import ForceStressOutput, GraphModel, SequentialGraphNetwork

        model = T._build_reference(ref)
        nequip_full.register_full()
        torch.cuda.is_available = lambda: True
        torch.version.hip = torch.version.hip or "7.0"
        converted = ref["modify_utils"].modify(model, [{"modifier": nequip_full.FULL_MODIFIER_NAME}])
        chain = converted.model.func
        assert all(type(m).__module__.startswith("nequip_amd.") for m in chain.children())
        seated = SequentialGraphNetwork(dict(chain.named_children()))
        out = GraphModel(ForceStressOutput(seated, converted.model.do_derivatives), type_names=T.HYPER["type_names"],
                         model_dtype=torch.float32, r_max=T.HYPER["r_max"])
        path = os.path.join(HERE, "converted_reference_model.pt")
        torch.save({"model": out, "hyper": T.HYPER}, path)
        print("wrote", path, os.path.getsize(path), "bytes")
    finally:
        try:
            next(gen)
        except StopIteration:
            pass


if __name__ == "__main__":
    main()
