"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement (plain PyTorch fp32, no functorch) of the reference's volumetric-rendering
train-step hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package; the product package never does (it fails loudly when the HIP
library is missing instead of falling back to this code).

Pinning: the reference ships no tests or golden vectors for this path.  The oracle
is pinned by outputs of the reference itself, imported and run in the build container by
``tests/golden/gen_golden.py``, which asserts oracle == reference and writes the committed
``tests/golden/*.npz`` fixtures.
"""
from . import ref_cpu  # noqa: F401
