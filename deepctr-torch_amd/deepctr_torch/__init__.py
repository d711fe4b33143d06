"""deepctr_torch -- MI355X-native drop-in for the embedding + feature-interaction hot path of
DeepCTR-Torch (reference v0.2.9).

Put ``<repo>/deepctr-torch_amd`` on ``sys.path`` and existing user code
(``from deepctr_torch.inputs import SparseFeat``; ``from deepctr_torch.models import DeepFM``)
runs on hand-written gfx950 kernels behind ``libdctr_hip.so`` (C-ABI in ``include/dctr.h``).
Unlike the reference (``deepctr_torch/__init__.py:5``, ``utils.py:19-44``) importing this package
starts no thread and makes no network request.
"""
__version__ = "0.2.9+mi355x.r1"
