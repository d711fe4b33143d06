"""ctypes binding of libdctr_hip.so (C-ABI: include/dctr.h) and the autograd glue around it.

``lib``  -- loader / struct mirrors / error handling         ``plan`` -- feature columns -> dctr_plan_t
``ops``  -- torch.autograd.Function wrappers
"""
