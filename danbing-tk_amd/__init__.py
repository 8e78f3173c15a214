"""danbing-tk_amd — MI355X-native `danbing-tk align` hot path.

Import with importlib.import_module("danbing-tk_amd") (the hyphen follows the
reference's name).  `abi` mirrors include/dbtk.h; `Dbtk` binds the C-ABI of
libdbtk_hip.so and raises if the HIP library is missing — there is no CPU path.
"""
from . import abi  # noqa: F401
